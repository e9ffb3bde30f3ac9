// batch_internal.h -- the pieces of a rule-base merge round of ONE frirl_hip_batch (batch.hip), shared with the multi-device runner
// (multi.hip), which strings them together across devices.  Not part of the C ABI.
#pragma once

#include <vector>

#include "device_common.h"

struct frirl_hip_batch;

namespace frirl_host {

struct BatchView {                 // what the multi-device exchange needs to see of a batch
    int nant, maxR, E, device;
    hipStream_t s;
    double *d_rant;                // [E][nant][maxR] raw antecedents
    double *d_rb;                  // [E][nant+1][maxR] VE antecedents + consequents
    int32_t *d_nrules, *d_converged, *d_epended;
};
BatchView batch_view(frirl_hip_batch *b);
// frirl_hip_batch_stats + whether agent 0 of the batch has converged (NULL: not wanted)
int batch_stats_first(frirl_hip_batch *b, frirl_hip_batch_stats_t *out, int32_t *first_converged);

// start of a merge round: the receivers' FIVERB.weights as learning left them (frirl_hip_weights_from_spread); pended = this chunk's
// `epended` flags of the agents (host copy): an agent that has "pended" neither sends nor -- the master -- is sent (frirl_agent.c:338,352)
int batch_merge_prepare(frirl_hip_batch *b, std::vector<int32_t> &pended);
// every agent of the batch (but agent 0 when skip_first) takes over the sender's rules: one launch of frirl_hip_merge_rb
int batch_merge_into_agents(frirl_hip_batch *b, const frirl_hip_sender *snd, bool skip_first);
// agent 0 of the batch (the master) takes over the sender's rules
int batch_merge_into_first(frirl_hip_batch *b, const frirl_hip_sender *snd);
// sender descriptor for agent id of this batch (a row set of its own SoA store)
frirl_hip_sender batch_sender(frirl_hip_batch *b, int id);
// end of a round: *full_agents += agents at capacity; every agent (but agent 0 when first_is_master) is set running again, `epended`
// cleared, and the rule count / consequent snapshot of the convergence test retaken from the merged rule bases
int batch_merge_finish(frirl_hip_batch *b, int32_t *full_agents, bool first_is_master);

}  // namespace frirl_host
