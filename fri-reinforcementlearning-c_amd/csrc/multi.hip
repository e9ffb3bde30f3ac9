// multi.hip -- many agents over several GPUs from plain C (frirl_hip_multi_*).
//
// The role of the reference's many-agent run modes (frirl_omp_run / frirl_mpi_run, src/frirl/frirl_agent.c:294-467) at node
// scale: the agents are sharded over the visible MI355X devices by GLOBAL environment id (balanced contiguous partition,
// frirl_hip_shard), every device owns a frirl_hip_batch and runs its episodes independently on its own host thread (the
// reference: one agent per OpenMP thread / MPI rank), and the ONLY exchange is the per-episode report (reward / steps / rules /
// converged sums + reward min / max, frirl_sequential_run.c:74-80): one RCCL all-reduce of 6 doubles (+ MIN, MAX) per episode
// over xGMI -- latency-bound, no data-path collective.  Single process, one communicator per device (ncclCommInitAll).
// RCCL is bound at first use with dlopen: the library itself does not depend on it (a process that already loaded an RCCL --
// torch -- keeps using that one; two copies in one process would not share a topology).
#include <dlfcn.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "batch_internal.h"
#include "device_common.h"

using namespace frirl_host;

extern "C" int frirl_hip_shard(int64_t total, int32_t world, int32_t rank, int64_t *start, int64_t *count)
{
    if (total < 0 || world < 1 || rank < 0 || rank >= world || !start || !count) { set_error("frirl_hip_shard: bad arguments"); return FRIRL_HIP_EINVAL; }
    const int64_t base = total / world, extra = total % world;
    *count = base + (rank < extra ? 1 : 0);
    *start = (int64_t)rank * base + (rank < extra ? rank : extra);
    return FRIRL_HIP_OK;
}

namespace {

// the few RCCL entry points used, resolved at run time
typedef struct ncclComm *ncclComm_t;
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_MIN = 3, NCCL_INT32 = 2, NCCL_FLOAT64 = 8 };      // ncclRedOp_t / ncclDataType_t values of rccl.h
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;      // rule-base exchange (train_merged)
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
};

bool rccl_load(Rccl &r)
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) { set_error("frirl_hip_multi: cannot load RCCL (librccl.so.1): %s", dlerror()); return false; }
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.lib, "ncclGetVersion"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.lib, "ncclBroadcast"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.lib, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.lib, "ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) { set_error("frirl_hip_multi: RCCL lacks ncclCommInitAll / ncclAllReduce"); return false; }
    return true;
}

struct Shard {
    int device = 0;
    int64_t start = 0, count = 0;
    frirl_hip_batch *batch = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t s = nullptr;
    double *d_stat = nullptr;        // [3][8]: sums, min, max (send = recv, in place)
    double h_stat[24];
    int rc = 0;
    char err[256];
    // rule-base exchange (frirl_hip_multi_train_merged), allocated by its first round
    double *d_stage = nullptr;       // [(nant+1)][maxR] the master's raw antecedent rows + consequents, broadcast from device 0
    int32_t *d_stage_i = nullptr;    // [1] the master's rule count
    double *d_pack_rconc = nullptr;  // devices >= 1: [count][maxR] consequent columns of the shard, packed for the send
    int32_t *d_pack_i = nullptr;     // devices >= 1: [2][count] rule counts, "complete" flags
    std::vector<double *> d_peer_rant, d_peer_rconc;      // device 0: what device g sent ([count_g][nant][maxR], [count_g][maxR])
    std::vector<int32_t *> d_peer_i;                      //           [2][count_g]
};

}  // namespace

struct frirl_hip_multi {
    Rccl rccl;
    std::vector<Shard> shards;
    int64_t total = 0;
    int32_t episodes = 0;
    frirl_hip_batch_stats_t last;
    int rccl_version = 0;
};

extern "C" void frirl_hip_multi_destroy(frirl_hip_multi *m)
{
    if (!m) return;
    for (Shard &sh : m->shards) {
        (void)hipSetDevice(sh.device);
        if (sh.comm && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(sh.comm);
        if (sh.batch) frirl_hip_batch_destroy(sh.batch);
        if (sh.d_stat) (void)hipFree(sh.d_stat);
        void *extra[] = {sh.d_stage, sh.d_stage_i, sh.d_pack_rconc, sh.d_pack_i};
        for (void *p : extra) if (p) (void)hipFree(p);
        for (double *p : sh.d_peer_rant) if (p) (void)hipFree(p);
        for (double *p : sh.d_peer_rconc) if (p) (void)hipFree(p);
        for (int32_t *p : sh.d_peer_i) if (p) (void)hipFree(p);
        if (sh.s) (void)hipStreamDestroy(sh.s);
    }
    delete m;
}

extern "C" frirl_hip_multi *frirl_hip_multi_create(const frirl_hip_batch_desc *d, int64_t total_agents, int32_t ngpus)
{
    if (!d || total_agents < 1) { set_error("frirl_hip_multi_create: bad arguments"); return nullptr; }
    if (check_device()) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("frirl_hip_multi_create: no device"); return nullptr; }
    if (ngpus <= 0) ngpus = ndev;
    if (ngpus > ndev) { set_error("frirl_hip_multi_create: %d GPUs requested, %d visible", ngpus, ndev); return nullptr; }
    if ((int64_t)ngpus > total_agents) ngpus = (int32_t)total_agents;
    frirl_hip_multi *m = new frirl_hip_multi();
    memset(&m->last, 0, sizeof m->last);
    m->total = total_agents;
    if (!rccl_load(m->rccl)) { delete m; return nullptr; }
    if (m->rccl.GetVersion) (void)m->rccl.GetVersion(&m->rccl_version);
    m->shards.resize(ngpus);
    std::vector<int> devs(ngpus);
    for (int g = 0; g < ngpus; g++) devs[g] = g;
    std::vector<ncclComm_t> comms(ngpus, nullptr);
    const int nrc = m->rccl.CommInitAll(comms.data(), ngpus, devs.data());
    if (nrc != 0) { set_error("frirl_hip_multi_create: ncclCommInitAll(%d): %s", ngpus, m->rccl.GetErrorString(nrc)); delete m; return nullptr; }
    for (int g = 0; g < ngpus; g++) {
        Shard &sh = m->shards[g];
        sh.device = g;
        sh.comm = comms[g];
        (void)frirl_hip_shard(total_agents, ngpus, g, &sh.start, &sh.count);
        if (hipSetDevice(g) != hipSuccess) { set_error("frirl_hip_multi_create: hipSetDevice(%d) failed", g); frirl_hip_multi_destroy(m); return nullptr; }
        frirl_hip_batch_desc dd = *d;
        dd.E = (int32_t)sh.count;
        dd.device = g;
        dd.agent.env_id_base = d->agent.env_id_base + (uint64_t)sh.start;      // RNG streams / start states keyed by the GLOBAL env id
        if (d->start_states) dd.start_states = d->start_states + (size_t)sh.start * (d->nant - 1);
        sh.batch = frirl_hip_batch_create(&dd);
        if (!sh.batch || hipStreamCreateWithFlags(&sh.s, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&sh.d_stat, sizeof(double) * 24) != hipSuccess) {
            frirl_hip_multi_destroy(m);
            return nullptr;
        }
    }
    (void)hipSetDevice(0);
    return m;
}

// local report of one shard -> all-reduced over the devices (SUM of 6 values, MIN and MAX of the reward); in place in h_stat
static int shard_allreduce(frirl_hip_multi *m, Shard &sh)
{
    frirl_hip_batch_stats_t st;
    int rc = frirl_hip_batch_stats(sh.batch, &st);
    if (rc) return rc;
    double *h = sh.h_stat;
    h[0] = st.reward_sum; h[1] = st.steps_sum; h[2] = st.rules_sum; h[3] = (double)st.converged; h[4] = (double)st.agents; h[5] = (double)st.total_env_steps;
    h[6] = (double)st.full_agents;
    h[7] = 0.0;                                   // "the master's rule base is complete": contributed by the device that owns global agent 0
    if (sh.start == 0) {
        const frirl_host::BatchView v = frirl_host::batch_view(sh.batch);
        int32_t done0 = 0;
        if (hipMemcpyAsync(&done0, v.d_converged, sizeof done0, hipMemcpyDeviceToHost, v.s) != hipSuccess || hipStreamSynchronize(v.s) != hipSuccess) {      // in the batch's stream order
            set_error("frirl_hip_multi: converged download failed");
            return FRIRL_HIP_ELAUNCH;
        }
        h[7] = done0 ? 1.0 : 0.0;
    }
    h[8] = st.reward_min; h[16] = st.reward_max; h[17] = (double)st.episodes_max;
    if (hipMemcpyAsync(sh.d_stat, h, sizeof(double) * 24, hipMemcpyHostToDevice, sh.s) != hipSuccess) { set_error("frirl_hip_multi: stats upload failed"); return FRIRL_HIP_ELAUNCH; }
    int n = m->rccl.AllReduce(sh.d_stat, sh.d_stat, 8, NCCL_FLOAT64, NCCL_SUM, sh.comm, sh.s);
    if (n == 0) n = m->rccl.AllReduce(sh.d_stat + 8, sh.d_stat + 8, 1, NCCL_FLOAT64, NCCL_MIN, sh.comm, sh.s);
    if (n == 0) n = m->rccl.AllReduce(sh.d_stat + 16, sh.d_stat + 16, 2, NCCL_FLOAT64, NCCL_MAX, sh.comm, sh.s);      // reward max, episodes_max
    if (n != 0) { set_error("frirl_hip_multi: ncclAllReduce: %s", m->rccl.GetErrorString(n)); return FRIRL_HIP_ELAUNCH; }
    if (hipMemcpyAsync(h, sh.d_stat, sizeof(double) * 24, hipMemcpyDeviceToHost, sh.s) != hipSuccess || hipStreamSynchronize(sh.s) != hipSuccess) {
        set_error("frirl_hip_multi: stats download failed: %s", hipGetErrorString(hipGetLastError()));
        return FRIRL_HIP_ELAUNCH;
    }
    return FRIRL_HIP_OK;
}

static void stats_from(const double *h, frirl_hip_batch_stats_t *out)
{
    memset(out, 0, sizeof *out);
    out->reward_sum = h[0]; out->steps_sum = h[1]; out->rules_sum = h[2]; out->converged = (int64_t)h[3]; out->agents = (int64_t)h[4];
    out->total_env_steps = (int64_t)h[5]; out->full_agents = (int64_t)h[6]; out->episodes_max = (int64_t)h[17];
    out->reward_min = h[8]; out->reward_max = h[16];
}

// One host thread per device (the reference: one OpenMP thread / MPI rank per agent): episodes until the GLOBAL report says every
// agent's rule base is complete.  Every thread sees the same all-reduced values, so all leave the loop in the same episode.
extern "C" int frirl_hip_multi_train(frirl_hip_multi *m, int32_t max_episodes, int32_t *episodes_run)
{
    if (!m) { set_error("frirl_hip_multi_train: NULL"); return FRIRL_HIP_EINVAL; }
    const int G = (int)m->shards.size();
    std::vector<int> eps(G, 0);
    auto worker = [&](int g) {
        Shard &sh = m->shards[g];
        sh.rc = 0;
        if (hipSetDevice(sh.device) != hipSuccess) { sh.rc = FRIRL_HIP_ENODEV; snprintf(sh.err, sizeof sh.err, "hipSetDevice(%d) failed", sh.device); return; }
        int ep = 0;
        for (ep = 1; ep < max_episodes; ep++) {             // at most max_episodes-1 episodes (frirl_sequential_run.c:51,59)
            int rc = frirl_hip_batch_episode(sh.batch);
            if (rc == 0) rc = shard_allreduce(m, sh);
            if (rc) {       // NOTE: a failing shard stops calling the collective; its peers would wait in RCCL -- report and abort
                sh.rc = rc; snprintf(sh.err, sizeof sh.err, "%s", frirl_hip_last_error());
                fprintf(stderr, "frirl_hip_multi_train: device %d failed in episode %d: %s\n", sh.device, ep, sh.err);
                abort();
            }
            if ((int64_t)sh.h_stat[3] >= (int64_t)sh.h_stat[4]) { ep++; break; }      // global: converged == agents
        }
        eps[g] = ep - 1;
    };
    if (G == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int g = 0; g < G; g++) th.emplace_back(worker, g);
        for (auto &t : th) t.join();
    }
    (void)hipSetDevice(0);
    for (int g = 0; g < G; g++) if (m->shards[g].rc) { set_error("frirl_hip_multi_train: device %d: %s", g, m->shards[g].err); return m->shards[g].rc; }
    m->episodes = eps[0];
    stats_from(m->shards[0].h_stat, &m->last);
    if (episodes_run) *episodes_run = eps[0];
    return FRIRL_HIP_OK;
}

// ---- the reference's many-agent mode WITH the rule-base exchange across devices (frirl_mpi_run's gather / scatter, frirl_agent.c:426-462;
// frirl_omp_run's round :424-462).  The master is the agent with global id 0 (device 0).  One round, every device in its own thread:
//   (1) the master's rule list (raw antecedent rows + consequents + count) is broadcast from device 0 (ncclBroadcast) and every other
//       agent takes it over -- one frirl_hip_merge_rb launch per device;
//   (2) devices >= 1 send their agents' rule lists to device 0 (ncclSend / ncclRecv: antecedent rows as they lie, consequent columns
//       packed, counts and "complete" flags), and the master takes over agent 1, 2, ... in GLOBAL id order -- the reference's order;
//   (3) every device restarts its convergence bookkeeping from the merged rule bases.
// Exactly frirl_hip_batch_merge_round when there is one device.
#define MCHK(call, what)                                                                                                     \
    do {                                                                                                                     \
        hipError_t e_ = (call);                                                                                              \
        if (e_ != hipSuccess) { set_error("frirl_hip_multi_train_merged: %s: %s", what, hipGetErrorString(e_)); return FRIRL_HIP_ELAUNCH; } \
    } while (0)
#define NCHK(call, what)                                                                                                     \
    do {                                                                                                                     \
        const int n_ = (call);                                                                                               \
        if (n_ != 0) { set_error("frirl_hip_multi_train_merged: %s: %s", what, m->rccl.GetErrorString(n_)); return FRIRL_HIP_ELAUNCH; } \
    } while (0)

static int merge_round_alloc(frirl_hip_multi *m, int g)
{
    Shard &sh = m->shards[g];
    if (sh.d_stage) return FRIRL_HIP_OK;
    const frirl_host::BatchView v = frirl_host::batch_view(sh.batch);
    const size_t n = v.nant, M = v.maxR;
    MCHK(hipMalloc((void **)&sh.d_stage, sizeof(double) * (n + 1) * M), "staging allocation");
    MCHK(hipMalloc((void **)&sh.d_stage_i, sizeof(int32_t) * 2), "staging allocation");
    if (g > 0) {
        MCHK(hipMalloc((void **)&sh.d_pack_rconc, sizeof(double) * (size_t)sh.count * M), "pack allocation");
        MCHK(hipMalloc((void **)&sh.d_pack_i, sizeof(int32_t) * 2 * (size_t)sh.count), "pack allocation");
    } else {
        const size_t G = m->shards.size();
        sh.d_peer_rant.assign(G, nullptr); sh.d_peer_rconc.assign(G, nullptr); sh.d_peer_i.assign(G, nullptr);
        for (size_t p = 1; p < G; p++) {
            const size_t c = (size_t)m->shards[p].count;
            MCHK(hipMalloc((void **)&sh.d_peer_rant[p], sizeof(double) * c * n * M), "receive allocation");
            MCHK(hipMalloc((void **)&sh.d_peer_rconc[p], sizeof(double) * c * M), "receive allocation");
            MCHK(hipMalloc((void **)&sh.d_peer_i[p], sizeof(int32_t) * 2 * c), "receive allocation");
        }
    }
    return FRIRL_HIP_OK;
}

static int merge_round_device(frirl_hip_multi *m, int g, int32_t *full_agents)
{
    using namespace frirl_host;
    Shard &sh = m->shards[g];
    const int G = (int)m->shards.size();
    const BatchView v = batch_view(sh.batch);
    const size_t n = v.nant, M = v.maxR;
    int rc = merge_round_alloc(m, g);
    if (rc) return rc;
    std::vector<int32_t> conv;
    if ((rc = batch_merge_prepare(sh.batch, conv))) return rc;
    if (!m->rccl.Broadcast || !m->rccl.Send || !m->rccl.Recv || !m->rccl.GroupStart || !m->rccl.GroupEnd) { set_error("frirl_hip_multi_train_merged: RCCL lacks ncclBroadcast / ncclSend / ncclRecv"); return FRIRL_HIP_ELAUNCH; }
    // (1) master -> everybody else
    if (g == 0) {
        MCHK(hipMemcpyAsync(sh.d_stage, v.d_rant, sizeof(double) * n * M, hipMemcpyDeviceToDevice, v.s), "master rows");
        MCHK(hipMemcpyAsync(sh.d_stage + n * M, v.d_rb + n * M, sizeof(double) * M, hipMemcpyDeviceToDevice, v.s), "master consequents");
        MCHK(hipMemcpyAsync(sh.d_stage_i, v.d_nrules, sizeof(int32_t), hipMemcpyDeviceToDevice, v.s), "master rule count");
    }
    NCHK(m->rccl.Broadcast(sh.d_stage, sh.d_stage, (n + 1) * M, NCCL_FLOAT64, 0, sh.comm, v.s), "ncclBroadcast(master rules)");
    NCHK(m->rccl.Broadcast(sh.d_stage_i, sh.d_stage_i, 1, NCCL_INT32, 0, sh.comm, v.s), "ncclBroadcast(master rule count)");
    frirl_hip_sender snd;
    memset(&snd, 0, sizeof snd);
    snd.rant = sh.d_stage; snd.rule_stride = 1; snd.dim_stride = (int64_t)M; snd.rconc = sh.d_stage + n * M; snd.S_dev = sh.d_stage_i;
    if ((rc = batch_merge_into_agents(sh.batch, &snd, g == 0))) return rc;
    // (2) everybody else -> master, in global id order
    if (g > 0) {
        const size_t c = (size_t)sh.count;
        MCHK(hipMemcpy2DAsync(sh.d_pack_rconc, sizeof(double) * M, v.d_rb + n * M, sizeof(double) * (n + 1) * M, sizeof(double) * M, c, hipMemcpyDeviceToDevice, v.s), "pack consequents");
        MCHK(hipMemcpyAsync(sh.d_pack_i, v.d_nrules, sizeof(int32_t) * c, hipMemcpyDeviceToDevice, v.s), "pack rule counts");
        MCHK(hipMemcpyAsync(sh.d_pack_i + c, v.d_converged, sizeof(int32_t) * c, hipMemcpyDeviceToDevice, v.s), "pack flags");
        NCHK(m->rccl.GroupStart(), "ncclGroupStart");
        NCHK(m->rccl.Send(v.d_rant, c * n * M, NCCL_FLOAT64, 0, sh.comm, v.s), "ncclSend(antecedents)");
        NCHK(m->rccl.Send(sh.d_pack_rconc, c * M, NCCL_FLOAT64, 0, sh.comm, v.s), "ncclSend(consequents)");
        NCHK(m->rccl.Send(sh.d_pack_i, 2 * c, NCCL_INT32, 0, sh.comm, v.s), "ncclSend(counts)");
        NCHK(m->rccl.GroupEnd(), "ncclGroupEnd");
    } else {
        std::vector<std::vector<int32_t>> peer_i(G);
        for (int p = 1; p < G; p++) {
            const size_t c = (size_t)m->shards[p].count;
            NCHK(m->rccl.GroupStart(), "ncclGroupStart");
            NCHK(m->rccl.Recv(sh.d_peer_rant[p], c * n * M, NCCL_FLOAT64, p, sh.comm, v.s), "ncclRecv(antecedents)");
            NCHK(m->rccl.Recv(sh.d_peer_rconc[p], c * M, NCCL_FLOAT64, p, sh.comm, v.s), "ncclRecv(consequents)");
            NCHK(m->rccl.Recv(sh.d_peer_i[p], 2 * c, NCCL_INT32, p, sh.comm, v.s), "ncclRecv(counts)");
            NCHK(m->rccl.GroupEnd(), "ncclGroupEnd");
            peer_i[p].resize(2 * c);
            MCHK(hipMemcpyAsync(peer_i[p].data(), sh.d_peer_i[p], sizeof(int32_t) * 2 * c, hipMemcpyDeviceToHost, v.s), "flags download");
        }
        MCHK(hipStreamSynchronize(v.s), "exchange sync");
        for (int id = 1; id < v.E; id++) {                       // this device's own agents come first in the global order
            if (conv[id]) continue;                              // a complete rule base does not send (:432,:444)
            const frirl_hip_sender own = batch_sender(sh.batch, id);
            if ((rc = batch_merge_into_first(sh.batch, &own))) return rc;
        }
        for (int p = 1; p < G; p++) {
            const size_t c = (size_t)m->shards[p].count;
            for (size_t j = 0; j < c; j++) {
                if (peer_i[p][c + j]) continue;
                frirl_hip_sender rs;
                memset(&rs, 0, sizeof rs);
                rs.rant = sh.d_peer_rant[p] + j * n * M; rs.rule_stride = 1; rs.dim_stride = (int64_t)M;
                rs.rconc = sh.d_peer_rconc[p] + j * M; rs.S_dev = sh.d_peer_i[p] + j;
                if ((rc = batch_merge_into_first(sh.batch, &rs))) return rc;
            }
        }
    }
    // (3)
    return batch_merge_finish(sh.batch, full_agents);
}

extern "C" int frirl_hip_multi_train_merged(frirl_hip_multi *m, int32_t max_episodes, int32_t chunk, int32_t *episodes_run, int32_t *rounds)
{
    if (!m || chunk < 2) { set_error("frirl_hip_multi_train_merged: bad arguments"); return FRIRL_HIP_EINVAL; }
    const int G = (int)m->shards.size();
    std::vector<int> eps(G, 0), nrounds(G, 0);
    auto worker = [&](int g) {
        Shard &sh = m->shards[g];
        sh.rc = 0;
        auto fail = [&](int rc, const char *what, int ep) {    // a failing device stops calling the collectives; its peers would wait in RCCL
            sh.rc = rc; snprintf(sh.err, sizeof sh.err, "%s", frirl_hip_last_error());
            fprintf(stderr, "frirl_hip_multi_train_merged: device %d failed in %s (episode %d): %s\n", sh.device, what, ep, sh.err);
            abort();
        };
        if (hipSetDevice(sh.device) != hipSuccess) { sh.rc = FRIRL_HIP_ENODEV; snprintf(sh.err, sizeof sh.err, "hipSetDevice(%d) failed", sh.device); return; }
        int ep = 1, nr = 0;
        for (;;) {                                              // frirl_omp_run's loop (frirl_agent.c:424-462): every device takes the same path
            bool master_done = false;
            for (int c = 1; c < chunk && ep < max_episodes; c++, ep++) {
                int rc = frirl_hip_batch_episode(sh.batch);
                if (rc == 0) rc = shard_allreduce(m, sh);
                if (rc) fail(rc, "episode", ep);
                master_done = sh.h_stat[7] > 0.0;
                if (master_done) { ep++; break; }
            }
            if (master_done || ep >= max_episodes) break;
            const int rc = merge_round_device(m, g, nullptr);
            if (rc) fail(rc, "merge round", ep);
            nr++;
        }
        eps[g] = ep - 1;
        nrounds[g] = nr;
    };
    if (G == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int g = 0; g < G; g++) th.emplace_back(worker, g);
        for (auto &t : th) t.join();
    }
    (void)hipSetDevice(0);
    for (int g = 0; g < G; g++) if (m->shards[g].rc) { set_error("frirl_hip_multi_train_merged: device %d: %s", g, m->shards[g].err); return m->shards[g].rc; }
    m->episodes = eps[0];
    stats_from(m->shards[0].h_stat, &m->last);
    if (episodes_run) *episodes_run = eps[0];
    if (rounds) *rounds = nrounds[0];
    return FRIRL_HIP_OK;
}

extern "C" int frirl_hip_multi_stats(frirl_hip_multi *m, frirl_hip_batch_stats_t *out)
{
    if (!m || !out) { set_error("frirl_hip_multi_stats: NULL"); return FRIRL_HIP_EINVAL; }
    const int G = (int)m->shards.size();
    std::vector<int> rcs(G, 0);
    auto worker = [&](int g) {
        Shard &sh = m->shards[g];
        rcs[g] = (hipSetDevice(sh.device) == hipSuccess) ? shard_allreduce(m, sh) : FRIRL_HIP_ENODEV;
        if (rcs[g]) { fprintf(stderr, "frirl_hip_multi_stats: device %d failed: %s\n", sh.device, frirl_hip_last_error()); abort(); }
    };
    if (G == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int g = 0; g < G; g++) th.emplace_back(worker, g);
        for (auto &t : th) t.join();
    }
    (void)hipSetDevice(0);
    stats_from(m->shards[0].h_stat, out);
    m->last = *out;
    return FRIRL_HIP_OK;
}

extern "C" int frirl_hip_multi_info(const frirl_hip_multi *m, int32_t *ngpus, int32_t *rccl_version, int64_t *shard_start, int64_t *shard_count)
{
    if (!m) { set_error("frirl_hip_multi_info: NULL"); return FRIRL_HIP_EINVAL; }
    if (ngpus) *ngpus = (int32_t)m->shards.size();
    if (rccl_version) *rccl_version = m->rccl_version;
    for (size_t g = 0; g < m->shards.size(); g++) {
        if (shard_start) shard_start[g] = m->shards[g].start;
        if (shard_count) shard_count[g] = m->shards[g].count;
    }
    return FRIRL_HIP_OK;
}

// rule base of the agent with GLOBAL id `agent`: routed to the device that owns it
extern "C" int frirl_hip_multi_get_rulebase(frirl_hip_multi *m, int64_t agent, int32_t *R, double *rant, double *rconc)
{
    if (!m || agent < 0 || agent >= m->total) { set_error("frirl_hip_multi_get_rulebase: bad agent id"); return FRIRL_HIP_EINVAL; }
    for (Shard &sh : m->shards)
        if (agent >= sh.start && agent < sh.start + sh.count) {
            if (hipSetDevice(sh.device) != hipSuccess) { set_error("frirl_hip_multi_get_rulebase: hipSetDevice failed"); return FRIRL_HIP_ENODEV; }
            const int rc = frirl_hip_batch_get_rulebase(sh.batch, (int32_t)(agent - sh.start), R, rant, rconc);
            (void)hipSetDevice(0);
            return rc;
        }
    set_error("frirl_hip_multi_get_rulebase: agent not found");
    return FRIRL_HIP_EINVAL;
}
