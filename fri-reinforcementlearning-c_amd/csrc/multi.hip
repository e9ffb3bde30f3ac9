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
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_MIN = 3, NCCL_FLOAT64 = 8 };      // ncclRedOp_t / ncclDataType_t values of rccl.h
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
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
    h[6] = (double)st.full_agents; h[7] = (double)st.episodes_max;
    h[8] = st.reward_min; h[16] = st.reward_max;
    if (hipMemcpyAsync(sh.d_stat, h, sizeof(double) * 24, hipMemcpyHostToDevice, sh.s) != hipSuccess) { set_error("frirl_hip_multi: stats upload failed"); return FRIRL_HIP_ELAUNCH; }
    int n = m->rccl.AllReduce(sh.d_stat, sh.d_stat, 7, NCCL_FLOAT64, NCCL_SUM, sh.comm, sh.s);
    if (n == 0) n = m->rccl.AllReduce(sh.d_stat + 7, sh.d_stat + 7, 1, NCCL_FLOAT64, NCCL_MAX, sh.comm, sh.s);       // episodes_max
    if (n == 0) n = m->rccl.AllReduce(sh.d_stat + 8, sh.d_stat + 8, 1, NCCL_FLOAT64, NCCL_MIN, sh.comm, sh.s);
    if (n == 0) n = m->rccl.AllReduce(sh.d_stat + 16, sh.d_stat + 16, 1, NCCL_FLOAT64, NCCL_MAX, sh.comm, sh.s);
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
    out->total_env_steps = (int64_t)h[5]; out->full_agents = (int64_t)h[6]; out->episodes_max = (int64_t)h[7];
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
