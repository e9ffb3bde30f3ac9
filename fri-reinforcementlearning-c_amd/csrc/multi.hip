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
//
// The three exchanges a run needs -- the report (every shard learns every shard's values), the master's broadcast and the shards'
// send / recv of their rule lists -- sit behind a small Transport interface with two implementations: RCCL over xGMI (one device per
// shard), and a LOOP-BACK transport (option "multi_loopback" / FRIRL_HIP_MULTI_LOOPBACK=1) that runs the same shards, threads, packing
// copies and merge order as LOGICAL shards on ONE device and moves the bytes with device-to-device copies: the test box has one
// GPU, and this is how every g > 0 / p >= 1 branch below is executed there (tests/test_multi.py).
// A shard that fails does not abort the process: its error rides in the report, so every shard leaves the loop in the same episode;
// a failure inside an exchange raises the shared `failed` flag, which ends every peer's wait (RCCL: ncclCommAbort of its communicator).
#include <dlfcn.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
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
enum { NCCL_SUM = 0, NCCL_INT8 = 0, NCCL_FLOAT64 = 8 };      // ncclRedOp_t / ncclDataType_t values of rccl.h (checked against /opt/rocm/include/rccl/rccl.h)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;
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
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
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
    double *d_gather = nullptr;      // RCCL: [G][REPORT_K] the report of every shard (own slot filled, SUM all-reduce = all-gather)
    double h_stat[24];               // the combined report: sums [0..8), reward min [8], reward max [16], episodes_max [17]
    int rc = 0;
    char err[256];
    // rule-base exchange (frirl_hip_multi_train_merged), allocated by its first round
    double *d_stage = nullptr;       // [(nant+1)][maxR] the master's raw antecedent rows + consequents, broadcast from shard 0
    int32_t *d_stage_i = nullptr;    // [1] the master's rule count
    double *d_pack_rconc = nullptr;  // shards >= 1: [count][maxR] consequent columns of the shard, packed for the send
    int32_t *d_pack_i = nullptr;     // shards >= 1: [2][count] rule counts, "complete" flags
    std::vector<double *> d_peer_rant, d_peer_rconc;      // shard 0: what shard g sent ([count_g][nant][maxR], [count_g][maxR])
    std::vector<int32_t *> d_peer_i;                      //          [2][count_g]
};

// values of one shard's report; combined in shard order, so every shard computes the same bits
enum { REP_REWARD_SUM, REP_STEPS_SUM, REP_RULES_SUM, REP_CONVERGED, REP_AGENTS, REP_ENV_STEPS, REP_FULL, REP_MASTER_DONE, REP_REWARD_MIN, REP_REWARD_MAX,
       REP_EPISODES_MAX, REP_ERROR, REPORT_K };

// The exchanges between shards.  Every call is made by the shard's own host thread; `g` is the calling shard.  A call returns only
// when its bytes have arrived (the callers need the values on the host next anyway) and fails, instead of waiting for ever, once
// any shard has raised `failed`.
struct Transport {
    virtual ~Transport() {}
    virtual const char *name() const = 0;
    // all[G][REPORT_K] <- the report of every shard (mine = this shard's REPORT_K values), same on every shard
    virtual int exchange_report(int g, const double *mine, double *all) = 0;
    virtual int broadcast(int g, void *buf, size_t bytes, int root, hipStream_t s) = 0;                       // device buffers
    // one batch of point-to-point transfers of shard g (device buffers): posted, then completed together by flush()
    virtual int send(int g, const void *buf, size_t bytes, int peer, hipStream_t s) = 0;
    virtual int recv(int g, void *buf, size_t bytes, int peer, hipStream_t s) = 0;
    virtual int begin(int g) = 0;
    virtual int flush(int g, hipStream_t s) = 0;
    virtual void abort(int g) = 0;                                                                             // after `failed` was raised
};

}  // namespace

struct frirl_hip_multi {
    Rccl rccl;
    std::vector<Shard> shards;
    std::unique_ptr<Transport> tr;
    std::atomic<bool> failed{false};
    bool loopback = false;
    int64_t total = 0;
    int32_t episodes = 0;
    frirl_hip_batch_stats_t last;
    int rccl_version = 0;
};

namespace {

// waits for a stream without blocking inside the runtime: a peer's failure must be able to end the wait
int wait_stream(frirl_hip_multi *m, int g, hipStream_t s, const char *what)
{
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return FRIRL_HIP_OK;
        if (q != hipErrorNotReady) { set_error("frirl_hip_multi: %s: %s", what, hipGetErrorString(q)); (void)hipGetLastError(); return FRIRL_HIP_ELAUNCH; }
        if (m->failed.load()) { m->tr->abort(g); set_error("frirl_hip_multi: %s: another shard failed", what); return FRIRL_HIP_ELAUNCH; }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// ---- RCCL over xGMI: one device and one communicator per shard -------------------------------------------------------------
struct RcclTransport : Transport {
    frirl_hip_multi *m;
    explicit RcclTransport(frirl_hip_multi *mm) : m(mm) {}
    const char *name() const override { return "RCCL"; }
    int nerr(int n, const char *what) { if (n != 0) { set_error("frirl_hip_multi: %s: %s", what, m->rccl.GetErrorString(n)); return FRIRL_HIP_ELAUNCH; } return FRIRL_HIP_OK; }
    int exchange_report(int g, const double *mine, double *all) override
    {
        // ONE all-reduce per episode: every shard contributes its values in its own slot of a zeroed [G][K] array, the SUM is the
        // all-gather (x + 0 is exact), and min / max / sums are then formed on the host in shard order
        Shard &sh = m->shards[g];
        const size_t G = m->shards.size(), n = G * REPORT_K;
        std::vector<double> h(n, 0.0);
        memcpy(h.data() + (size_t)g * REPORT_K, mine, sizeof(double) * REPORT_K);
        if (hipMemcpyAsync(sh.d_gather, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, sh.s) != hipSuccess) { set_error("frirl_hip_multi: report upload failed"); return FRIRL_HIP_ELAUNCH; }
        int rc = nerr(m->rccl.AllReduce(sh.d_gather, sh.d_gather, n, NCCL_FLOAT64, NCCL_SUM, sh.comm, sh.s), "ncclAllReduce(report)");
        if (rc) return rc;
        if (hipMemcpyAsync(all, sh.d_gather, sizeof(double) * n, hipMemcpyDeviceToHost, sh.s) != hipSuccess) { set_error("frirl_hip_multi: report download failed"); return FRIRL_HIP_ELAUNCH; }
        return wait_stream(m, g, sh.s, "report exchange");
    }
    int broadcast(int g, void *buf, size_t bytes, int root, hipStream_t s) override
    {
        if (!m->rccl.Broadcast) { set_error("frirl_hip_multi: RCCL lacks ncclBroadcast"); return FRIRL_HIP_ELAUNCH; }
        int rc = nerr(m->rccl.Broadcast(buf, buf, bytes, NCCL_INT8, root, m->shards[g].comm, s), "ncclBroadcast");
        return rc ? rc : wait_stream(m, g, s, "broadcast");
    }
    int begin(int) override { if (!m->rccl.GroupStart || !m->rccl.Send || !m->rccl.Recv) { set_error("frirl_hip_multi: RCCL lacks ncclSend / ncclRecv"); return FRIRL_HIP_ELAUNCH; } return nerr(m->rccl.GroupStart(), "ncclGroupStart"); }
    int send(int g, const void *buf, size_t bytes, int peer, hipStream_t s) override { return nerr(m->rccl.Send(buf, bytes, NCCL_INT8, peer, m->shards[g].comm, s), "ncclSend"); }
    int recv(int g, void *buf, size_t bytes, int peer, hipStream_t s) override { return nerr(m->rccl.Recv(buf, bytes, NCCL_INT8, peer, m->shards[g].comm, s), "ncclRecv"); }
    int flush(int g, hipStream_t s) override
    {
        int rc = nerr(m->rccl.GroupEnd(), "ncclGroupEnd");
        return rc ? rc : wait_stream(m, g, s, "send / recv");
    }
    void abort(int g) override { Shard &sh = m->shards[g]; if (sh.comm && m->rccl.CommAbort) { (void)m->rccl.CommAbort(sh.comm); sh.comm = nullptr; } }
};

// ---- loop-back: the shards are LOGICAL shards of one device; bytes move by device-to-device copies on the receiver's stream ----
struct LoopbackTransport : Transport {
    frirl_hip_multi *m;
    int G;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<std::vector<double>> slots;              // reports
    const void *bc_src = nullptr;                        // broadcast source of the current round
    struct Msg { const void *buf; size_t bytes; bool done; };
    std::vector<std::deque<Msg *>> box;                  // [src * G + dst] posted sends, FIFO
    std::vector<std::vector<Msg *>> mine;                // [g] this shard's posted sends of the current batch
    explicit LoopbackTransport(frirl_hip_multi *mm, int g) : m(mm), G(g), slots(g, std::vector<double>(REPORT_K, 0.0)), box((size_t)g * g), mine(g) {}
    const char *name() const override { return "loop-back"; }
    bool barrier()                                        // false once a shard has failed
    {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        if (++arrived == G) { arrived = 0; generation++; cv.notify_all(); return !m->failed.load(); }
        while (generation == gen) {
            if (m->failed.load()) return false;
            cv.wait_for(lk, std::chrono::milliseconds(1));
        }
        return !m->failed.load();
    }
    int peer_failed(const char *what) { set_error("frirl_hip_multi (loop-back): %s: another shard failed", what); return FRIRL_HIP_ELAUNCH; }
    int exchange_report(int g, const double *mine_v, double *all) override
    {
        { std::lock_guard<std::mutex> lk(mu); memcpy(slots[g].data(), mine_v, sizeof(double) * REPORT_K); }
        if (!barrier()) return peer_failed("report");
        for (int p = 0; p < G; p++) memcpy(all + (size_t)p * REPORT_K, slots[p].data(), sizeof(double) * REPORT_K);
        if (!barrier()) return peer_failed("report");    // nobody overwrites a slot before everybody has read it
        return FRIRL_HIP_OK;
    }
    int broadcast(int g, void *buf, size_t bytes, int root, hipStream_t s) override
    {
        if (g == root) {
            int rc = wait_stream(m, g, s, "broadcast source");
            if (rc) return rc;
            std::lock_guard<std::mutex> lk(mu);
            bc_src = buf;
        }
        if (!barrier()) return peer_failed("broadcast");
        if (g != root) {
            if (hipMemcpyAsync(buf, bc_src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { set_error("frirl_hip_multi (loop-back): broadcast copy failed"); return FRIRL_HIP_ELAUNCH; }
            int rc = wait_stream(m, g, s, "broadcast copy");
            if (rc) return rc;
        }
        if (!barrier()) return peer_failed("broadcast");  // the root may reuse its buffer
        return FRIRL_HIP_OK;
    }
    int begin(int g) override { mine[g].clear(); return FRIRL_HIP_OK; }
    int send(int g, const void *buf, size_t bytes, int peer, hipStream_t s) override
    {
        int rc = wait_stream(m, g, s, "send source");    // the bytes are final before they are offered
        if (rc) return rc;
        Msg *msg = new Msg{buf, bytes, false};
        std::lock_guard<std::mutex> lk(mu);
        box[(size_t)g * G + peer].push_back(msg);
        mine[g].push_back(msg);
        cv.notify_all();
        return FRIRL_HIP_OK;
    }
    int recv(int g, void *buf, size_t bytes, int peer, hipStream_t s) override
    {
        Msg *msg = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu);
            std::deque<Msg *> &q = box[(size_t)peer * G + g];
            while (q.empty()) {
                if (m->failed.load()) return peer_failed("recv");
                cv.wait_for(lk, std::chrono::milliseconds(1));
            }
            msg = q.front();
            q.pop_front();
        }
        if (msg->bytes != bytes) { set_error("frirl_hip_multi (loop-back): recv of %zu B meets a send of %zu B", bytes, msg->bytes); return FRIRL_HIP_EINVAL; }
        if (hipMemcpyAsync(buf, msg->buf, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { set_error("frirl_hip_multi (loop-back): recv copy failed"); return FRIRL_HIP_ELAUNCH; }
        int rc = wait_stream(m, g, s, "recv copy");
        std::lock_guard<std::mutex> lk(mu);
        msg->done = true;
        cv.notify_all();
        return rc;
    }
    int flush(int g, hipStream_t) override               // a sender's buffers stay untouched until every receiver has copied them
    {
        std::unique_lock<std::mutex> lk(mu);
        for (Msg *msg : mine[g]) {
            while (!msg->done) {
                if (m->failed.load()) return peer_failed("send");
                cv.wait_for(lk, std::chrono::milliseconds(1));
            }
            delete msg;
        }
        mine[g].clear();
        return FRIRL_HIP_OK;
    }
    void abort(int) override {}
};

}  // namespace

extern "C" void frirl_hip_multi_destroy(frirl_hip_multi *m)
{
    if (!m) return;
    DeviceGuard keep;
    for (Shard &sh : m->shards) {
        (void)hipSetDevice(sh.device);
        if (sh.comm && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(sh.comm);
        if (sh.batch) frirl_hip_batch_destroy(sh.batch);
        if (sh.d_gather) (void)hipFree(sh.d_gather);
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
    DeviceGuard keep;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("frirl_hip_multi_create: no device"); return nullptr; }
    const bool loopback = opts().multi_loopback == 1;      // logical shards on the CURRENT device (tests; one-GPU boxes)
    if (ngpus <= 0) ngpus = ndev;
    if (!loopback && ngpus > ndev) { set_error("frirl_hip_multi_create: %d GPUs requested, %d visible", ngpus, ndev); return nullptr; }
    if ((int64_t)ngpus > total_agents) ngpus = (int32_t)total_agents;
    int cur = 0;
    (void)hipGetDevice(&cur);
    frirl_hip_multi *m = new frirl_hip_multi();
    memset(&m->last, 0, sizeof m->last);
    m->total = total_agents;
    m->loopback = loopback;
    m->shards.resize(ngpus);
    std::vector<ncclComm_t> comms(ngpus, nullptr);
    if (loopback) {
        m->tr.reset(new LoopbackTransport(m, ngpus));
    } else {
        if (!rccl_load(m->rccl)) { delete m; return nullptr; }
        if (m->rccl.GetVersion) (void)m->rccl.GetVersion(&m->rccl_version);
        std::vector<int> devs(ngpus);
        for (int g = 0; g < ngpus; g++) devs[g] = g;
        const int nrc = m->rccl.CommInitAll(comms.data(), ngpus, devs.data());
        if (nrc != 0) { set_error("frirl_hip_multi_create: ncclCommInitAll(%d): %s", ngpus, m->rccl.GetErrorString(nrc)); delete m; return nullptr; }
        m->tr.reset(new RcclTransport(m));
    }
    for (int g = 0; g < ngpus; g++) {
        Shard &sh = m->shards[g];
        sh.device = loopback ? cur : g;
        sh.comm = comms[g];
        (void)frirl_hip_shard(total_agents, ngpus, g, &sh.start, &sh.count);
        if (hipSetDevice(sh.device) != hipSuccess) { set_error("frirl_hip_multi_create: hipSetDevice(%d) failed", sh.device); frirl_hip_multi_destroy(m); return nullptr; }
        frirl_hip_batch_desc dd = *d;
        dd.E = (int32_t)sh.count;
        dd.device_select = 1;
        dd.device = sh.device;
        dd.agent.env_id_base = d->agent.env_id_base + (uint64_t)sh.start;      // RNG streams / start states keyed by the GLOBAL env id
        if (d->start_states) dd.start_states = d->start_states + (size_t)sh.start * (d->nant - 1);
        sh.batch = frirl_hip_batch_create(&dd);
        if (!sh.batch || hipStreamCreateWithFlags(&sh.s, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void **)&sh.d_gather, sizeof(double) * REPORT_K * (size_t)ngpus) != hipSuccess) {
            frirl_hip_multi_destroy(m);
            return nullptr;
        }
    }
    return m;
}

// local report of one shard (local_error: what this shard has to tell the others) -> every shard's report -> combined in shard order
static int shard_report(frirl_hip_multi *m, int g, int local_error)
{
    Shard &sh = m->shards[g];
    const int G = (int)m->shards.size();
    double mine[REPORT_K];
    memset(mine, 0, sizeof mine);
    frirl_hip_batch_stats_t st;
    int32_t first_done = 0;
    int rc = local_error ? local_error : frirl_host::batch_stats_first(sh.batch, &st, &first_done);
    if (rc == 0) {
        mine[REP_REWARD_SUM] = st.reward_sum; mine[REP_STEPS_SUM] = st.steps_sum; mine[REP_RULES_SUM] = st.rules_sum; mine[REP_CONVERGED] = (double)st.converged;
        mine[REP_AGENTS] = (double)st.agents; mine[REP_ENV_STEPS] = (double)st.total_env_steps; mine[REP_FULL] = (double)st.full_agents;
        mine[REP_MASTER_DONE] = (sh.start == 0 && first_done) ? 1.0 : 0.0;      // "the master's rule base is complete": from the shard that owns global agent 0
        mine[REP_REWARD_MIN] = st.reward_min; mine[REP_REWARD_MAX] = st.reward_max; mine[REP_EPISODES_MAX] = (double)st.episodes_max;
    } else {
        snprintf(sh.err, sizeof sh.err, "%s", frirl_hip_last_error());
        sh.rc = rc;
        mine[REP_ERROR] = 1.0;                      // rides with the report: every shard leaves the loop in this episode
    }
    std::vector<double> all((size_t)G * REPORT_K);
    const int xrc = m->tr->exchange_report(g, mine, all.data());
    if (xrc) return xrc;
    double *h = sh.h_stat;
    memset(h, 0, sizeof sh.h_stat);
    bool any = false, err = false;
    for (int p = 0; p < G; p++) {
        const double *r = all.data() + (size_t)p * REPORT_K;
        if (r[REP_ERROR] != 0.0) { err = true; continue; }
        for (int i = 0; i < 8; i++) h[i] += r[i];
        if (!any || r[REP_REWARD_MIN] < h[8]) h[8] = r[REP_REWARD_MIN];
        if (!any || r[REP_REWARD_MAX] > h[16]) h[16] = r[REP_REWARD_MAX];
        if (!any || r[REP_EPISODES_MAX] > h[17]) h[17] = r[REP_EPISODES_MAX];
        any = true;
    }
    if (err) {
        if (!sh.rc) { sh.rc = FRIRL_HIP_ELAUNCH; snprintf(sh.err, sizeof sh.err, "another shard reported an error"); set_error("frirl_hip_multi: another shard reported an error"); }
        return sh.rc;
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

// a shard could not finish an exchange: the others must not wait for it
static void shard_failed(frirl_hip_multi *m, int g, int rc, const char *what, int ep)
{
    Shard &sh = m->shards[g];
    if (!sh.rc) { sh.rc = rc; snprintf(sh.err, sizeof sh.err, "%s (%s, episode %d)", frirl_hip_last_error(), what, ep); }
    m->failed.store(true);
    m->tr->abort(g);
}

template <class F>
static int run_shards(frirl_hip_multi *m, const char *who, F &&worker)
{
    const int G = (int)m->shards.size();
    DeviceGuard keep;
    m->failed.store(false);
    for (Shard &sh : m->shards) { sh.rc = 0; sh.err[0] = 0; }
    auto body = [&](int g) {
        Shard &sh = m->shards[g];
        if (hipSetDevice(sh.device) != hipSuccess) { set_error("hipSetDevice(%d) failed", sh.device); shard_failed(m, g, FRIRL_HIP_ENODEV, "start", 0); return; }
        worker(g);
    };
    if (G == 1) body(0);
    else {                                                  // one host thread per shard (the reference: one OpenMP thread / MPI rank per agent)
        std::vector<std::thread> th;
        for (int g = 0; g < G; g++) th.emplace_back(body, g);
        for (auto &t : th) t.join();
    }
    int first = -1;                                         // the shard that failed by itself, not one that only heard of it
    for (int g = 0; g < G; g++) if (m->shards[g].rc && (first < 0 || (strstr(m->shards[first].err, "another shard") && !strstr(m->shards[g].err, "another shard")))) first = g;
    if (first >= 0) { set_error("%s: shard %d (device %d): %s", who, first, m->shards[first].device, m->shards[first].err); return m->shards[first].rc; }
    return FRIRL_HIP_OK;
}

// Episodes until the GLOBAL report says every agent's rule base is complete.  Every shard sees the same combined values, so all
// leave the loop in the same episode -- also when one of them reports an error.
extern "C" int frirl_hip_multi_train(frirl_hip_multi *m, int32_t max_episodes, int32_t *episodes_run)
{
    if (!m) { set_error("frirl_hip_multi_train: NULL"); return FRIRL_HIP_EINVAL; }
    const int G = (int)m->shards.size();
    std::vector<int> eps(G, 0);
    const int rc = run_shards(m, "frirl_hip_multi_train", [&](int g) {
        Shard &sh = m->shards[g];
        int ep = 0;
        for (ep = 1; ep < max_episodes; ep++) {             // at most max_episodes-1 episodes (frirl_sequential_run.c:51,59)
            const int erc = frirl_hip_batch_episode(sh.batch);
            const int rrc = shard_report(m, g, erc);
            if (rrc) { if (!sh.rc) shard_failed(m, g, rrc, "report", ep); break; }
            if ((int64_t)sh.h_stat[3] >= (int64_t)sh.h_stat[4]) { ep++; break; }      // global: converged == agents
        }
        eps[g] = ep - 1;
    });
    if (rc) return rc;
    m->episodes = eps[0];
    stats_from(m->shards[0].h_stat, &m->last);
    if (episodes_run) *episodes_run = eps[0];
    return FRIRL_HIP_OK;
}

// ---- the reference's many-agent mode WITH the rule-base exchange across devices (frirl_mpi_run's gather / scatter, frirl_agent.c:426-462;
// frirl_omp_run's round :424-462).  The master is the agent with global id 0 (shard 0).  One round, every shard in its own thread:
//   (1) the master's rule list (raw antecedent rows + consequents + count) is broadcast from shard 0 and every other agent takes it
//       over -- one frirl_hip_merge_rb launch per shard;
//   (2) shards >= 1 send their agents' rule lists to shard 0 (antecedent rows as they lie, consequent columns packed, counts and
//       "complete" flags), and the master takes over agent 1, 2, ... in GLOBAL id order -- the reference's order;
//   (3) every shard restarts its convergence bookkeeping from the merged rule bases.
// Exactly frirl_hip_batch_merge_round when there is one shard.
#define MCHK(call, what)                                                                                                     \
    do {                                                                                                                     \
        hipError_t e_ = (call);                                                                                              \
        if (e_ != hipSuccess) { set_error("frirl_hip_multi_train_merged: %s: %s", what, hipGetErrorString(e_)); return FRIRL_HIP_ELAUNCH; } \
    } while (0)
#define TCHK(call)                                                                                                           \
    do {                                                                                                                     \
        const int t_ = (call);                                                                                               \
        if (t_ != 0) return t_;                                                                                              \
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

static int merge_round_shard(frirl_hip_multi *m, int g, int32_t *full_agents)
{
    using namespace frirl_host;
    Shard &sh = m->shards[g];
    Transport &tr = *m->tr;
    const int G = (int)m->shards.size();
    const BatchView v = batch_view(sh.batch);
    const size_t n = v.nant, M = v.maxR;
    int rc = merge_round_alloc(m, g);
    if (rc) return rc;
    std::vector<int32_t> conv;
    if ((rc = batch_merge_prepare(sh.batch, conv))) return rc;
    // (1) master -> everybody else
    if (g == 0) {
        MCHK(hipMemcpyAsync(sh.d_stage, v.d_rant, sizeof(double) * n * M, hipMemcpyDeviceToDevice, v.s), "master rows");
        MCHK(hipMemcpyAsync(sh.d_stage + n * M, v.d_rb + n * M, sizeof(double) * M, hipMemcpyDeviceToDevice, v.s), "master consequents");
        MCHK(hipMemcpyAsync(sh.d_stage_i, v.d_nrules, sizeof(int32_t), hipMemcpyDeviceToDevice, v.s), "master rule count");
        if (conv[0]) MCHK(hipMemsetAsync(sh.d_stage_i, 0, sizeof(int32_t), v.s), "master pended");      // "#0 pended, skipping sync to master": nobody takes its rules over (:338-350)
    }
    TCHK(tr.broadcast(g, sh.d_stage, sizeof(double) * (n + 1) * M, 0, v.s));
    TCHK(tr.broadcast(g, sh.d_stage_i, sizeof(int32_t), 0, v.s));
    frirl_hip_sender snd;
    memset(&snd, 0, sizeof snd);
    snd.rant = sh.d_stage; snd.rule_stride = 1; snd.dim_stride = (int64_t)M; snd.rconc = sh.d_stage + n * M; snd.S_dev = sh.d_stage_i;
    if ((rc = batch_merge_into_agents(sh.batch, &snd, g == 0))) return rc;
    // (2) everybody else -> master, in global id order
    if (g > 0) {
        const size_t c = (size_t)sh.count;
        MCHK(hipMemcpy2DAsync(sh.d_pack_rconc, sizeof(double) * M, v.d_rb + n * M, sizeof(double) * (n + 1) * M, sizeof(double) * M, c, hipMemcpyDeviceToDevice, v.s), "pack consequents");
        MCHK(hipMemcpyAsync(sh.d_pack_i, v.d_nrules, sizeof(int32_t) * c, hipMemcpyDeviceToDevice, v.s), "pack rule counts");
        MCHK(hipMemcpyAsync(sh.d_pack_i + c, v.d_epended, sizeof(int32_t) * c, hipMemcpyDeviceToDevice, v.s), "pack flags");
        TCHK(tr.begin(g));
        TCHK(tr.send(g, v.d_rant, sizeof(double) * c * n * M, 0, v.s));
        TCHK(tr.send(g, sh.d_pack_rconc, sizeof(double) * c * M, 0, v.s));
        TCHK(tr.send(g, sh.d_pack_i, sizeof(int32_t) * 2 * c, 0, v.s));
        TCHK(tr.flush(g, v.s));
    } else {
        std::vector<std::vector<int32_t>> peer_i(G);
        for (int p = 1; p < G; p++) {
            const size_t c = (size_t)m->shards[p].count;
            TCHK(tr.begin(g));
            TCHK(tr.recv(g, sh.d_peer_rant[p], sizeof(double) * c * n * M, p, v.s));
            TCHK(tr.recv(g, sh.d_peer_rconc[p], sizeof(double) * c * M, p, v.s));
            TCHK(tr.recv(g, sh.d_peer_i[p], sizeof(int32_t) * 2 * c, p, v.s));
            TCHK(tr.flush(g, v.s));
            peer_i[p].resize(2 * c);
            MCHK(hipMemcpyAsync(peer_i[p].data(), sh.d_peer_i[p], sizeof(int32_t) * 2 * c, hipMemcpyDeviceToHost, v.s), "flags download");
        }
        TCHK(wait_stream(m, g, v.s, "exchange"));
        for (int id = 1; id < v.E; id++) {                       // this shard's own agents come first in the global order
            if (conv[id]) continue;                              // an agent that has "pended" in this chunk does not send (:352-358)
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
    return batch_merge_finish(sh.batch, full_agents, g == 0);
}

extern "C" int frirl_hip_multi_train_merged(frirl_hip_multi *m, int32_t max_episodes, int32_t chunk, int32_t *episodes_run, int32_t *rounds)
{
    if (!m || chunk < 2) { set_error("frirl_hip_multi_train_merged: bad arguments"); return FRIRL_HIP_EINVAL; }
    const int G = (int)m->shards.size();
    std::vector<int> eps(G, 0), nrounds(G, 0);
    const int rc = run_shards(m, "frirl_hip_multi_train_merged", [&](int g) {
        Shard &sh = m->shards[g];
        int episode_num = 1, episodes = 0, nr = 0;          // the master's frirl_desc.episode_num; every shard counts the same way
        bool stop = false;
        while (!stop) {                                         // frirl_omp_run's loop (frirl_agent.c:319-360): every shard takes the same path
            bool master_done = false;
            for (int c = 1; c < chunk; c++) {                   // a whole chunk: max_episodes is looked at when it is over (frirl_sequential_run.c:57-63)
                const int erc = frirl_hip_batch_episode(sh.batch);
                episodes++;
                const int rrc = shard_report(m, g, erc);
                if (rrc) { if (!sh.rc) shard_failed(m, g, rrc, "report", episodes); stop = true; break; }
                master_done = sh.h_stat[7] > 0.0;
                if (master_done) break;
                episode_num++;
            }
            if (stop || master_done || !(episode_num < max_episodes)) break;
            const int mrc = merge_round_shard(m, g, nullptr);
            if (mrc) { shard_failed(m, g, mrc, "merge round", episodes); break; }
            nr++;
        }
        int ep = episodes + 1;
        eps[g] = ep - 1;
        nrounds[g] = nr;
    });
    if (rc) return rc;
    m->episodes = eps[0];
    stats_from(m->shards[0].h_stat, &m->last);
    if (episodes_run) *episodes_run = eps[0];
    if (rounds) *rounds = nrounds[0];
    return FRIRL_HIP_OK;
}

extern "C" int frirl_hip_multi_stats(frirl_hip_multi *m, frirl_hip_batch_stats_t *out)
{
    if (!m || !out) { set_error("frirl_hip_multi_stats: NULL"); return FRIRL_HIP_EINVAL; }
    const int rc = run_shards(m, "frirl_hip_multi_stats", [&](int g) {
        const int rrc = shard_report(m, g, 0);
        if (rrc && !m->shards[g].rc) shard_failed(m, g, rrc, "report", 0);
    });
    if (rc) return rc;
    stats_from(m->shards[0].h_stat, out);
    m->last = *out;
    return FRIRL_HIP_OK;
}

extern "C" int frirl_hip_multi_info(const frirl_hip_multi *m, int32_t *ngpus, int32_t *rccl_version, int64_t *shard_start, int64_t *shard_count)
{
    if (!m) { set_error("frirl_hip_multi_info: NULL"); return FRIRL_HIP_EINVAL; }
    if (ngpus) *ngpus = (int32_t)m->shards.size();
    if (rccl_version) *rccl_version = m->loopback ? -1 : m->rccl_version;          // -1: loop-back transport, logical shards on one device
    for (size_t g = 0; g < m->shards.size(); g++) {
        if (shard_start) shard_start[g] = m->shards[g].start;
        if (shard_count) shard_count[g] = m->shards[g].count;
    }
    return FRIRL_HIP_OK;
}

// rule base of the agent with GLOBAL id `agent`: routed to the shard that owns it
extern "C" int frirl_hip_multi_get_rulebase(frirl_hip_multi *m, int64_t agent, int32_t *R, double *rant, double *rconc)
{
    if (!m || agent < 0 || agent >= m->total) { set_error("frirl_hip_multi_get_rulebase: bad agent id"); return FRIRL_HIP_EINVAL; }
    DeviceGuard keep;
    for (Shard &sh : m->shards)
        if (agent >= sh.start && agent < sh.start + sh.count) {
            if (hipSetDevice(sh.device) != hipSuccess) { set_error("frirl_hip_multi_get_rulebase: hipSetDevice failed"); return FRIRL_HIP_ENODEV; }
            return frirl_hip_batch_get_rulebase(sh.batch, (int32_t)(agent - sh.start), R, rant, rconc);
        }
    set_error("frirl_hip_multi_get_rulebase: agent not found");
    return FRIRL_HIP_EINVAL;
}
