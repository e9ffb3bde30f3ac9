// five_rule_distance.hip -- batched rule-distance scan + first exact hit.
//
// Replaces reference src/five/five_rule_distance.c:63-295 (AVX2 kernels K1 :82-100 and K2
// :160-236): the reference writes per-dimension squared differences to a 64 B/rule scratch and
// re-reads them; here the two passes are fused in registers, so the only HBM traffic is the
// compulsory one: the antecedents once (+ 8 B written when the distances are materialised).
//
// HBM-bound stream with no reuse => no MFMA, no LDS tiling of the rule data.  What decides the rate is the ORDER in
// which the chip walks memory (profiles/r02_rule_distance_order.md): a traffic-only kernel with the same loads and
// stores reaches 0.59-0.70 of the 8 TB/s peak when concurrently running workgroups belong to different environments
// (thousands of scattered 4 KiB streams: round 1's grid = (environment, chunk)) and 0.74-0.80 when they sweep the
// chunks of a few consecutive environments together (compact DRAM windows).  So:
//   work item = (environment e, chunk c) of `chunk` consecutive rules, numbered item = e * cpe + c  (c fastest);
//   every lane loads two rules per column and instruction (16 B of f64 / 4 B of u16 indices): a wave instruction reads
//   1 KiB / 256 contiguous bytes and stores 1 KiB of contiguous distances; wider per-lane index loads (8 / 16 B) leave
//   the stores only half / quarter dense and measured 0.42 / 0.19.
// Three kernels, the same arithmetic on the same doubles (bit-identical results):
//   rule_distance_kernel          f64 VE columns (the reference's layout); one workgroup per item, hardware dispatch
//                                 order = item order;
//   rule_distance_idx_kernel      16-bit universe-index mirror, VE values gathered from an LDS copy of the tables;
//                                 one workgroup per item -- for SMALL tables (<= 4 KiB: the copy is refilled by every
//                                 workgroup from L2, cheaper than anything persistent: 0.73-0.79 moved);
//   rule_distance_idx_persist     LARGE tables (cartpole 40 KB, cfg5 125 KB): persistent workgroups fill the table ONCE
//                                 and take items from an in-order counter, four consecutive items per atomic (one per
//                                 item serialises on the atomic: 0.37), so the resident workgroups still advance as one
//                                 compact window (static striding lets them drift apart: 0.65); the next item's
//                                 indices and scalars are fetched before the current item is computed; the observation
//                                 VE values of every environment are computed once by observe_reset_kernel (which also
//                                 resets the hit words and the counter): no barrier, no LDS write and no dependent
//                                 global chain per item.  0.74-0.79 moved at cfg3 (round 1: 0.55), 0.69 at cfg5 (0.56).
// First exact hit: per-lane minimum index -> wave butterfly -> (LDS ->) one integer atomicMin per workgroup / item
// (deterministic; only taken when a hit exists).
// "rd_*" / "no_uidx" options (frirl_hip_set_option) are experiment hooks (tools/ab_rd.py); unset, the shipped configuration runs.
#include <stdlib.h>

#include "device_common.h"

namespace frirl {

// item -> (environment, first rule): chunk index fastest (shipped) or environment fastest (round-1 order, A/B only)
__device__ __forceinline__ void item_to_env_chunk(unsigned item, int cpe, int E, bool env_fastest, int &e, int &c)
{
    if (env_fastest) { c = (int)(item / (unsigned)E); e = (int)(item - (unsigned)c * (unsigned)E); }
    else { e = (int)(item / (unsigned)cpe); c = (int)(item - (unsigned)e * (unsigned)cpe); }
}

template <int NANT, bool WRITE, int UNROLL, int NT>
__global__ __launch_bounds__(FRIRL_BLOCK) void rule_distance_kernel(
    const double *__restrict__ u, const double *__restrict__ ve, int U, const double *__restrict__ rb,
    const int32_t *__restrict__ nrules, int maxR, const double *__restrict__ x, double *__restrict__ dists,
    uint32_t *__restrict__ hit, int rules_per_block, int cpe, int E, int env_fastest)
{
    int e, c;
    item_to_env_chunk(blockIdx.x, cpe, E, env_fastest != 0, e, c);
    const int R = nrules[e];
    const int r0 = c * rules_per_block;
    if (r0 >= R) return;   // uniform for the workgroup
    int r_end = r0 + rules_per_block;
    if (r_end > R) r_end = R;

    __shared__ double q_s[NANT];
    __shared__ unsigned red_s[FRIRL_WAVES_PER_BLOCK];
    if (threadIdx.x < NANT) q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, x[(size_t)e * NANT + threadIdx.x]);
    __syncthreads();
    double q[NANT];
#pragma unroll
    for (int k = 0; k < NANT; k++) q[k] = q_s[k];

    const double *__restrict__ base = rb + (size_t)e * (NANT + 1) * maxR;
    double *__restrict__ out = WRITE ? dists + (size_t)e * maxR : nullptr;
    unsigned best = FRIRL_HIP_NO_HIT;
    constexpr int STEP = FRIRL_BLOCK * 2;

    for (int r = r0 + 2 * (int)threadIdx.x; r < r_end; r += STEP * UNROLL) {
        double2 v[UNROLL][NANT];
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int rr = r + j * STEP;
            if (rr < r_end) {
#pragma unroll
                for (int k = 0; k < NANT; k++) {
                    const double2 *src = reinterpret_cast<const double2 *>(base + (size_t)k * maxR + rr);
                    if (NT == 1 || NT == 2) { v[j][k].x = __builtin_nontemporal_load(&src->x); v[j][k].y = __builtin_nontemporal_load(&src->y); }
                    else v[j][k] = *src;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int rr = r + j * STEP;
            if (rr < r_end) {
                // dimension-ordered sum of squares, separate mul / add (five_rule_distance.c:88-90,171-208)
                double d0 = q[0] - v[j][0].x, d1 = q[0] - v[j][0].y;
                double a0 = d0 * d0, a1 = d1 * d1;
#pragma unroll
                for (int k = 1; k < NANT; k++) {
                    d0 = q[k] - v[j][k].x;
                    d1 = q[k] - v[j][k].y;
                    const double s0 = d0 * d0, s1 = d1 * d1;
                    a0 = a0 + s0;
                    a1 = a1 + s1;
                }
                double2 d;
                d.x = __dsqrt_rn(a0);   // IEEE sqrt (vsqrtpd, five_rule_distance.c:211)
                d.y = __dsqrt_rn(a1);
                if (WRITE) {                                              // rows are even-sized: rr+1 < maxR
                    if (NT == 1 || NT == 3) { __builtin_nontemporal_store(d.x, out + rr); __builtin_nontemporal_store(d.y, out + rr + 1); }
                    else *reinterpret_cast<double2 *>(out + rr) = d;
                }
                // first exact hit among valid rules (five_rule_distance.c:215-217,241-262)
                if (d.y == 0.0 && rr + 1 < R) best = min(best, (unsigned)(rr + 1));
                if (d.x == 0.0) best = min(best, (unsigned)rr);
            }
        }
    }
    best = block_min_u32(best, red_s);
    if (threadIdx.x == 0 && best != FRIRL_HIP_NO_HIT) atomicMin(&hit[e], best);
}

// Distances of two adjacent rules from their packed 16-bit universe indices (w[k] = rule r | rule r+1 << 16) and the
// LDS copy of the VE tables: the same subtract / multiply / add sequence as above on the same doubles.
template <int NANT>
__device__ __forceinline__ double2 idx_pair_distance(const uint32_t (&w)[NANT], const double (&q)[NANT], const double *__restrict__ tab_s, int U)
{
    double2 t = lds_table_pair(tab_s, w[0]);                  // one v_mad_u32_u16 per index (device_common.h)
    double d0 = q[0] - t.x, d1 = q[0] - t.y;
    double a0 = d0 * d0, a1 = d1 * d1;
#pragma unroll
    for (int k = 1; k < NANT; k++) {
        t = lds_table_pair(tab_s + k * U, w[k]);
        d0 = q[k] - t.x;
        d1 = q[k] - t.y;
        const double s0 = d0 * d0, s1 = d1 * d1;
        a0 = a0 + s0;
        a1 = a1 + s1;
    }
    double2 d;
    d.x = __dsqrt_rn(a0);
    d.y = __dsqrt_rn(a1);
    return d;
}

// Compressed-antecedent form, small tables: one workgroup per item, the table copy refilled per workgroup (<= 4 KiB from L2).
template <int NANT, bool WRITE, int UNROLL, int BLOCK = FRIRL_BLOCK>
__global__ __launch_bounds__(BLOCK) void rule_distance_idx_kernel(
    const double *__restrict__ u, const double *__restrict__ ve, int U, const uint16_t *__restrict__ uidx,
    const int32_t *__restrict__ nrules, int maxR, const double *__restrict__ x, double *__restrict__ dists,
    uint32_t *__restrict__ hit, int rules_per_block, int cpe, int E, int env_fastest)
{
    extern __shared__ double tab_s[];            // [NANT][U] vague environments
    int e, c;
    item_to_env_chunk(blockIdx.x, cpe, E, env_fastest != 0, e, c);
    const int R = nrules[e];
    const int r0 = c * rules_per_block;
    if (r0 >= R) return;
    int r_end = r0 + rules_per_block;
    if (r_end > R) r_end = R;

    __shared__ double q_s[NANT];
    __shared__ unsigned red_s[BLOCK / FRIRL_WAVE];
    for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    if (threadIdx.x < NANT) q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, x[(size_t)e * NANT + threadIdx.x]);
    __syncthreads();
    double q[NANT];
#pragma unroll
    for (int k = 0; k < NANT; k++) q[k] = q_s[k];

    const uint16_t *__restrict__ base = uidx + (size_t)e * NANT * maxR;
    double *__restrict__ out = WRITE ? dists + (size_t)e * maxR : nullptr;
    unsigned best = FRIRL_HIP_NO_HIT;
    constexpr int STEP = BLOCK * 2;

    for (int r = r0 + 2 * (int)threadIdx.x; r < r_end; r += STEP * UNROLL) {
        uint32_t w[UNROLL][NANT];
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int rr = r + j * STEP;
            if (rr < r_end) {
#pragma unroll
                for (int k = 0; k < NANT; k++) w[j][k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(base + (size_t)k * maxR + rr));
            }
        }
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int rr = r + j * STEP;
            if (rr < r_end) {
                const double2 d = idx_pair_distance<NANT>(w[j], q, tab_s, U);
                if (WRITE) { __builtin_nontemporal_store(d.x, out + rr); __builtin_nontemporal_store(d.y, out + rr + 1); }
                if (d.y == 0.0 && rr + 1 < R) best = min(best, (unsigned)(rr + 1));
                if (d.x == 0.0) best = min(best, (unsigned)rr);
            }
        }
    }
    best = wave_min_u32(best);
    if ((threadIdx.x & (FRIRL_WAVE - 1)) == 0) red_s[threadIdx.x / FRIRL_WAVE] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = red_s[0];
        for (int w = 1; w < BLOCK / FRIRL_WAVE; w++) m = red_s[w] < m ? red_s[w] : m;
        if (m != FRIRL_HIP_NO_HIT) atomicMin(&hit[e], m);
    }
}

// Prologue of the persistent form: qv[e][k] = ve[k][snap(x[e][k])] (five_rule_distance.c:75,80) for every environment,
// hit[e] = "none", item counter = 0.
__global__ void observe_reset_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U, int nant, int E, const double *__restrict__ x,
                                     double *__restrict__ qv, uint32_t *__restrict__ hit, unsigned *__restrict__ counter)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *counter = 0u;
    if (i >= E * nant) return;
    const int e = i / nant, k = i - e * nant;
    qv[i] = observe_ve(u, ve, U, k, x[i]);
    if (k == 0) hit[e] = FRIRL_HIP_NO_HIT;
}

// Compressed-antecedent form, large tables: persistent workgroups (see the header of this file).  `counter` hands out
// batches of KB consecutive items in order; every workgroup leaves when the counter passes the last batch.
template <int NANT, bool WRITE, int UNROLL, int BLOCK, int KB>
__global__ __launch_bounds__(BLOCK) void rule_distance_idx_persist_kernel(
    const double *__restrict__ ve, int U, const uint16_t *__restrict__ uidx, const int32_t *__restrict__ nrules, int maxR,
    const double *__restrict__ qv, double *__restrict__ dists, uint32_t *__restrict__ hit, int cpe, int nitems, unsigned *__restrict__ counter)
{
    extern __shared__ double tab_s[];            // [NANT][U] vague environments
    __shared__ int batch_s[2];
    constexpr int STEP = BLOCK * 2, CH = STEP * UNROLL;
    const int nbatches = (nitems + KB - 1) / KB;
    uint32_t w[UNROLL][NANT];                    // indices of the NEXT item (in flight while the current one is computed)
    double qn[NANT];
    int en = 0, cn = 0, Rn = 0;
    auto load_item = [&](int it) {
        en = it / cpe; cn = it - en * cpe;
        Rn = nrules[en];
        const uint16_t *__restrict__ base = uidx + (size_t)en * NANT * maxR;
        const int r = cn * CH + 2 * (int)threadIdx.x;
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int rr = r + j * STEP;
            if (rr < Rn) {
#pragma unroll
                for (int k = 0; k < NANT; k++) w[j][k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(base + (size_t)k * maxR + rr));
            }
        }
#pragma unroll
        for (int k = 0; k < NANT; k++) qn[k] = qv[(size_t)en * NANT + k];      // uniform: scalar loads
    };
    int slot = 0, sub = 0;
    if (threadIdx.x == 0) batch_s[0] = (int)atomicAdd(counter, 1u);
    __syncthreads();
    int batch = batch_s[0];
    int item = batch < nbatches ? batch * KB : nitems;
    if (item < nitems) load_item(item);          // first loads are in flight while the table is filled
    for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    __syncthreads();
    while (item < nitems) {                      // uniform for the workgroup: every wave sees the same item sequence
        const int e = en, c = cn, R = Rn;
        uint32_t cw[UNROLL][NANT];
        double q[NANT];
#pragma unroll
        for (int j = 0; j < UNROLL; j++)
#pragma unroll
            for (int k = 0; k < NANT; k++) cw[j][k] = w[j][k];
#pragma unroll
        for (int k = 0; k < NANT; k++) q[k] = qn[k];
        int nxt;
        sub++;
        if (sub < KB && item + 1 < nitems) nxt = item + 1;
        else {                                   // next batch; the double-buffered slot needs one barrier per fetch
            sub = 0;
            slot ^= 1;
            if (threadIdx.x == 0) batch_s[slot] = (int)atomicAdd(counter, 1u);
            __syncthreads();
            batch = batch_s[slot];
            nxt = batch < nbatches ? batch * KB : nitems;
        }
        if (nxt < nitems) load_item(nxt);
        double *__restrict__ out = WRITE ? dists + (size_t)e * maxR : nullptr;
        const int r = c * CH + 2 * (int)threadIdx.x;
        unsigned best = FRIRL_HIP_NO_HIT;
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int rr = r + j * STEP;
            if (rr < R) {
                const double2 d = idx_pair_distance<NANT>(cw[j], q, tab_s, U);
                if (WRITE) { __builtin_nontemporal_store(d.x, out + rr); __builtin_nontemporal_store(d.y, out + rr + 1); }
                if (d.y == 0.0 && rr + 1 < R) best = min(best, (unsigned)(rr + 1));
                if (d.x == 0.0) best = min(best, (unsigned)rr);
            }
        }
        if (best != FRIRL_HIP_NO_HIT) atomicMin(&hit[e], best);      // rare; integer min: order-independent
        item = nxt;
    }
}

struct RdTune { int unroll, chunk, nt; };
static RdTune rd_tune()
{
    const frirl_host::Options &o = frirl_host::opts();     // read once from the environment / frirl_hip_set_option, not per launch
    RdTune t;
    t.unroll = o.rd_unroll;          // 0 = shipped default
    t.chunk = o.rd_chunk;            // 0 = shipped default
    t.nt = o.rd_nt;                  // -1 = shipped default (non-temporal loads and stores)
    return t;
}

struct RdGrid { unsigned items; int rules_per_block, cpe, env_fastest; };

template <int NANT, int UNROLL, int NT>
static void launch_variant(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *x, double *ruledists, uint32_t *hit,
                           hipStream_t s, const RdGrid &g)
{
    if (ruledists)
        hipLaunchKernelGGL((rule_distance_kernel<NANT, true, UNROLL, NT>), dim3(g.items), dim3(FRIRL_BLOCK), 0, s, t->u, t->ve, t->U, b->rb, b->nrules,
                           b->maxR, x, ruledists, hit, g.rules_per_block, g.cpe, b->E, g.env_fastest);
    else
        hipLaunchKernelGGL((rule_distance_kernel<NANT, false, UNROLL, NT>), dim3(g.items), dim3(FRIRL_BLOCK), 0, s, t->u, t->ve, t->U, b->rb, b->nrules,
                           b->maxR, x, ruledists, hit, g.rules_per_block, g.cpe, b->E, g.env_fastest);
}

// one workgroup per item; chunk = multiple of one 512-rule sweep of the workgroup
static bool make_grid(const frirl_hip_rulebases *b, int rules_per_block, RdGrid &g)
{
    rules_per_block = ((rules_per_block + 2 * FRIRL_BLOCK - 1) / (2 * FRIRL_BLOCK)) * (2 * FRIRL_BLOCK);
    const long cpe = ((long)b->maxR + rules_per_block - 1) / rules_per_block;
    const long items = cpe * (long)b->E;
    if (items > 0x7FFFFFFFL) return false;
    g.items = (unsigned)items; g.rules_per_block = rules_per_block; g.cpe = (int)cpe; g.env_fastest = frirl_host::opts().rd_order == 1;
    return true;
}

static int device_cus()
{
    static thread_local int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return cus;
}

// Persistent form: scratch = [counter | qv[E][NANT]] taken from the stream's memory pool (stream-ordered: concurrent calls on
// other streams never share it) and returned to it right after the launch.
template <int NANT, int UNROLL, int BLOCK, int KB>
static int launch_persist(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *x, double *ruledists, uint32_t *hit, hipStream_t s,
                          size_t tab_bytes, int wg_per_cu)
{
    using namespace frirl_host;
    constexpr int CH = BLOCK * 2 * UNROLL;
    const long cpe = ((long)b->maxR + CH - 1) / CH;
    const long nitems = cpe * (long)b->E;
    if (nitems > 0x7FFFFFF0L) { set_error("five_hip_rule_distance: %ld work items exceed the 31-bit item index", nitems); return FRIRL_HIP_EINVAL; }
    const long nbatches = (nitems + KB - 1) / KB;
    long grid = (long)device_cus() * wg_per_cu;
    if (grid > nbatches) grid = nbatches;
    const size_t qv_off = 256, bytes = qv_off + sizeof(double) * (size_t)b->E * NANT;
    void *scratch = nullptr;
    hipError_t e1 = hipMallocAsync(&scratch, bytes, s);
    if (e1 != hipSuccess) { (void)hipGetLastError(); set_error("five_hip_rule_distance: hipMallocAsync(%zu B) failed: %s", bytes, hipGetErrorString(e1)); return FRIRL_HIP_ELAUNCH; }
    unsigned *counter = static_cast<unsigned *>(scratch);
    double *qv = reinterpret_cast<double *>(static_cast<char *>(scratch) + qv_off);
    const int nq = b->E * NANT;
    hipLaunchKernelGGL(observe_reset_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, t->u, t->ve, t->U, NANT, b->E, x, qv, hit, counter);
    if (ruledists) {
        auto k = rule_distance_idx_persist_kernel<NANT, true, UNROLL, BLOCK, KB>;
        e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes);
        if (e1 == hipSuccess) hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(BLOCK), tab_bytes, s, t->ve, t->U, b->uidx, b->nrules, b->maxR, qv, ruledists, hit, (int)cpe, (int)nitems, counter);
    } else {
        auto k = rule_distance_idx_persist_kernel<NANT, false, UNROLL, BLOCK, KB>;
        e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes);
        if (e1 == hipSuccess) hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(BLOCK), tab_bytes, s, t->ve, t->U, b->uidx, b->nrules, b->maxR, qv, ruledists, hit, (int)cpe, (int)nitems, counter);
    }
    const hipError_t e2 = hipFreeAsync(scratch, s);
    if (e1 != hipSuccess) { set_error("five_hip_rule_distance: cannot reserve %zu B of LDS: %s", tab_bytes, hipGetErrorString(e1)); return FRIRL_HIP_ELAUNCH; }
    if (e2 != hipSuccess) { set_error("five_hip_rule_distance: hipFreeAsync failed: %s", hipGetErrorString(e2)); return FRIRL_HIP_ELAUNCH; }
    return check_launch("five_hip_rule_distance(uidx, persistent)");
}

// Shipped configuration (A/B-measured on MI355X, tools/ab_rd.py, tools/exp/rd_bench.hip; profiles/r01_rule_distance_tuning.md,
// profiles/r02_rule_distance_order.md): non-temporal loads AND stores (pure stream, nothing is re-read: +5..8 %), 8 independent
// column sets per lane for nant <= 5 (all loads issued before the first use), 4 for nant <= 8, 2 above.
template <int NANT>
struct RdConfig {
    static constexpr int UNROLL = (NANT <= 5) ? 8 : (NANT <= 8 ? 4 : 2);
};

// tables up to this size use one workgroup per item with a per-workgroup table copy; larger ones the persistent form
static constexpr size_t RD_SMALL_TABLE_BYTES = 4 * 1024;

template <int NANT>
static int launch_nant(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *x, double *ruledists,
                       uint32_t *hit, hipStream_t s)
{
    using namespace frirl_host;
    constexpr int UNROLL = RdConfig<NANT>::UNROLL;
    const size_t tab_bytes = sizeof(double) * NANT * (size_t)t->U;
    const RdTune tn = rd_tune();
    const bool idx = b->uidx && t->U <= 65536 && tab_bytes <= 150 * 1024 && !opts().no_uidx;
    const int persist = opts().rd_persist;       // -1 = by table size
    if (idx && (persist == 1 || (persist != 0 && tab_bytes > RD_SMALL_TABLE_BYTES))) {
        // 2 workgroups of 512 threads per CU while two table copies fit beside each other (<= 64 KiB each), else one of 1024
        if (tab_bytes <= 64 * 1024) return launch_persist<NANT, (NANT <= 8 ? 2 : 1), 512, 4>(t, b, x, ruledists, hit, s, tab_bytes, 2);
        return launch_persist<NANT, 1, 1024, 4>(t, b, x, ruledists, hit, s, tab_bytes, 1);
    }
    if (hipMemsetAsync(hit, 0xFF, sizeof(uint32_t) * (size_t)b->E, s) != hipSuccess) return check_launch("five_hip_rule_distance(memset)");
    RdGrid g;
    if (idx) {
        if (tab_bytes > 64 * 1024) { set_error("five_hip_rule_distance: option rd_persist=0 needs VE tables <= 64 KiB"); return FRIRL_HIP_EINVAL; }
        // chunk = ONE sweep of the workgroup (256 threads x 2 rules x UI column sets = 2048 rules for nant <= 8): smaller items keep
        // the window of concurrently streamed memory compact
        constexpr int UI = (NANT <= 8) ? 4 : 2;
        const int un = (NANT <= 5 && tn.unroll) ? tn.unroll : UI;      // tuning hook (experiments only)
        if (!make_grid(b, tn.chunk > 0 ? tn.chunk : 2 * FRIRL_BLOCK * UI, g)) { set_error("five_hip_rule_distance: too many work items"); return FRIRL_HIP_EINVAL; }
        hipError_t e1 = hipSuccess;
#define VI(U_)                                                                                                                               \
    do {                                                                                                                                     \
        if (ruledists) {                                                                                                                     \
            auto k = rule_distance_idx_kernel<NANT, true, U_>;                                                                               \
            if (tab_bytes > 48 * 1024) e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes); \
            if (e1 == hipSuccess) hipLaunchKernelGGL(k, dim3(g.items), dim3(FRIRL_BLOCK), tab_bytes, s, t->u, t->ve, t->U, b->uidx, b->nrules, b->maxR, x, ruledists, hit, g.rules_per_block, g.cpe, b->E, g.env_fastest); \
        } else {                                                                                                                             \
            auto k = rule_distance_idx_kernel<NANT, false, U_>;                                                                              \
            if (tab_bytes > 48 * 1024) e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes); \
            if (e1 == hipSuccess) hipLaunchKernelGGL(k, dim3(g.items), dim3(FRIRL_BLOCK), tab_bytes, s, t->u, t->ve, t->U, b->uidx, b->nrules, b->maxR, x, ruledists, hit, g.rules_per_block, g.cpe, b->E, g.env_fastest); \
        }                                                                                                                                    \
    } while (0)
        if (NANT <= 5 && un == 8) VI(8);
        else if (NANT <= 5 && un == 2) VI(2);
        else if (NANT <= 5 && un == 1) VI(1);
        else VI(UI);
#undef VI
        if (e1 != hipSuccess) { set_error("five_hip_rule_distance: cannot reserve %zu B of LDS: %s", tab_bytes, hipGetErrorString(e1)); return FRIRL_HIP_ELAUNCH; }
        return check_launch("five_hip_rule_distance(uidx)");
    }
    // f64 columns: 1024-rule items for small rule bases, 2048 above (A/B at cfg2: 1024 -> 6.53 TB/s, 2048 -> 6.10, 4096 -> 5.93;
    // at cfg4: 1024 -> 5.61, 2048 -> 6.20, 4096 -> 6.00, 8192 -> 5.96; tools/ab_rd.py with AB_F64=1)
    if (!make_grid(b, tn.chunk > 0 ? tn.chunk : (b->maxR <= 16384 + 512 ? 1024 : 2048), g)) { set_error("five_hip_rule_distance: too many work items"); return FRIRL_HIP_EINVAL; }
    if (NANT <= 5 && (tn.unroll || tn.nt >= 0)) {      // tuning hooks (experiments only)
        const int un = tn.unroll ? tn.unroll : UNROLL;
        const int nt = tn.nt >= 0 ? tn.nt : 1;
#define V(U_, N_) launch_variant<NANT, U_, N_>(t, b, x, ruledists, hit, s, g)
        if (nt == 1) { if (un == 1) V(1, 1); else if (un == 2) V(2, 1); else if (un == 8) V(8, 1); else V(4, 1); }
        else if (nt == 2) { if (un == 1) V(1, 2); else if (un == 2) V(2, 2); else if (un == 8) V(8, 2); else V(4, 2); }
        else if (nt == 3) { if (un == 1) V(1, 3); else if (un == 2) V(2, 3); else if (un == 8) V(8, 3); else V(4, 3); }
        else { if (un == 1) V(1, 0); else if (un == 2) V(2, 0); else if (un == 8) V(8, 0); else V(4, 0); }
#undef V
    } else {
        launch_variant<NANT, UNROLL, 1>(t, b, x, ruledists, hit, s, g);
    }
    return check_launch("five_hip_rule_distance");
}

}  // namespace frirl

extern "C" int five_hip_rule_distance_uses_uidx(int32_t nant, int32_t U)
{
    return nant >= 1 && nant <= FRIRL_HIP_MAX_NANT && U <= 65536 && sizeof(double) * nant * (size_t)U <= 150 * 1024 && !frirl_host::opts().no_uidx;
}

extern "C" int five_hip_rule_distance(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *x,
                                      double *ruledists, uint32_t *hit, void *stream)
{
    using namespace frirl_host;
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if (!x || !hit) { set_error("five_hip_rule_distance: NULL x/hit"); return FRIRL_HIP_EINVAL; }
    if (ruledists && (reinterpret_cast<uintptr_t>(ruledists) & 15)) { set_error("five_hip_rule_distance: ruledists must be 16-byte aligned"); return FRIRL_HIP_EINVAL; }
    if ((rc = check_device())) return rc;
    hipStream_t s = as_stream(stream);

    switch (t->nant) {
#define FRIRL_CASE(N) case N: return frirl::launch_nant<N>(t, b, x, ruledists, hit, s);
        FRIRL_CASE(1) FRIRL_CASE(2) FRIRL_CASE(3) FRIRL_CASE(4) FRIRL_CASE(5) FRIRL_CASE(6) FRIRL_CASE(7) FRIRL_CASE(8)
        FRIRL_CASE(9) FRIRL_CASE(10) FRIRL_CASE(11) FRIRL_CASE(12) FRIRL_CASE(13) FRIRL_CASE(14) FRIRL_CASE(15) FRIRL_CASE(16)
#undef FRIRL_CASE
    }
    set_error("five_hip_rule_distance: unsupported nant=%d", t->nant);
    return FRIRL_HIP_EINVAL;
}
