// lanes.hip -- frirl_episode's loop for MANY agents with SMALL rule bases (the demos' real learning regime, <= a few
// hundred rules per agent): a group of G consecutive lanes owns one environment, 64/G environments per wave, the whole
// episode in one launch.
//
// Why another mapping: with one wave (or workgroup) per environment a 100-rule base gives every lane 1-2 rules and the
// step is all reductions, barriers and single-lane sections.  Here nothing is reduced: the lanes of a group split the
// A + 1 conclusions a step needs -- Q(s',a) for each action a and Q(s,a) of the pending update -- and each lane walks ALL
// rules in index order, accumulating its Shepard sums sequentially (the reference's own summation order,
// FIVEVagConcl.c:224-235).  The only cross-lane traffic is a handful of __shfl's per step inside the group.
//
// Layout: the rule bases are transposed so that at a given rule the 64/G environments of a wave read one contiguous run
// (the G lanes of a group read the same address); tile = wave, i = environment within the tile:
//   f64 store   T[tile][k][r][i]            doubles, k = 0..nant (VE columns + consequents): 8(nant+1) B per rule
//   index store Ti[tile][r][i] + Tq[tile][r][i]  one packed record of nant 16-bit universe indices (8 or 16 B) + the
//               consequent; VE values are gathered from an LDS copy of the tables (rb[e][k][r] == ve[k][uidx[e][k][r]]
//               exactly, five_add_rule.c:76-81): 16 (nant <= 4) or 24 B per rule and 2 loads instead of nant+1
// With few agents the groups alone cannot fill the chip (8 192 agents x 4 lanes = 512 waves on 1024 SIMDs, and a wave's
// step is a latency chain): H = 2 / 4 lanes then share each conclusion, lane h summing the rules r = h (mod H) in order;
// the H partial sums are added in slice order and the lowest exact hit wins -- a fixed two-level order instead of the
// reference's single chain (as the per-environment kernels' trees), chosen only when waves are scarce.
// frirl_hip_episode_run_lanes imports the canonical slabs, runs, and exports them again (two small transposes per call:
// only the first nrules[e] rules move).
#include "sweeps.h"
#include "envs.h"
#include <cstdlib>
#include <type_traits>

namespace frirl {

constexpr int LN_WPB = 4;                         // waves (tiles) per workgroup: they share the LDS tables
constexpr int LN_BLOCK = LN_WPB * FRIRL_WAVE;

// ---- import / export: canonical rb[e][k][r] (+ uidx[e][k][r]) <-> tiles, rules r < nrules[e] ---------------------------
__global__ __launch_bounds__(256) void lanes_import_kernel(const double *__restrict__ rb, const int32_t *__restrict__ nrules, int E, int nant1, int maxR,
                                                            int EPW, double *__restrict__ T)
{
    __shared__ double s[16][65];
    const int tile = blockIdx.x, k = blockIdx.y, e0 = tile * EPW;
    int rmax = 0;
    for (int i = 0; i < EPW; i++) if (e0 + i < E) { const int r = nrules[e0 + i]; rmax = r > rmax ? r : rmax; }
    for (int r0 = 0; r0 < rmax; r0 += 64) {
        for (int idx = threadIdx.x; idx < EPW * 64; idx += 256) {
            const int i = idx >> 6, j = idx & 63, e = e0 + i, r = r0 + j;
            s[i][j] = (e < E && r < maxR) ? rb[((size_t)e * nant1 + k) * maxR + r] : 0.0;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < EPW * 64; idx += 256) {
            const int j = idx / EPW, i = idx - j * EPW, r = r0 + j;
            if (r < maxR) T[(((size_t)tile * nant1 + k) * maxR + r) * EPW + i] = s[i][j];
        }
        __syncthreads();
    }
}

// k = 0..nant1-1 selects the column exported from the f64 tiles (kfirst: first column to export)
__global__ __launch_bounds__(256) void lanes_export_kernel(double *__restrict__ rb, const int32_t *__restrict__ nrules, int E, int nant1, int maxR, int EPW,
                                                            const double *__restrict__ T, int tile_cols, int kfirst)
{
    __shared__ double s[16][65];
    __shared__ int nr[16];
    const int tile = blockIdx.x, kt = blockIdx.y, k = kfirst + kt, e0 = tile * EPW;
    if ((int)threadIdx.x < EPW) nr[threadIdx.x] = (e0 + (int)threadIdx.x < E) ? nrules[e0 + threadIdx.x] : 0;
    __syncthreads();
    int rmax = 0;
    for (int i = 0; i < EPW; i++) rmax = nr[i] > rmax ? nr[i] : rmax;
    for (int r0 = 0; r0 < rmax; r0 += 64) {
        for (int idx = threadIdx.x; idx < EPW * 64; idx += 256) {
            const int j = idx / EPW, i = idx - j * EPW, r = r0 + j;
            s[i][j] = (r < maxR) ? T[(((size_t)tile * tile_cols + kt) * maxR + r) * EPW + i] : 0.0;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < EPW * 64; idx += 256) {
            const int i = idx >> 6, j = idx & 63, e = e0 + i, r = r0 + j;
            if (e < E && r < nr[i]) rb[((size_t)e * nant1 + k) * maxR + r] = s[i][j];
        }
        __syncthreads();
    }
}

// index store: packed records (W 32-bit words, two 16-bit indices each) + consequents
__global__ __launch_bounds__(256) void lanes_import_idx_kernel(const double *__restrict__ rb, const uint16_t *__restrict__ uidx,
                                                                const int32_t *__restrict__ nrules, int E, int nant, int maxR, int EPW, int W,
                                                                uint32_t *__restrict__ Ti, double *__restrict__ Tq)
{
    __shared__ uint32_t sw[16][65][4];
    __shared__ double sq[16][65];
    const int tile = blockIdx.x, e0 = tile * EPW;
    int rmax = 0;
    for (int i = 0; i < EPW; i++) if (e0 + i < E) { const int r = nrules[e0 + i]; rmax = r > rmax ? r : rmax; }
    for (int r0 = 0; r0 < rmax; r0 += 64) {
        for (int idx = threadIdx.x; idx < EPW * 64; idx += 256) {
            const int i = idx >> 6, j = idx & 63, e = e0 + i, r = r0 + j;
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            double q = 0.0;
            if (e < E && r < maxR) {
                for (int k = 0; k < nant; k++) w[k >> 1] |= (uint32_t)uidx[((size_t)e * nant + k) * maxR + r] << (16 * (k & 1));
                q = rb[((size_t)e * (nant + 1) + nant) * maxR + r];
            }
            for (int x = 0; x < 4; x++) sw[i][j][x] = w[x];
            sq[i][j] = q;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < EPW * 64; idx += 256) {
            const int j = idx / EPW, i = idx - j * EPW, r = r0 + j;
            if (r < maxR) {
                const size_t o = ((size_t)tile * maxR + r) * EPW + i;
                for (int x = 0; x < W; x++) Ti[o * W + x] = sw[i][j][x];
                Tq[o] = sq[i][j];
            }
        }
        __syncthreads();
    }
}

// ---- rule stores ------------------------------------------------------------------------------------------------
template <int NANT>
struct StoreF64 {
    static constexpr int UR = 4;
    double *Te;                                  // tile base + environment slot
    int maxR, EPW;
    struct Rec { double c[NANT + 1]; };
    template <bool NEEDQ>
    __device__ __forceinline__ Rec load(int r) const
    {
        Rec x;
#pragma unroll
        for (int k = 0; k < NANT; k++) x.c[k] = Te[(unsigned)((k * maxR + r) * EPW)];
        x.c[NANT] = NEEDQ ? Te[(unsigned)((NANT * maxR + r) * EPW)] : 0.0;
        return x;
    }
    __device__ __forceinline__ void decode(const Rec &x, double (&c)[NANT + 1]) const
    {
#pragma unroll
        for (int k = 0; k <= NANT; k++) c[k] = x.c[k];
    }
    __device__ __forceinline__ double *qptr(int r) const { return Te + (unsigned)((NANT * maxR + r) * EPW); }
    __device__ __forceinline__ void append(int R, const double *ve3, const unsigned *, double q) const
    {
#pragma unroll
        for (int k = 0; k < NANT; k++) Te[(unsigned)((k * maxR + R) * EPW)] = ve3[k];     // five_add_rule.c:80-81
        *qptr(R) = q;
    }
};

template <int NANT>
struct StoreIdx {
    static constexpr int W = NANT <= 4 ? 2 : 4, UR = 8;
    using RecI = typename std::conditional<W == 2, uint2, uint4>::type;
    RecI *Ti;                                    // tile base + environment slot
    double *Tq;
    const double *ve_s;                          // LDS copy of the VE tables [NANT][U]
    int U, EPW;
    struct Rec { RecI w; double q; };
    template <bool NEEDQ>
    __device__ __forceinline__ Rec load(int r) const
    {
        Rec x;
        x.w = Ti[(unsigned)(r * EPW)];
        x.q = NEEDQ ? Tq[(unsigned)(r * EPW)] : 0.0;
        return x;
    }
    __device__ __forceinline__ static uint32_t word(const uint2 &w, int i) { return i == 0 ? w.x : w.y; }
    __device__ __forceinline__ static uint32_t word(const uint4 &w, int i) { return i == 0 ? w.x : (i == 1 ? w.y : (i == 2 ? w.z : w.w)); }
    __device__ __forceinline__ void decode(const Rec &x, double (&c)[NANT + 1]) const
    {
#pragma unroll
        for (int k = 0; k < NANT; k++) {           // one v_mad_u32_u16 per antecedent (sweeps.h: lds_table_entry)
            if (k & 1) c[k] = lds_table_entry<1>(ve_s + k * U, word(x.w, k >> 1));
            else c[k] = lds_table_entry<0>(ve_s + k * U, word(x.w, k >> 1));
        }
        c[NANT] = x.q;
    }
    __device__ __forceinline__ double *qptr(int r) const { return Tq + (unsigned)(r * EPW); }
    __device__ __forceinline__ void append(int R, const double *, const unsigned *idx3, double q) const
    {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < NANT; k++) w[k >> 1] |= (idx3[k] & 0xffffu) << (16 * (k & 1));
        RecI x;
        if constexpr (W == 2) { x.x = w[0]; x.y = w[1]; } else { x.x = w[0]; x.y = w[1]; x.z = w[2]; x.w = w[3]; }
        Ti[(unsigned)(R * EPW)] = x;
        *qptr(R) = q;
    }
};

// The per-lane rule loops are chains of dependent global loads (0.5-1 us each, one or two waves per SIMD): rules are
// fetched UR at a time into registers, two batches in flight (the next one is requested before the current one is
// consumed).  Out-of-range slots re-read the last rule (never consumed).
// for r = first, first + stride, ... < R (in order): f(r, columns of rule r)
template <bool NEEDQ, int NANT, class STORE, class F>
__device__ __forceinline__ void for_rules(const STORE &st, int first, int stride, int R, F &&f)
{
    constexpr int UR = STORE::UR;
    typename STORE::Rec a[UR], b[UR];
    auto fetch = [&](typename STORE::Rec(&x)[UR], int r0) {
#pragma unroll
        for (int j = 0; j < UR; j++) {
            int r = r0 + j * stride;
            r = r < R ? r : R - 1;
            r = r < 0 ? 0 : r;
            x[j] = st.template load<NEEDQ>(r);
        }
    };
    auto consume = [&](const typename STORE::Rec(&x)[UR], int r0) {
#pragma unroll
        for (int j = 0; j < UR; j++) {
            const int r = r0 + j * stride;
            if (r < R) { double c[NANT + 1]; st.decode(x[j], c); f(r, c); }
        }
    };
    fetch(a, first);
    for (int r0 = first; r0 < R; r0 += 2 * UR * stride) {
        fetch(b, r0 + UR * stride);
        consume(a, r0);
        fetch(a, r0 + 2 * UR * stride);
        consume(b, r0 + UR * stride);
    }
}

struct LaneQ {              // one conclusion's raw result
    double v, w;
    unsigned hit;
};

// FIVE_vag_concl's sums for one VE point, all rules, on one lane (sequential, rule order)
template <int NANT, class STORE, class POW>
__device__ __forceinline__ LaneQ lane_sweep_q(const STORE &st, int R, const double (&q)[NANT], POW p)
{
    LaneQ o{0.0, 0.0, FRIRL_HIP_NO_HIT};
    const auto pk = pin_pow(p);
    for_rules<true, NANT>(st, 0, 1, R, [&](int r, const double (&c)[NANT + 1]) {
        const double d0 = q[0] - c[0];
        double s = d0 * d0;
#pragma unroll
        for (int k = 1; k < NANT; k++) { const double d = q[k] - c[k]; s = __fma_rn(d, d, s); }
        // an exact hit is noted with a select and poisons the two sums, which the caller ignores then (FIVEVagConcl.c:89-93): no
        // divergent branch around the weight
        o.hit = (s == 0.0 && o.hit == FRIRL_HIP_NO_HIT) ? (unsigned)r : o.hit;
        const double wi = shepard_w(s, pk);
        o.v = __fma_rn(wi, c[NANT], o.v);
        o.w = o.w + wi;
    });
    return o;
}

struct LanesArgs {
    const double *u, *ve;        // tables [nant][U]
    int U;
    void *T;                     // workspace
    size_t tq_off;               // index store: byte offset of Tq inside the workspace
    double *rb;                  // canonical slabs (index store: VE columns of appended rules are written through)
    uint16_t *uidx;
    int32_t *nrules;
    int E, maxR, tiles;
    int lds_u;                   // universes staged in LDS next to the VE tables
    int lds_ve;                  // VE tables staged in LDS (always for the index store)
};

// One wave = 64/G environments.  APL = conclusions per lane: lanes 0..G-2 hold APL actions each ((G-1)*APL >= A), lane
// G-1 holds Q(s,a).
// PN: the Shepard power is the default p = nant as a compile-time constant (PowC<NANT>: every demo and BASELINE shape); else the
// run-time power of the agent (PowU; instantiated without rule slices only)
template <int NANT, int APL, int G, int H, int WPE, bool IDX, bool PN = true>
__global__ __launch_bounds__(LN_BLOCK, WPE) void episode_run_lanes_kernel(const LanesArgs la, const frirl_hip_agent ag, const frirl_hip_envs ev, int nsteps)
{
    constexpr int NS = NANT - 1, GH = G * H, EPW = FRIRL_WAVE / GH;
    using STORE = typename std::conditional<IDX, StoreIdx<NANT>, StoreF64<NANT>>::type;
    extern __shared__ double tab_s[];                          // [NANT][U] VE tables (if lds_ve), then [NANT][U] universes (if lds_u)
    __shared__ double grid_s[NANT * FRIRL_HIP_MAX_GRID];
    __shared__ double ave_s[FRIRL_HIP_MAX_ACTIONS];
    const int U = la.U, maxR = la.maxR;
    const int lane = threadIdx.x & (FRIRL_WAVE - 1), wave = threadIdx.x / FRIRL_WAVE;
    const int gl = lane % GH, sub = gl % G, h = gl / G, il = lane / GH, base = lane - gl;      // group lane = (rule slice h, conclusion slot sub)
    const int tile = blockIdx.x * LN_WPB + wave, e = tile * EPW + il;
    const bool exists = tile < la.tiles && e < la.E;
    if (la.lds_ve) for (int i = threadIdx.x; i < NANT * U; i += LN_BLOCK) tab_s[i] = la.ve[i];
    if (la.lds_u) for (int i = threadIdx.x; i < NANT * U; i += LN_BLOCK) tab_s[NANT * U + i] = la.u[i];
    for (int i = threadIdx.x; i < NANT * FRIRL_HIP_MAX_GRID; i += LN_BLOCK) grid_s[i] = ag.grid_values[i];
    if ((int)threadIdx.x < ag.A) ave_s[threadIdx.x] = ag.action_ve[threadIdx.x];
    __syncthreads();
    const double *ves = la.lds_ve ? tab_s : la.ve, *us = la.lds_u ? tab_s + NANT * U : la.u;
    STORE st;
    if constexpr (IDX) {
        st.Ti = reinterpret_cast<typename StoreIdx<NANT>::RecI *>(la.T) + (size_t)tile * maxR * EPW + il;
        st.Tq = reinterpret_cast<double *>(static_cast<char *>(la.T) + la.tq_off) + (size_t)tile * maxR * EPW + il;
        st.ve_s = tab_s;
        st.U = U;
        st.EPW = EPW;
    } else {
        st.Te = static_cast<double *>(la.T) + (size_t)tile * (NANT + 1) * maxR * EPW + il;
        st.maxR = maxR;
        st.EPW = EPW;
    }
    using POW = typename std::conditional<PN, PowC<NANT>, PowU>::type;
    POW p;
    if constexpr (!PN) p.p = ag.p > 0 ? ag.p : NANT;
    const auto pk = pin_pow(p);                        // series coefficients of the Shepard weight in registers (sweeps.h)
    const bool has_q = (sub == G - 1);

    double states[NS], q_ant[NANT], total = 0.0;
    int R = 0, fus = 0, steps = 0, status = FRIRL_HIP_UPD_INACTIVE;
    bool active = false;
    uint32_t episode = 0;
#pragma unroll
    for (int k = 0; k < NS; k++) states[k] = exists ? ev.states[(size_t)e * NS + k] : 0.0;
#pragma unroll
    for (int k = 0; k < NANT; k++) q_ant[k] = exists ? ev.q_ant[(size_t)e * NANT + k] : 0.0;
    if (exists) {
        R = la.nrules[e]; fus = ev.fus[e]; steps = ev.ep_steps[e]; total = ev.ep_reward[e];
        active = ev.done[e] == 0;
        episode = ev.episode ? (uint32_t)ev.episode[e] : 0u;
    }
    const bool was_active = active;
    int last_it = -1;                 // last iteration this environment took part in

    for (int it = 0; it < nsteps; it++) {
        if (!__any(active ? 1 : 0)) break;
        if (active) {
            last_it = it;
            double cur[NS], cur_q[NANT], reward;
            int success;
            env_do_action(ag.env_kind, q_ant[NS], states, cur);                                         // frirl_episode.c:97
            env_get_reward(ag.env_kind, cur, reward, success);                                          // :106
            env_quantize(ag.env_kind, NS, grid_s, ag.grid_len, ag.grid_div, cur, cur_q);                // :112
            double ve1[NANT], ve2[NS];
#pragma unroll
            for (int k = 0; k < NANT; k++) ve1[k] = observe_ve(us, ves, U, k, q_ant[k]);
#pragma unroll
            for (int k = 0; k < NS; k++) ve2[k] = observe_ve(us, ves, U, k, cur_q[k]);

            // ---- one pass over the rules: this lane's conclusions (frirl_get_best_action :148 / frirl_update_sarsa.c:357)
            double qsel[NS], apt[APL], sv[APL], sw[APL], conc[APL];
            unsigned hit[APL];
#pragma unroll
            for (int k = 0; k < NS; k++) qsel[k] = has_q ? ve1[k] : ve2[k];
            int nacc = has_q ? 1 : ag.A - sub * APL;
            nacc = nacc < 0 ? 0 : (nacc > APL ? APL : nacc);
#pragma unroll
            for (int i = 0; i < APL; i++) {
                const int a = sub * APL + i;
                apt[i] = has_q ? ve1[NS] : ave_s[a < ag.A ? a : 0];
                sv[i] = 0.0; sw[i] = 0.0; hit[i] = FRIRL_HIP_NO_HIT;
            }
            for_rules<true, NANT>(st, h, H, R, [&](int r, const double (&c)[NANT + 1]) {
                const double d0 = qsel[0] - c[0];
                double s = d0 * d0;
#pragma unroll
                for (int k = 1; k < NS; k++) { const double d = qsel[k] - c[k]; s = __fma_rn(d, d, s); }
                const double va = c[NS], cq = c[NANT];
#pragma unroll
                for (int i = 0; i < APL; i++) {
                    if (i < nacc) {
                        const double ea = apt[i] - va;
                        const double d2 = __fma_rn(ea, ea, s);
                        // exact hit: noted with a select, its conclusion's sums are poisoned and not read (the conclusion is the
                        // hit rule's consequent, FIVEVagConcl_FRIRL_BestAct.c:89-93)
                        hit[i] = (d2 == 0.0 && hit[i] == FRIRL_HIP_NO_HIT) ? (unsigned)r : hit[i];
                        const double wi = shepard_w(d2, pk);
                        sv[i] = __fma_rn(wi, cq, sv[i]);
                        sw[i] = sw[i] + wi;
                    }
                }
            });
            if (H > 1) {                    // the H rule slices of each conclusion: sums in slice order, lowest exact hit
#pragma unroll
                for (int i = 0; i < APL; i++) {
                    double tv = __shfl(sv[i], base + sub), tw = __shfl(sw[i], base + sub);
                    unsigned th = (unsigned)__shfl((int)hit[i], base + sub);
#pragma unroll
                    for (int hh = 1; hh < H; hh++) {
                        const double v = __shfl(sv[i], base + hh * G + sub), w = __shfl(sw[i], base + hh * G + sub);
                        const unsigned x = (unsigned)__shfl((int)hit[i], base + hh * G + sub);
                        tv = tv + v;
                        tw = tw + w;
                        th = x < th ? x : th;
                    }
                    sv[i] = tv; sw[i] = tw; hit[i] = th;
                }
            }
            double bv = -__builtin_inf();
            int bi = ag.A;
#pragma unroll
            for (int i = 0; i < APL; i++) {
                conc[i] = 0.0;
                if (i < nacc) {
                    conc[i] = (hit[i] != FRIRL_HIP_NO_HIT) ? *st.qptr((int)hit[i]) : sv[i] / sw[i];
                    const int a = sub * APL + i;
                    if (!has_q && (a == 0 || bv < conc[i])) { bv = conc[i]; bi = a; }                   // first maximum, max.inl:21
                }
            }
            // greedy action over the group's action lanes, in action order
            double cb = __shfl(bv, base);
            int ci = __shfl(bi, base);
#pragma unroll
            for (int g = 1; g < G - 1; g++) {
                const double v = __shfl(bv, base + g);
                const int i2 = __shfl(bi, base + g);
                if (cb < v) { cb = v; ci = i2; }
            }
            const int chosen = e_greedy(ag, ci, (uint32_t)e, episode, (uint32_t)steps + 1u);
            const int slot = chosen % APL;
            double mine = conc[0];
#pragma unroll
            for (int i = 1; i < APL; i++) if (i == slot) mine = conc[i];
            const double qp = __shfl(mine, base + chosen / APL);                                        // Q(s',a'), frirl_update_sarsa.c:356
            const double qnow = __shfl(conc[0], base + G - 1);                                          // Q(s,a), :357
            const double ws1 = __shfl(sw[0], base + G - 1);
            const double vs1 = __shfl(sv[0], base + G - 1);
            const unsigned hit1 = (unsigned)__shfl((int)hit[0], base + G - 1);
            cur_q[NS] = grid_s[NS * FRIRL_HIP_MAX_GRID + chosen];                                        // frirl_episode.c:151

            // ---- frirl_update_sarsa + update_rules (frirl_update_sarsa.c:348-385, :22-143); every lane of the group follows
            //      the same branch, stores are issued by one lane (or split over the lanes for the weighted spread)
            status = FRIRL_HIP_UPD_INACTIVE;
            if (!ag.evaluate) {                                                                         // frirl_episode.c:155
                const double qdiff = ag.alpha * (reward + ag.gamma * qp - qnow);                        // :358
                bool finished = false;
                if (qdiff > ag.qdiff_pos_boundary || qdiff < ag.qdiff_neg_boundary) {                   // :363
                    double rant[NANT], ve3[NANT];
                    unsigned idx3[NANT];
                    bool same = true;
#pragma unroll
                    for (int k = 0; k < NANT; k++) {
                        rant[k] = check_possible_states(q_ant[k], grid_s + k * FRIRL_HIP_MAX_GRID, ag.grid_len[k]);   // :146-170
                        const double *uni = us + (size_t)k * U;
                        idx3[k] = snap_index(uni, U, rant[k], universe_div(uni, U));
                        ve3[k] = ves[(size_t)k * U + idx3[k]];
                        same = same && (ve3[k] == ve1[k]);
                    }
                    LaneQ rr{vs1, ws1, hit1};                                                           // :370 (same VE point => same sums)
                    if (!same) rr = lane_sweep_q<NANT>(st, R, ve3, p);
                    if (rr.hit == FRIRL_HIP_NO_HIT) {                                                   // :373-377 append and leave
                        if (R >= maxR) {
                            status = FRIRL_HIP_UPD_FULL;
                        } else {
                            if (gl == 0) {
                                st.append(R, ve3, idx3, rr.v / rr.w + qdiff);
#pragma unroll
                                for (int k = 0; k < NANT; k++) {
                                    if (IDX) la.rb[((size_t)e * (NANT + 1) + k) * maxR + R] = ve3[k];    // five_add_rule.c:80-81 (canonical slab)
                                    if (la.uidx) la.uidx[((size_t)e * NANT + k) * maxR + R] = (uint16_t)idx3[k];   // :76
                                    if (ev.rant) ev.rant[((size_t)e * NANT + k) * maxR + R] = rant[k];
                                }
                            }
                            R++;
                            fus = 1;
                            status = FRIRL_HIP_UPD_INSERTED;
                        }
                        finished = true;
                    } else {
                        fus = 0;                                                                        // :378
                    }
                }
                if (!finished) {
                    const int rules = fus ? R - 1 : R;                                                  // :30-33
                    if (hit1 != FRIRL_HIP_NO_HIT && (ag.skip_rules == 0 || (ag.skip_rules == 1 && (int)hit1 < rules))) {
                        if (gl == 0) *st.qptr((int)hit1) = qnow + qdiff;                                // :55
                        status = FRIRL_HIP_UPD_EXACT;
                    } else if (ag.skip_rules == 1 && hit1 != FRIRL_HIP_NO_HIT && (int)hit1 == rules) {
                        status = FRIRL_HIP_UPD_SKIPPED;                                                 // :61-63
                    } else {
                        if (ag.skip_rules == 0) fus = 0;                                                // :70-73
                        const int r_skip = fus ? R - 1 : -1;                                            // :76,124-126
                        if (hit1 == FRIRL_HIP_NO_HIT && gl == 0 && ev.spread_ant) {     // this call defines FIVERB.weights from now on (frirl_hip.h)
#pragma unroll
                            for (int k = 0; k < NANT; k++) ev.spread_ant[(size_t)e * NANT + k] = q_ant[k];
                            if (ev.spread_R) ev.spread_R[e] = R;
                        }
                        const double iws = 1.0 / ws1;
                        for_rules<false, NANT>(st, gl, GH, R, [&](int r, const double (&c)[NANT + 1]) {   // K6 + K7, rules split over the group
                            const double d0 = ve1[0] - c[0];
                            double s = d0 * d0;
#pragma unroll
                            for (int k = 1; k < NANT; k++) { const double d = ve1[k] - c[k]; s = __fma_rn(d, d, s); }
                            const double w = shepard_w(s, p) * iws;
                            if (w > ag.weight_significant && r != r_skip) { const double t = qdiff * w; *st.qptr(r) = qnow + t; }
                        });
                        status = FRIRL_HIP_UPD_SPREAD;
                    }
                }
                __threadfence_block();      // the group's stores are visible to its other lanes before the next sweep
            }
#pragma unroll
            for (int k = 0; k < NS; k++) { states[k] = cur[k]; q_ant[k] = cur_q[k]; }                   // :163-168
            q_ant[NS] = cur_q[NS];
            steps++;                                                                                    // :174
            total = total + reward;                                                                     // :107
            if (success == 1 || steps >= ag.max_steps) active = false;                                  // :183, :86
        }
    }
    if (!exists || gl != 0) return;
    // as nsteps launches of the step kernel would leave it: the update of the LAST of the nsteps steps, INACTIVE for an
    // environment that had finished before it
    if (ev.status) ev.status[e] = (was_active && last_it == nsteps - 1) ? status : FRIRL_HIP_UPD_INACTIVE;
    if (!was_active) return;
#pragma unroll
    for (int k = 0; k < NS; k++) ev.states[(size_t)e * NS + k] = states[k];
#pragma unroll
    for (int k = 0; k < NANT; k++) ev.q_ant[(size_t)e * NANT + k] = q_ant[k];
    ev.fus[e] = fus;
    ev.ep_steps[e] = steps;
    ev.ep_reward[e] = total;
    ev.done[e] = active ? 0 : 1;
    la.nrules[e] = R;
}

}  // namespace frirl

using namespace frirl_host;
int frirl_check_episode(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, const frirl_hip_envs *envs, const char *who);

static int lanes_group(int A) { return A <= 3 ? 4 : 8; }
// rule slices per conclusion: 1 once the groups alone give the chip >= 2048 waves, else 2, 4 or 8
static int lanes_slices(int E, int A)
{
    { const int v = frirl_host::opts().lanes_slices; if (v == 1 || v == 2 || v == 4 || v == 8) return v; }
    const int epw = FRIRL_WAVE / lanes_group(A);
    const int tiles = (E + epw - 1) / epw;
    return tiles >= 2048 ? 1 : (tiles >= 1024 ? 2 : (tiles >= 512 ? 4 : 8));
}
static int lanes_apl(int A) { const int g = lanes_group(A); const int apl = (A + g - 2) / (g - 1); return apl <= 1 ? 1 : (apl <= 3 ? 3 : 5); }

extern "C" size_t frirl_hip_lanes_workspace_bytes(int32_t nant, int32_t E, int32_t maxR, int32_t A)
{
    if (nant < 1 || E < 1 || maxR < 1 || A < 1) return 0;
    const size_t envs = ((size_t)E + FRIRL_WAVE - 1) / FRIRL_WAVE * FRIRL_WAVE;      // whole tiles for any group size
    return envs * (size_t)(nant + 1) * (size_t)maxR * sizeof(double);             // f64 store; the index store needs less
}

// Measured against the per-environment kernels (profiles/r01_learning.md): with up to 8 rule slices per conclusion the
// lane groups win at every batch size tried (96 ... 65 536 agents of the three demos).
extern "C" int frirl_hip_lanes_preferred(int32_t nant, int32_t E, int32_t A)
{
    if (nant < 1 || E < 1 || A < 1) return 0;
    const int epw = FRIRL_WAVE / lanes_group(A);
    (void)epw;
    return 1;
}

template <int N, int APL, int G, int H, bool IDX, bool PN = true>
static void launch_lanes(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev, int nsteps,
                         void *T, hipStream_t s)
{
    constexpr int EPW = FRIRL_WAVE / (G * H), W = N <= 4 ? 2 : 4;
    frirl::LanesArgs la;
    la.u = t->u; la.ve = t->ve; la.U = t->U; la.T = T; la.rb = b->rb; la.uidx = b->uidx; la.nrules = b->nrules; la.E = b->E; la.maxR = b->maxR;
    la.tiles = (b->E + EPW - 1) / EPW;
    const size_t tab = sizeof(double) * N * (size_t)t->U;
    la.lds_ve = IDX ? 1 : (2 * tab <= 32 * 1024);
    la.lds_u = 2 * tab <= 32 * 1024;
    la.tq_off = (size_t)la.tiles * EPW * b->maxR * W * sizeof(uint32_t);
    const size_t dyn = (la.lds_ve ? tab : 0) + (la.lds_u ? tab : 0);
    if (IDX)
        hipLaunchKernelGGL(frirl::lanes_import_idx_kernel, dim3(la.tiles), dim3(256), 0, s, b->rb, b->uidx, b->nrules, b->E, N, b->maxR, EPW, W,
                           static_cast<uint32_t *>(T), reinterpret_cast<double *>(static_cast<char *>(T) + la.tq_off));
    else
        hipLaunchKernelGGL(frirl::lanes_import_kernel, dim3(la.tiles, N + 1), dim3(256), 0, s, b->rb, b->nrules, b->E, N + 1, b->maxR, EPW, static_cast<double *>(T));
    // registers: 2 waves per SIMD keep both rule batches and the environment state in VGPRs; for the 3-antecedent kernels
    // with more environments than that can hold at once, 4 waves per SIMD (a few cold values in scratch) hide more
    // latency (measured: mountaincar x 65 536 agents 1.25 -> 1.54e9 env-steps/s; the 5-antecedent kernels lose)
    const int blocks = (la.tiles + frirl::LN_WPB - 1) / frirl::LN_WPB;
    int wpe = (la.tiles > 2048 && N <= 3) ? 4 : 2;
    { const int v = frirl_host::opts().lanes_wpe; if (v == 2 || v == 4) wpe = v; }
#define LANES_GO(WPE) hipLaunchKernelGGL((frirl::episode_run_lanes_kernel<N, APL, G, H, WPE, IDX, PN>), dim3(blocks), dim3(frirl::LN_BLOCK), dyn, s, la, *ag, *ev, nsteps)
    if constexpr (!PN) { (void)wpe; LANES_GO(2); }
    else { if (wpe == 4) LANES_GO(4); else LANES_GO(2); }
#undef LANES_GO
    if (IDX)          // antecedents of appended rules were written through; only the consequents come back
        hipLaunchKernelGGL(frirl::lanes_export_kernel, dim3(la.tiles, 1), dim3(256), 0, s, b->rb, b->nrules, b->E, N + 1, b->maxR, EPW,
                           reinterpret_cast<const double *>(static_cast<char *>(T) + la.tq_off), 1, N);
    else
        hipLaunchKernelGGL(frirl::lanes_export_kernel, dim3(la.tiles, N + 1), dim3(256), 0, s, b->rb, b->nrules, b->E, N + 1, b->maxR, EPW,
                           static_cast<const double *>(T), N + 1, 0);
}

extern "C" int frirl_hip_episode_run_lanes(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                                           const frirl_hip_envs *envs, int32_t nsteps, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = frirl_check_episode(t, b, agent, envs, "frirl_hip_episode_run_lanes");
    if (rc) return rc;
    if (reinterpret_cast<uintptr_t>(workspace) & 15) { set_error("frirl_hip_episode_run_lanes: workspace must be 16-byte aligned"); return FRIRL_HIP_EINVAL; }
    const bool pn = agent->p <= 0 || agent->p == t->nant;      // default Shepard power: compile-time constant in the kernels
    if (nsteps < 0 || !workspace) { set_error("frirl_hip_episode_run_lanes: nsteps=%d / workspace=%p", nsteps, workspace); return FRIRL_HIP_EINVAL; }
    const size_t need = frirl_hip_lanes_workspace_bytes(t->nant, b->E, b->maxR, agent->A);
    if (workspace_bytes < need) { set_error("frirl_hip_episode_run_lanes: workspace %zu B < %zu B (frirl_hip_lanes_workspace_bytes)", workspace_bytes, need); return FRIRL_HIP_EINVAL; }
    hipStream_t s = as_stream(stream);
    const int G = lanes_group(agent->A), apl = lanes_apl(agent->A);
    // index store when the caller keeps the 16-bit index mirror and the VE tables fit in LDS (48 KiB)
    bool idx = b->uidx != nullptr && sizeof(double) * t->nant * (size_t)t->U <= 48 * 1024 && t->U <= 65536;
    if (frirl_host::opts().no_uidx == 1) idx = false;
    const int H = pn ? lanes_slices(b->E, agent->A) : 1;       // run-time power: the variants without rule slices
#define RUN2P(N, IDX)                                                                                 \
    do {                                                                                              \
        if (G == 4) launch_lanes<N, 1, 4, 1, IDX, false>(t, b, agent, envs, nsteps, workspace, s);    \
        else if (apl == 1) launch_lanes<N, 1, 8, 1, IDX, false>(t, b, agent, envs, nsteps, workspace, s); \
        else if (apl == 3) launch_lanes<N, 3, 8, 1, IDX, false>(t, b, agent, envs, nsteps, workspace, s); \
        else launch_lanes<N, 5, 8, 1, IDX, false>(t, b, agent, envs, nsteps, workspace, s);           \
    } while (0)
#define RUN2(N, IDX, HH)                                                                              \
    do {                                                                                              \
        if (G == 4) launch_lanes<N, 1, 4, HH, IDX>(t, b, agent, envs, nsteps, workspace, s);          \
        else if (apl == 1) launch_lanes<N, 1, 8, HH, IDX>(t, b, agent, envs, nsteps, workspace, s);   \
        else if (apl == 3) launch_lanes<N, 3, 8, HH, IDX>(t, b, agent, envs, nsteps, workspace, s);   \
        else launch_lanes<N, 5, 8, HH, IDX>(t, b, agent, envs, nsteps, workspace, s);                 \
    } while (0)
#define RUN(N, IDX)                                                            \
    do {                                                                       \
        if (H == 8) RUN2(N, IDX, 8); else if (H == 4) RUN2(N, IDX, 4); else if (H == 2) RUN2(N, IDX, 2); else RUN2(N, IDX, 1); \
    } while (0)
    if (!pn) {
        if (t->nant == 3) { if (idx) RUN2P(3, true); else RUN2P(3, false); }
        else { if (idx) RUN2P(5, true); else RUN2P(5, false); }
    } else if (t->nant == 3) { if (idx) RUN(3, true); else RUN(3, false); }
    else { if (idx) RUN(5, true); else RUN(5, false); }
#undef RUN
#undef RUN2
#undef RUN2P
    return check_launch("frirl_hip_episode_run_lanes");
}
