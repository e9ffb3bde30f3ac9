// sweeps.h -- workgroup-level sweeps over ONE rule base (device functions).
//
// One workgroup owns one environment's rule base for the duration of a sweep (the reference's
// model: one private FIVERB per agent, src/frirl/frirl_agent.c:229-238), so no inter-workgroup
// communication and no float atomics are needed.  Every lane streams 16 B (two rules) per SoA
// column per iteration; partial sums are lane-strided and combined by a fixed butterfly + wave
// order (deterministic).  These sweeps are shared by the thin per-function kernels
// (fiveq.hip) and by the fused SARSA / episode-step kernels (sarsa.hip).
#pragma once

#include <stdlib.h>

#include <type_traits>

#include "device_common.h"

namespace frirl {

// b^p by repeated multiplication -- reference src/inl/fast_pow.inl:17-31, in plain double (the
// reference keeps the running product in x87 extended precision; see include/frirl_hip.h).
__device__ __forceinline__ double pow_int(double b, int p)
{
    double r = b;
    for (int i = 1; i < p; i++) r = r * b;
    return r;
}

// Shepard weight 1/d^p from the SQUARED distance s = d^2 > 0 (reference: wi = 1.0 / fast_pow(fast_abs(d), p),
// FIVEVagConcl.c:226, with d = sqrt(s)).  The reference's sqrt + p-1 multiplies + divide cost ~70 FP64
// instructions per rule on CDNA4 (software sqrt and divide) and make the Q sweeps ALU-bound; here
// y = rsqrt(s) (v_rsq_f64) is refined by one third-order step (relative error ~1e-16, i.e. the same size as the
// rounding of the reference's own sqrt/pow/divide chain) and raised to p by multiplication: ~20 instructions.
// Exact-hit detection does not depend on it (d == 0 <=> s == 0).  Interpolated Q values are contractually
// within 1e-6 relative of the reference (include/frirl_hip.h); materialised distances (five_hip_rule_distance)
// keep the IEEE sqrt and stay bit-exact.
template <bool UNROLLED_POWERS = false>
__device__ __forceinline__ double inv_dist_pow(double s, int p)
{
    // v_rsq_f64 is good to 5.2e-8; ONE third-order step y (1 + e/2 + 3 e^2/8), e = 1 - s y^2, brings it to 1.4e-16
    // (measured, tools/exp/rsq_prec.hip: the same as two Newton steps) in 5 instead of 8 FP64 instructions
    double y = __builtin_amdgcn_rsq(s);
    const double t = s * y;
    const double e = __fma_rn(-t, y, 1.0);
    const double c = __fma_rn(0.375, e, 0.5);
    const double ce = c * e;
    y = __fma_rn(y, ce, y);
    // y^p.  UNROLLED_POWERS: the demos' powers (p = nant = 3, 5) without a loop -- a data-dependent trip count costs a
    // wave its branch latency on every rule when only one or two waves share a SIMD (lane-group / shared-base kernels:
    // +6..9 %); the many-wave per-environment sweeps keep the plain loop (measured 8 % faster there)
    if (UNROLLED_POWERS) {
        const double y2 = y * y;
        if (p == 3) return y2 * y;
        if (p == 5) return (y2 * y2) * y;
    }
    double w = y;
    for (int i = 1; i < p; i++) w = w * y;
    return w;
}

// The weighted sums sum wi * Q accumulate with one explicit FMA per term (one rounding instead of two, one instruction
// instead of two); they are interpolated values (<= 1e-6 contract), never compared bit for bit.
// Shepard power as a type: PowC<P> (compile-time, straight-line y^P) or plain int (run-time loop).  Every demo and every
// BASELINE configuration uses p = nant (FIVEInit.c:89-93), so the hot kernels are instantiated with PowC<NANT>: no scalar
// loop and no branch per conclusion (the env-step kernel's loop body was ~40 % scalar/branch instructions).
template <int P>
struct PowC {};

__device__ __forceinline__ double shepard_w(double s, int p) { return inv_dist_pow<false>(s, p); }

// Straight-line form for a compile-time power (the hot kernels).  Instead of refining y = s^(-1/2) and THEN raising it to P, the
// power of the raw v_rsq_f64 value y0 (relative error <= 5.2e-8) is corrected once: with e = 1 - s y0^2,
//   s^(-P/2) = y0^P (1 - e)^(-P/2) = y0^P (1 + a e + a (a + 1) / 2 e^2 + O(e^3)),  a = P / 2,  |O(e^3)| < 1e-20,
// and y0^2 is shared between e and the power: 7 FP64 instructions after the rsq for P = 5 (y2, e, y4, y5, c, c e, fma) instead of
// 9 (6 instead of 7 for P = 3), same 1e-16 accuracy (tools/exp/shepard_prec.hip).  v_rsq_f64 itself issues in ~3.4 FP64 slots
// (tools/exp/valu_cost.hip); the f32 detour (cvt, v_rsq_f32, cvt) costs the same.
template <int P>
__device__ __forceinline__ double shepard_series(double s, double a, double a2)
{
    const double y = __builtin_amdgcn_rsq(s);
    const double y2 = y * y;
    const double e = __fma_rn(-s, y2, 1.0);
    double yp;
    if constexpr (P == 1) yp = y;
    else if constexpr (P == 2) yp = y2;
    else if constexpr (P == 3) yp = y2 * y;
    else if constexpr (P == 4) yp = y2 * y2;
    else if constexpr (P == 5) { const double y4 = y2 * y2; yp = y4 * y; }
    else if constexpr (P == 6) { const double y4 = y2 * y2; yp = y4 * y2; }
    else if constexpr (P == 8) { const double y4 = y2 * y2; yp = y4 * y4; }
    else {
        yp = y2;
#pragma unroll
        for (int i = 2; i < P; i++) yp = yp * y;
    }
    const double c = __fma_rn(a2, e, a);
    const double ce = c * e;
    return __fma_rn(yp, ce, yp);
}

template <int P>
__device__ __forceinline__ double shepard_w(double s, PowC<P>)
{
    constexpr double a = 0.5 * P, a2 = 0.5 * a * (a + 1.0);
    return shepard_series<P>(s, a, a2);
}

// The same with the two series coefficients held in registers for the whole sweep (a: VGPR pair, a2: SGPR pair -- a VOP3 takes one
// scalar operand): as immediates the compiler rebuilds `a` with two v_mov_b32 in front of every v_fmac (neither is an inline
// constant), ~1 FP64 issue slot per conclusion.  For the FP64-issue-bound action-parallel sweep, which has the registers.
template <int P>
struct PowCP { double a, a2; };
template <int P>
__device__ __forceinline__ PowCP<P> pin_pow(PowC<P>)
{
    PowCP<P> r;
    r.a = 0.5 * P;
    r.a2 = 0.5 * (0.5 * P) * (0.5 * P + 1.0);
    asm volatile("" : "+v"(r.a));
    asm volatile("" : "+s"(r.a2));
    return r;
}
__device__ __forceinline__ int pin_pow(int p) { return p; }
template <int P>
__device__ __forceinline__ double shepard_w(double s, PowCP<P> k)
{
    return shepard_series<P>(s, k.a, k.a2);
}

// run-time power with the p = 3 / 5 cases unrolled (low-occupancy kernels)
struct PowU { int p; };
__device__ __forceinline__ double shepard_w(double s, PowU r) { return inv_dist_pow<true>(s, r.p); }
__device__ __forceinline__ PowU pin_pow(PowU p) { return p; }

template <bool PN, int N>
struct PowSel { static __device__ __forceinline__ int make(int p) { return p; } };
template <int N>
struct PowSel<true, N> { static __device__ __forceinline__ PowC<N> make(int) { return PowC<N>(); } };

template <int BLOCK>
struct BlockRed {
    static constexpr int WAVES = BLOCK / FRIRL_WAVE;
    double d[WAVES];
    unsigned u[WAVES];
};

template <int BLOCK>
__device__ __forceinline__ unsigned blk_min(unsigned v, BlockRed<BLOCK> &s)
{
    v = wave_min_u32(v);
    const int lane = threadIdx.x & (FRIRL_WAVE - 1), wave = threadIdx.x / FRIRL_WAVE;
    if (lane == 0) s.u[wave] = v;
    __syncthreads();
    unsigned m = s.u[0];
#pragma unroll
    for (int w = 1; w < BlockRed<BLOCK>::WAVES; w++) m = (s.u[w] < m) ? s.u[w] : m;
    __syncthreads();
    return m;
}

template <int BLOCK>
__device__ __forceinline__ double blk_sum(double v, BlockRed<BLOCK> &s)
{
    v = wave_sum_f64(v);
    const int lane = threadIdx.x & (FRIRL_WAVE - 1), wave = threadIdx.x / FRIRL_WAVE;
    if (lane == 0) s.d[wave] = v;
    __syncthreads();
    double t = s.d[0];
#pragma unroll
    for (int w = 1; w < BlockRed<BLOCK>::WAVES; w++) t = t + s.d[w];
    __syncthreads();
    return t;
}

// Streaming 16-byte column load.  The sweeps read every rule once per pass and nothing is re-read before the
// slab has left the caches, so the non-temporal hint applies (measured: episode step 0.459 -> 0.418 ms at cfg2).
// (Issuing two column sets per iteration was tried and lost 15 %: registers, not memory parallelism, bind here.)
__device__ __forceinline__ double2 load_col2(const double *__restrict__ p)
{
    double2 v;
    v.x = __builtin_nontemporal_load(p);
    v.y = __builtin_nontemporal_load(p + 1);
    return v;
}

// the same without the non-temporal hint: for sweeps whose waves read the SAME columns (sweep_gba_wide: the action-parallel waves of a
// workgroup) -- the first wave's line then serves the others from the CU's vector L1 / the XCD's L2 instead of going out four times
__device__ __forceinline__ double2 load_col2_shared(const double *__restrict__ p)
{
    return *reinterpret_cast<const double2 *>(p);
}

// Antecedent-column accessors.  ColsF64 streams the f64 SoA columns (reference layout).  ColsIdx streams the 16-bit
// universe indices (frirl_hip_rulebases.uidx) and reads the VE values from an LDS copy of the tables: the same
// doubles (rb[k][r] == ve[k][uidx[k][r]] exactly), a quarter of the antecedent bytes.
struct ColsF64 {
    static constexpr bool GLOBAL_Q = true;     // the consequent column lies in global memory (QColBuf applies)
    const double *base;   // rb slab of the environment
    int maxR;
    __device__ __forceinline__ double2 pair(int k, int r) const { return load_col2(base + (size_t)k * maxR + r); }
    using raw_t = double2;                 // raw()/decode(): the global load split from its use (software prefetch)
    __device__ __forceinline__ raw_t raw(int k, int r) const { return pair(k, r); }
    __device__ __forceinline__ raw_t raw_shared(int k, int r) const { return load_col2_shared(base + (size_t)k * maxR + r); }
    __device__ __forceinline__ double2 decode(int, const raw_t &w) const { return w; }
};

// Streams of the hot sweeps as raw BUFFER loads: one descriptor per slab (4 SGPRs, built once), the column offset k * maxR in a scalar
// register, ONE per-lane byte offset shared by all columns -- against one 64-bit pointer per column, advanced with two vector
// instructions each per iteration.  The step kernels are bound by vector issue; this takes ~6 of their ~200 instructions per pair of rules.
#ifndef FRIRL_BUF_LOADS
#define FRIRL_BUF_LOADS 1
#endif
static constexpr int BUFFER_RSRC_DWORD3 = 0x00020000;      // gfx9 family: raw buffer, 32-bit data format
__device__ __forceinline__ __amdgpu_buffer_rsrc_t raw_buffer(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)0x7fffffff, BUFFER_RSRC_DWORD3);
}
// consequent column as a buffer: 16-byte non-temporal loads of (q[r], q[r + 1])
struct QColBuf {
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ explicit QColBuf(const double *qcol) : rs(raw_buffer(qcol)) {}
    __device__ __forceinline__ double2 load2(int r) const
    {
        const auto w = __builtin_amdgcn_raw_buffer_load_b128(rs, 8 * r, 0, 2);
        double2 v;
        v.x = __longlong_as_double((long long)(((unsigned long long)w[1] << 32) | w[0]));
        v.y = __longlong_as_double((long long)(((unsigned long long)w[3] << 32) | w[2]));
        return v;
    }
};

// one entry: table value at the low (HALF = 0) or high (HALF = 1) 16-bit index of a packed word
template <int HALF>
__device__ __forceinline__ double lds_table_entry(const double *tab_k, uint32_t w)
{
#if FRIRL_DECODE_MAD
    const uint32_t base = (uint32_t)(uintptr_t)(lds_cdouble *)tab_k;
    uint32_t a;
    if constexpr (HALF == 0) asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[0,0,0,0]" : "=v"(a) : "v"(w), "s"(base));
    else asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[1,0,0,0]" : "=v"(a) : "v"(w), "s"(base));
    return *(lds_cdouble *)(uintptr_t)a;
#else
    return tab_k[HALF ? (w >> 16) : (w & 0xFFFFu)];
#endif
}

struct ColsIdx {
    static constexpr bool GLOBAL_Q = true;
    const uint16_t *idx;  // uidx slab of the environment
    const double *tab;    // LDS [nant][U]
    int maxR, U;
    __device__ __forceinline__ double2 pair(int k, int r) const
    {
        const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(idx + (size_t)k * maxR + r));
        return lds_table_pair(tab + k * U, w);
    }
    using raw_t = uint32_t;
    __device__ __forceinline__ raw_t raw(int k, int r) const
    {
#if FRIRL_BUF_LOADS
        return __builtin_amdgcn_raw_buffer_load_b32(raw_buffer(idx), 2 * r, 2 * k * maxR, 2);       // aux 2: non-temporal
#else
        return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(idx + (size_t)k * maxR + r));
#endif
    }
    __device__ __forceinline__ raw_t raw_shared(int k, int r) const { return *reinterpret_cast<const uint32_t *>(idx + (size_t)k * maxR + r); }
    __device__ __forceinline__ double2 decode(int k, raw_t w) const { return lds_table_pair(tab + k * U, w); }
};

// rule base resident in LDS (persistent episode kernel): plain 16-byte reads of the workgroup's own slab copy
struct ColsLds {
    static constexpr bool GLOBAL_Q = false;
    const double *base;   // LDS [nant+1][cap]
    int cap;
    __device__ __forceinline__ double2 pair(int k, int r) const { return *reinterpret_cast<const double2 *>(base + (size_t)k * cap + r); }
    using raw_t = double2;
    __device__ __forceinline__ raw_t raw(int k, int r) const { return pair(k, r); }
    __device__ __forceinline__ raw_t raw_shared(int k, int r) const { return pair(k, r); }
    __device__ __forceinline__ double2 decode(int, const raw_t &w) const { return w; }
};

template <bool IDX>
struct ColsSel;
template <>
struct ColsSel<false> {
    using type = ColsF64;
    static __device__ __forceinline__ type make(const double *base, const uint16_t *, const double *, int maxR, int) { return ColsF64{base, maxR}; }
};
template <>
struct ColsSel<true> {
    using type = ColsIdx;
    static __device__ __forceinline__ type make(const double *, const uint16_t *idx, const double *tab, int maxR, int U) { return ColsIdx{idx, tab, maxR, U}; }
};

// Host-side choice: stream the 16-bit index mirror when it exists, the LDS table fits next to the kernel's static
// LDS (<= 48 KiB) and the rule bases are large enough for bandwidth to matter (small bases: the per-workgroup
// table fill would cost more than it saves).
static inline bool use_uidx(const frirl_hip_tables *t, const frirl_hip_rulebases *b)
{
    return b->uidx && t->U <= 65536 && sizeof(double) * t->nant * (size_t)t->U <= 48 * 1024 && b->maxR > 2048 && !frirl_host::opts().no_uidx;
}

// Squared VE distance of two adjacent rules (r, r+1) to the observation q over dims [0, NDIM):
// dimension-ordered, separate multiply and add (five_rule_distance.c:88-90,171-208).
template <int NDIM, class COLS>
__device__ __forceinline__ void sq_dist2(const COLS &cols, int r, const double (&q)[NDIM], double &a0, double &a1)
{
    double2 v[NDIM];
#pragma unroll
    for (int k = 0; k < NDIM; k++) v[k] = cols.pair(k, r);
    double d0 = q[0] - v[0].x, d1 = q[0] - v[0].y;
    a0 = d0 * d0;
    a1 = d1 * d1;
#pragma unroll
    for (int k = 1; k < NDIM; k++) {
        d0 = q[k] - v[k].x;
        d1 = q[k] - v[k].y;
        a0 = __fma_rn(d0, d0, a0);      // Q sweeps only: the materialised distances (five_rule_distance.hip) keep separate mul / add
        a1 = __fma_rn(d1, d1, a1);
    }
}

// Candidates for update_rules' masked write-back (frirl_update_sarsa.c:89-120): rconc[r] changes only where the normalised
// Shepard weight wi_r / ws exceeds the threshold (0.05 => at most 19 rules of the whole rule base), but ws is known only
// after the sweep -- the reference (and round 1) therefore sweep the rule base a second time.  Instead every lane keeps, during
// the Q(s,a) sweep, the two largest wi it has seen (+ the value of the third) in a slot of its own in LDS (registers are the
// scarce resource of these sweeps; kept in VGPRs the five values spilled the hot loop: 8x slower): a rule that finally qualifies satisfies
// wi > thr * ws >= thr * (any earlier partial sum of the WAVE), so the hot loop only compares wi with a wave-uniform bound T and
// the slot is touched on the rare iterations where some lane passes; T is refreshed there from the wave's running sums (a bound
// from the LANE's own sum is useless: Shepard weights are heavy-tailed, some lane of 64 passes it on ~90 % of the iterations
// -- measured 3x slower).  After the
// reduction the <= 2 candidates per lane are filtered with the exact test of the second sweep (wi * (1/ws) > thr, same
// operands, same bits); if a lane's THIRD value would qualify too (more than two qualifying rules in one lane) the
// workgroup falls back to the second sweep.  Saves one pass over the slab for ~40 % of the updates at the BASELINE shapes.
// the lane-local pre-filter compares against a slightly smaller threshold, so that the roundings of wi * (1 / ws) in the exact
// test can never let a rule qualify that the pre-filter skipped
static constexpr double SPREAD_PREFILTER_SLACK = 1.0 - 1.0 / 1048576.0;

struct SpreadCand {
    double w1, w2, w3;      // largest, second, third wi of this lane (0 = none)
    unsigned r1, r2;        // their rule indices
    __device__ __forceinline__ void clear() { w1 = w2 = w3 = 0.0; r1 = r2 = FRIRL_HIP_NO_HIT; }
    __device__ __forceinline__ void offer(double wi, unsigned r)
    {
        if (wi > w1) { w3 = w2; w2 = w1; r2 = r1; w1 = wi; r1 = r; }
        else if (wi > w2) { w3 = w2; w2 = wi; r2 = r; }
        else if (wi > w3) w3 = wi;
    }
};

// One rule's weight against the wave-uniform bound T (see above); `run` = this lane's running sum of the same sweep.
// Every lane of the wave must call it together (the refresh is a wave reduction).
// a wave-uniform double kept in scalar registers (the sweeps are VGPR-bound: every vector register spilled costs scratch traffic
// in the hot loop)
__device__ __forceinline__ double wave_uniform(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ void spread_track(SpreadCand *slots, double &T, double thr, double w0, double w1, unsigned r, double run)
{
    const bool p0 = w0 > T, p1 = w1 > T;      // w = 0 for exact hits / rules past the end: never passes
    if (__builtin_amdgcn_ballot_w64(p0 || p1) != 0ull) {       // wave-uniform, and rare once T is meaningful
        SpreadCand *slot = slots + threadIdx.x;
        if (p0) slot->offer(w0, r);
        if (p1) slot->offer(w1, r + 1u);
        T = wave_uniform(thr * wave_sum_f64(run));
    }
}

// trip count of a lane-strided sweep over R rules rounded up so that all 64 lanes of a wave leave the loop together
__device__ __forceinline__ int wave_uniform_limit(int R) { return ((R + 2 * FRIRL_WAVE - 1) / (2 * FRIRL_WAVE)) * (2 * FRIRL_WAVE); }

struct QResult {
    unsigned hit;   // lowest rule index with distance exactly 0, or FRIRL_HIP_NO_HIT
    double vagc;    // sum wi * Q   (valid when hit == NO_HIT)
    double ws;      // sum wi
    bool tracked;   // the lanes' SpreadCand slots hold the candidates of this sweep
};

// One PAIR of rules (r, r + 1) of a Q-value sweep with squared distances a0, a1 (a1 = NO_RULE_STATE_PART when rule r + 1 does not
// exist): the first exact hit is noted in a side path taken only when some lane of the wave has one; the Shepard terms are added
// unconditionally -- an exact hit poisons the two sums (rsq(0)), and a conclusion with an exact hit is the hit rule's consequent, its sums
// are not read (FIVEVagConcl.c:89-93; QResult::vagc / ws are valid when hit == NO_HIT).  No branch and no register copies around the
// FP64 chains of the common case.
static constexpr double NO_RULE_STATE_PART = 1.0e300;
template <class POW>
__device__ __forceinline__ void q_pair(double a0, double a1, const double2 &c, unsigned r, POW pk, unsigned &best, double &sv, double &sw, double &tw0,
                                       double &tw1)
{
    if (__builtin_amdgcn_ballot_w64(a0 == 0.0 || a1 == 0.0) != 0ull) {
        if (a0 == 0.0) best = min(best, r);
        if (a1 == 0.0) best = min(best, r + 1u);
    }
    const double w0 = shepard_w(a0, pk), w1 = shepard_w(a1, pk);
    sv = __fma_rn(w0, c.x, sv);
    sw = sw + w0;
    sv = __fma_rn(w1, c.y, sv);
    sw = sw + w1;
    tw0 = w0;
    tw1 = w1;
}

// FIVE_vag_concl's sweep (reference src/five/FIVEVagConcl.c:64-351 live path): distances, first
// exact hit, Shepard sums wi = 1/d^p, vagc = sum wi*Q, ws = sum wi (:224-235).  All threads return
// the same QResult.
template <int NANT, int BLOCK, bool TRACK = false, class COLS, class POW>
__device__ QResult sweep_q(const COLS &cols, const double *__restrict__ qcol, int R, const double (&q)[NANT], POW p, BlockRed<BLOCK> &red, double track_thr = 0.0,
                           SpreadCand *slot = nullptr)
{
    unsigned best = FRIRL_HIP_NO_HIT;
    double sv = 0.0, sw = 0.0;
    QResult res;
    res.tracked = TRACK;
    if (TRACK) slot[threadIdx.x].clear();          // `slot` = the workgroup's slot array, one entry per lane
    track_thr = wave_uniform(track_thr * SPREAD_PREFILTER_SLACK);
    double T = 0.0;
    const auto pk = pin_pow(p);
    // tracked form: wave-uniform trip count (spread_track is a wave-level operation); lanes past R idle through their last turn
    const int r_lim = TRACK ? wave_uniform_limit(R) : R;
    for (int r = 2 * (int)threadIdx.x; r < r_lim; r += 2 * BLOCK) {
        double tw0 = 0.0, tw1 = 0.0;
        if (!TRACK || r < R) {
            double a0, a1;
            sq_dist2<NANT>(cols, r, q, a0, a1);
            double2 c = load_col2(qcol + r);
            if (r + 1 >= R) { a1 = NO_RULE_STATE_PART; c.y = 0.0; }      // the phantom rule of an odd tail: weight exactly 0, nothing of it is read
            q_pair(a0, a1, c, (unsigned)r, pk, best, sv, sw, tw0, tw1);
        }
        if (TRACK) spread_track(slot, T, track_thr, tw0, tw1, (unsigned)r, sw);
    }
    res.hit = blk_min<BLOCK>(best, red);
    res.vagc = blk_sum<BLOCK>(sv, red);
    res.ws = blk_sum<BLOCK>(sw, red);
    return res;
}

// FIVE_vag_concl_weight's second pass (reference src/five/FIVEVagConclWeight.c:125-166, K6):
// weights[r] = (1/d_r^p) / ws for r < R.  Distances are recomputed (8*nant B/rule re-read) instead
// of spilling wi[] to HBM and reading it back (16 B/rule).
template <int NANT, int BLOCK, class COLS, class POW>
__device__ void sweep_weights(const COLS &cols, int R, const double (&q)[NANT], POW p, double ws, double *__restrict__ weights)
{
    const double iws = 1.0 / ws;
    for (int r = 2 * (int)threadIdx.x; r < R; r += 2 * BLOCK) {
        double a0, a1;
        sq_dist2<NANT>(cols, r, q, a0, a1);
        const double w0 = shepard_w(a0, p) * iws;
        const double w1 = shepard_w(a1, p) * iws;
        if (r + 1 < R) {
            double2 w; w.x = w0; w.y = w1;
            *reinterpret_cast<double2 *>(weights + r) = w;
        } else weights[r] = w0;
    }
}

// update_rules' masked write-back (reference src/frirl/frirl_update_sarsa.c:89-120, K7):
// rconc[r] = qnow + qdiff * w_r where w_r = wi_r / ws > threshold (strict, ordered compare).
// `r_skip` (or -1) is left untouched: the just-inserted last rule under skip_rules (:31-33,124-126).
template <int NANT, int BLOCK, class COLS, class POW>
__device__ void sweep_update(const COLS &cols, double *__restrict__ qcol, int R, const double (&q)[NANT], POW p, double ws, double qnow,
                             double qdiff, double threshold, int r_skip)
{
    const double iws = 1.0 / ws;
    for (int r = 2 * (int)threadIdx.x; r < R; r += 2 * BLOCK) {
        double a0, a1;
        sq_dist2<NANT>(cols, r, q, a0, a1);
        const double w0 = shepard_w(a0, p) * iws;
        const double w1 = shepard_w(a1, p) * iws;
        if (w0 > threshold && r != r_skip) { const double t = qdiff * w0; qcol[r] = qnow + t; }
        if (r + 1 < R && w1 > threshold && r + 1 != r_skip) { const double t = qdiff * w1; qcol[r + 1] = qnow + t; }
    }
}

// The same write-back from the per-lane candidates of a tracked Q(s,a) sweep (SpreadCand); false = some lane holds more
// than two qualifying rules: nothing was written, the caller runs sweep_update.
template <int BLOCK>
__device__ bool spread_from_candidates(const SpreadCand &cd, double *__restrict__ qcol, double ws, double qnow, double qdiff, double threshold, int r_skip,
                                       BlockRed<BLOCK> &red)
{
    const double iws = 1.0 / ws;
    const unsigned ovf = (cd.w3 * iws > threshold) ? 0u : 1u;
    if (blk_min<BLOCK>(ovf, red) == 0u) return false;
    if (cd.r1 != FRIRL_HIP_NO_HIT) {
        const double w = cd.w1 * iws;
        if (w > threshold && (int)cd.r1 != r_skip) { const double t = qdiff * w; qcol[cd.r1] = qnow + t; }
    }
    if (cd.r2 != FRIRL_HIP_NO_HIT) {
        const double w = cd.w2 * iws;
        if (w > threshold && (int)cd.r2 != r_skip) { const double t = qdiff * w; qcol[cd.r2] = qnow + t; }
    }
    return true;
}

// frirl_get_best_action's sweep (reference src/frirl/frirl_get_best_action.c:31-341): the state
// part of the squared distance is computed once per rule (K3/K4 :58-155) and shared by all A
// actions; per action a: d = sqrt((vevalues[a] - ract_veval[r])^2 + statesum) (K5 :252-275), then
// FIVEVagConcl_FRIRL_BestAct (first exact hit, else Shepard; FIVEVagConcl_FRIRL_BestAct.c:89-93,
// 212-217,265).  A accumulator pairs live in registers (AMAX is the compile-time bucket).
// Results: actconc[a] for a < A in `actconc_s` (LDS, >= AMAX doubles); returns the first maximum
// (src/inl/max.inl:16-28).  `scratch` needs WAVES*AMAX doubles x2 and WAVES*AMAX unsigned.
template <int AMAX, int BLOCK>
struct GbaScratch {
    static constexpr int WAVES = BLOCK / FRIRL_WAVE;
    double v[WAVES][AMAX];
    double w[WAVES][AMAX];
    unsigned hit[AMAX];         // first exact hit per action, recorded by note_state_hits (atomic min)
    double actconc[AMAX];
    double ave[AMAX];
    double tv[AMAX], tw[AMAX];  // the actions' reduced Shepard sums (sweep_gba_q: the pending conclusion of a same-cell step reads its action's)
    int best;
};

// One action against a PAIR of rules, Shepard terms only (no hit handling): the conclusion terms of every greedy sweep.  sweep_gba,
// sweep_gba_q and sweep_gba_many record exact hits outside the loop (note_state_hits below) and let a hit poison the sums of its own
// action; sweep_gba_wide (one action group per wave, hits in registers) takes this body on the wave-uniform fast path -- no lane holds a
// rule with a zero state part or an odd tail, so no hit is possible (d^2 >= s > 0) -- and a branchy body otherwise (cartpole's 21-action
// step with it: 2.54 -> 2.41 ms per 4096 environments).  Same operations in the same order in every form => bit-identical sums.
#ifndef FRIRL_GBA_FASTPATH
#define FRIRL_GBA_FASTPATH 1
#endif
template <class POW>
__device__ __forceinline__ void concl_pair_nohit(double av, const double2 &va, double s0, double s1, const double2 &c, POW p, double &sv, double &sw)
{
    const double e0 = av - va.x, e1 = av - va.y;
    const double d0 = __fma_rn(e0, e0, s0), d1 = __fma_rn(e1, e1, s1);
    const double w0 = shepard_w(d0, p), w1 = shepard_w(d1, p);
    sv = __fma_rn(w0, c.x, sv);
    sw = sw + w0;
    sv = __fma_rn(w1, c.y, sv);
    sw = sw + w1;
}
template <class POW>
__device__ __forceinline__ void concl_pair_nohit(double av, const double2 &va, double s0, double s1, const double2 &c, POW p, double &sv, double &sw, double &w0,
                                                 double &w1)
{
    const double e0 = av - va.x, e1 = av - va.y;
    const double d0 = __fma_rn(e0, e0, s0), d1 = __fma_rn(e1, e1, s1);
    w0 = shepard_w(d0, p); w1 = shepard_w(d1, p);
    sv = __fma_rn(w0, c.x, sv);
    sw = sw + w0;
    sv = __fma_rn(w1, c.y, sv);
    sw = sw + w1;
}
__device__ __forceinline__ bool wave_no_state_hit(bool second, double s0, double s1)
{
    return FRIRL_GBA_FASTPATH && __builtin_amdgcn_ballot_w64(!second || s0 == 0.0 || s1 == 0.0) == 0ull;
}

// Exact hits of the greedy sweeps, recorded OUTSIDE the hot loop.  An action that has an exact hit anywhere takes its conclusion from
// that rule and its Shepard sums are discarded (FIVEVagConcl_FRIRL_BestAct.c:89-93), so the sweeps run ONE branch-free loop body that lets
// rsq(0) poison the two sums of exactly that action, and only note the hit: d^2 = (av - va)^2 + s == 0 needs a zero state part s -- rare
// -- and then means av == va.  The side path scans the actions and takes an LDS atomic min (first hit = lowest rule index).  No per-action
// hit register, compare, branch or select.  The odd tail (r + 1 == R) gets a huge state part instead of a branch: its weight
// (1e300)^(-P/2) is 0 (P >= 3: underflow) or at most 1e-150 -- added to sums that are >= ~0.1 (d <= sqrt(nant)) it changes no bit.
#ifndef FRIRL_STEP_PREFETCH
#define FRIRL_STEP_PREFETCH 1
#endif
static constexpr int STEP_PREFETCH = FRIRL_STEP_PREFETCH;      // rule pairs requested ahead per lane in sweep_gba_q
__device__ __forceinline__ void note_state_hits(unsigned *hit_s, const double *ave_s, int A, double s0, double s1, const double2 &va, unsigned r)
{
    if (__builtin_amdgcn_ballot_w64(s0 == 0.0 || s1 == 0.0) != 0ull) {
        for (int a = 0; a < A; a++) {
            const double ava = ave_s[a];
            if (s0 == 0.0 && ava == va.x) atomicMin(&hit_s[a], r);
            if (s1 == 0.0 && ava == va.y) atomicMin(&hit_s[a], r + 1u);
        }
    }
}

template <int NANT, int AMAX, int BLOCK, class COLS, class POW>
__device__ int sweep_gba(const COLS &cols, const double *__restrict__ qcol, int R, const double (&qs)[NANT - 1 > 0 ? NANT - 1 : 1], POW p, int A,
                         GbaScratch<AMAX, BLOCK> &s)
{
    constexpr int NS = NANT - 1;
    double sv[AMAX], sw[AMAX], av[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; a++) { sv[a] = 0.0; sw[a] = 0.0; av[a] = wave_uniform(s.ave[a < A ? a : 0]); }      // action values in SGPRs
    if ((int)threadIdx.x < AMAX) s.hit[threadIdx.x] = FRIRL_HIP_NO_HIT;
    __syncthreads();
    const auto pk = pin_pow(p);
    for (int r = 2 * (int)threadIdx.x; r < R; r += 2 * BLOCK) {
        double s0 = 0.0, s1 = 0.0;
        if (NS > 0) sq_dist2<(NS > 0 ? NS : 1)>(cols, r, qs, s0, s1);
        double2 va = cols.pair(NS, r);
        double2 c = load_col2(qcol + r);
        const bool second = (r + 1 < R);
        if (!second) { s1 = NO_RULE_STATE_PART; va.y = 0.0; c.y = 0.0; }
        note_state_hits(s.hit, s.ave, A, s0, s1, va, (unsigned)r);
#pragma unroll
        for (int a = 0; a < AMAX; a++)
            if (a < A) concl_pair_nohit(av[a], va, s0, s1, c, pk, sv[a], sw[a]);
    }
    const int lane = threadIdx.x & (FRIRL_WAVE - 1), wave = threadIdx.x / FRIRL_WAVE;
    wave_sum_f64_n(sv);                              // the 2 AMAX butterflies in one pass (device_common.h)
    wave_sum_f64_n(sw);
#pragma unroll
    for (int a = 0; a < AMAX; a++)
        if (a < A && lane == 0) { s.v[wave][a] = sv[a]; s.w[wave][a] = sw[a]; }
    __syncthreads();
    if ((int)threadIdx.x < A) {
        const int a = threadIdx.x;
        double tv = s.v[0][a], tw = s.w[0][a];
        for (int w = 1; w < GbaScratch<AMAX, BLOCK>::WAVES; w++) { tv = tv + s.v[w][a]; tw = tw + s.w[w][a]; }
        const unsigned th = s.hit[a];                   // first exact hit of this action (note_state_hits), or NO_HIT
        s.actconc[a] = (th != FRIRL_HIP_NO_HIT) ? qcol[th] : tv / tw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int a = 1; a < A; a++) if (s.actconc[best] < s.actconc[a]) best = a;   // strict <: first maximum wins
        s.best = best;
    }
    __syncthreads();
    return s.best;
}

// Fused sweep of one environment step: the greedy sweep for the NEW state (sweep_gba) and the Q(s,a) sweep
// for the pending update (sweep_q) read the same rule base, which nothing modifies in between
// (frirl_episode.c:148 -> :159), so one pass over the slab serves both: 8*(nant+1) B per rule and step
// instead of twice that.  Per-lane accumulation order and the reduction tree are those of the two separate
// sweeps, so every result is bit-identical to running them one after the other.
// SAMES: the caller has found the pending observation's state part identical to the new observation's (the agent has not left its
// quantisation cell: 62 % of acrobot's steps, 84 % of mountaincar's).  The pending point (s, a) is then the new observation with action
// `apend` = a: Q(s, a) IS the greedy sweep's conclusion for that action over the same rule base -- its exact hit, its Shepard sums and
// (TRACK) its per-rule weights are taken from there, and the pending conclusion is not computed at all: 14.4 + 2 (nant - 1) FP64
// instructions per rule less.  Per-lane sums are the same operations in the same order as the separate pending sums; they are
// combined in the greedy sweep's reduction order (wave butterfly, then waves) instead of the block tree: Q(s, a) within ~1e-16 relative.
template <int NANT, int AMAX, int BLOCK, bool TRACK = false, bool SAMES = false, class COLS, class POW>
__device__ int sweep_gba_q(const COLS &cols, const double *__restrict__ qcol, int R, const double (&qs_in)[NANT - 1 > 0 ? NANT - 1 : 1],
                           const double (&q1_in)[NANT], POW p, int A, GbaScratch<AMAX, BLOCK> &s, BlockRed<BLOCK> &red, QResult &qres, double track_thr = 0.0,
                           SpreadCand *slot = nullptr, int apend = 0)
{
    constexpr int NS = NANT - 1;
    // the two observations are wave-uniform: in scalar registers they cost the sweep no VGPRs (a VOP3 reads one scalar operand)
    double qs[NS > 0 ? NS : 1], q1[NANT];
#pragma unroll
    for (int k = 0; k < (NS > 0 ? NS : 1); k++) qs[k] = wave_uniform(qs_in[k]);
#pragma unroll
    for (int k = 0; k < NANT; k++) q1[k] = wave_uniform(q1_in[k]);
    if (TRACK) slot[threadIdx.x].clear();          // `slot` = the workgroup's slot array, one entry per lane
    qres.tracked = TRACK;
    track_thr = wave_uniform(track_thr * SPREAD_PREFILTER_SLACK);
    double sv[AMAX], sw[AMAX], av[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; a++) { sv[a] = 0.0; sw[a] = 0.0; av[a] = wave_uniform(s.ave[a < A ? a : 0]); }      // action values in SGPRs
    if ((int)threadIdx.x < AMAX) s.hit[threadIdx.x] = FRIRL_HIP_NO_HIT;
    __syncthreads();
    const auto pk = pin_pow(p);
    unsigned qbest = FRIRL_HIP_NO_HIT;
    double qv = 0.0, qw = 0.0;
    // software prefetch: the loads of the NEXT pair of rules are issued before the ~100 FP64 instructions of the current
    // pair, so a wave does not sit on s_waitcnt at the top of every iteration
    // PD stages in flight (FRIRL_STEP_PREFETCH): with one stage a 256-thread workgroup has 9 KB requested ahead, ~45 KB per CU --
    // about what 8 TB/s x 1.5 us of loaded HBM latency needs chip-wide, nothing to spare
    constexpr int PD = STEP_PREFETCH;
    const QColBuf qb(qcol);
    auto ldq = [&](int r) {
        if constexpr (COLS::GLOBAL_Q && FRIRL_BUF_LOADS) return qb.load2(r);
        else return load_col2(qcol + r);
    };
    typename COLS::raw_t nraw[PD][NANT];
    double2 nc[PD];
#pragma unroll
    for (int u = 0; u < PD; u++) {
        const int r0 = 2 * (int)threadIdx.x + u * 2 * BLOCK;
        nc[u] = double2{0.0, 0.0};
        if (r0 < R) {
#pragma unroll
            for (int k = 0; k < NANT; k++) nraw[u][k] = cols.raw(k, r0);
            nc[u] = ldq(r0);
        }
    }
    double T = 0.0;
    // tracked form: wave-uniform trip count, because spread_track is a wave-level operation with ONE call site that every
    // lane of the wave reaches together; lanes whose last pair lies past R skip the arithmetic of that turn
    const int r_lim = TRACK ? wave_uniform_limit(R) : R;
    auto rule_pair = [&](const int r, typename COLS::raw_t (&sraw)[NANT], double2 &sc) {
        const bool live = !TRACK || r < R;
        const bool second = (r + 1 < R);
        double2 v[NANT];
        double2 c = {0.0, 0.0};
        double tw0 = 0.0, tw1 = 0.0;
        if (live) {
            // decode FIRST, then refill the stage: the index words die in the decode, so the loads of the next pair can land in the
            // same registers (requested before the decode they cost one register copy per column and iteration)
            c = sc;
#pragma unroll
            for (int k = 0; k < NANT; k++) v[k] = cols.decode(k, sraw[k]);
            if (r + PD * 2 * BLOCK < R) {
#pragma unroll
                for (int k = 0; k < NANT; k++) sraw[k] = cols.raw(k, r + PD * 2 * BLOCK);
                sc = ldq(r + PD * 2 * BLOCK);
            }
            if constexpr (!SAMES) {
                // (1) Q(s,a): full distance to the pending antecedents
                double d0 = q1[0] - v[0].x, d1 = q1[0] - v[0].y;
                double a0 = d0 * d0, a1 = d1 * d1;
#pragma unroll
                for (int k = 1; k < NANT; k++) {
                    d0 = q1[k] - v[k].x; d1 = q1[k] - v[k].y;
                    a0 = __fma_rn(d0, d0, a0); a1 = __fma_rn(d1, d1, a1);
                }
                if (!second) { a1 = NO_RULE_STATE_PART; c.y = 0.0; }
                q_pair(a0, a1, c, (unsigned)r, pk, qbest, qv, qw, tw0, tw1);
            }
        }
        if constexpr (!SAMES) { if (TRACK) spread_track(slot, T, track_thr, tw0, tw1, (unsigned)r, qw); }
        if (live) {
            // (2) greedy sweep for the new state: state part once, then every action
            double s0, s1;
            {
                double d0 = qs[0] - v[0].x, d1 = qs[0] - v[0].y;
                s0 = d0 * d0; s1 = d1 * d1;
#pragma unroll
                for (int k = 1; k < NS; k++) {
                    d0 = qs[k] - v[k].x; d1 = qs[k] - v[k].y;
                    s0 = __fma_rn(d0, d0, s0); s1 = __fma_rn(d1, d1, s1);
                }
            }
            double2 va = v[NS];
            if (!second) { s1 = NO_RULE_STATE_PART; va.y = 0.0; c.y = 0.0; }
            note_state_hits(s.hit, s.ave, A, s0, s1, va, (unsigned)r);
#pragma unroll
            for (int a = 0; a < AMAX; a++)
                if (a < A) {
                    if constexpr (SAMES && TRACK) {
                        double w0, w1;
                        concl_pair_nohit(av[a], va, s0, s1, c, pk, sv[a], sw[a], w0, w1);
                        if (a == apend) { tw0 = w0; tw1 = w1; qw = sw[a]; }      // uniform: the pending conclusion's weights and running sum
                    } else {
                        concl_pair_nohit(av[a], va, s0, s1, c, pk, sv[a], sw[a]);
                    }
                }
        }
        if constexpr (SAMES) { if (TRACK) spread_track(slot, T, track_thr, tw0, tw1, (unsigned)r, qw); }
    };
    for (int r = 2 * (int)threadIdx.x; r < r_lim; r += PD * 2 * BLOCK) {
#pragma unroll
        for (int u = 0; u < PD; u++) {
            const int ru = r + u * 2 * BLOCK;
            if (ru < r_lim) rule_pair(ru, nraw[u], nc[u]);       // tracked form: wave-uniform (r_lim is a multiple of 128)
        }
    }
    if constexpr (!SAMES) {
        qres.hit = blk_min<BLOCK>(qbest, red);
        qres.vagc = blk_sum<BLOCK>(qv, red);
        qres.ws = blk_sum<BLOCK>(qw, red);
    }
    const int lane = threadIdx.x & (FRIRL_WAVE - 1), wave = threadIdx.x / FRIRL_WAVE;
    wave_sum_f64_n(sv);                              // the 2 AMAX butterflies in one pass (device_common.h)
    wave_sum_f64_n(sw);
#pragma unroll
    for (int a = 0; a < AMAX; a++)
        if (a < A && lane == 0) { s.v[wave][a] = sv[a]; s.w[wave][a] = sw[a]; }
    __syncthreads();
    if ((int)threadIdx.x < A) {
        const int a = threadIdx.x;
        double tv = s.v[0][a], tw = s.w[0][a];
        for (int w = 1; w < GbaScratch<AMAX, BLOCK>::WAVES; w++) { tv = tv + s.v[w][a]; tw = tw + s.w[w][a]; }
        const unsigned th = s.hit[a];                   // first exact hit of this action (note_state_hits), or NO_HIT
        s.actconc[a] = (th != FRIRL_HIP_NO_HIT) ? qcol[th] : tv / tw;
        s.tv[a] = tv; s.tw[a] = tw;
    }
    __syncthreads();
    if constexpr (SAMES) { qres.hit = s.hit[apend]; qres.vagc = s.tv[apend]; qres.ws = s.tw[apend]; }
    if (threadIdx.x == 0) {
        int best = 0;
        for (int a = 1; a < A; a++) if (s.actconc[best] < s.actconc[a]) best = a;
        s.best = best;
    }
    __syncthreads();
    return s.best;
}

// Many-action greedy sweep with EVERY action in registers (9 <= A <= AMAX = 24; cartpole: 21).  The action-parallel form below
// (sweep_gba_wide) repeats the decode and the state part of every rule pair in each of its four waves (~18 % of its issue slots) and
// relies on the vector L1 to absorb the fourfold column reads.  Here the waves take different rule pairs, as in sweep_gba_q, and one lane
// keeps all A accumulator pairs: 4 A VGPRs (84 at A = 21) -- affordable because nothing else per action lives in vector registers:
//  * the action VE values sit in SGPRs (wave-uniform; a VOP3 reads one scalar operand);
//  * exact hits never touch the sums: an action that has an exact hit anywhere takes its conclusion from that rule and its Shepard sums
//    are discarded (FIVEVagConcl_FRIRL_BestAct.c:89-93), so the hot loop lets 1/0 poison the sums of exactly that action and only
//    RECORDS the hit -- in a rare side path (some lane's rule has a zero state part) that scans the actions for av[a] == va and
//    takes an LDS atomic min.  One loop body, no per-action compare, branch or select;
//  * the odd tail (r + 1 == R) gets a huge state part: its weight underflows to exactly 0, adding +0 to both sums.
// Actions are unrolled in groups of three behind one scalar test per group (A is rounded up to a multiple of three with copies of the
// last action, whose sums nobody reads): six independent FP64 chains per group.  Per-lane sums are in rule order, the combine is the
// fixed butterfly + wave order of sweep_gba.  WITH_Q adds the Q(s,a) sums of the fused step with sweep_q's lane mapping.
template <int NANT, int AMAX, int BLOCK, bool WITH_Q, class COLS, class POW>
__device__ int sweep_gba_many(const COLS &cols, const double *__restrict__ qcol, int R, const double (&qs)[NANT - 1 > 0 ? NANT - 1 : 1],
                              const double (&q1)[NANT], POW p, int A, GbaScratch<AMAX, BLOCK> &s, BlockRed<BLOCK> &red, QResult *qres)
{
    constexpr int NS = NANT - 1;
    constexpr int G = 3;
    static_assert(AMAX % G == 0 && AMAX % 12 == 0, "groups of three actions, reduction passes of twelve");
    const int lane = threadIdx.x & (FRIRL_WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / FRIRL_WAVE));
    double sv[AMAX], sw[AMAX], av[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; a++) { sv[a] = 0.0; sw[a] = 0.0; av[a] = wave_uniform(s.ave[a < A ? a : A - 1]); }
    if ((int)threadIdx.x < AMAX) s.hit[threadIdx.x] = FRIRL_HIP_NO_HIT;
    __syncthreads();
    unsigned qbest = FRIRL_HIP_NO_HIT;
    double qv = 0.0, qw = 0.0;
    const auto pk = pin_pow(p);
    const QColBuf qb(qcol);
    auto ldq = [&](int r) {
        if constexpr (COLS::GLOBAL_Q && FRIRL_BUF_LOADS) return qb.load2(r);
        else return load_col2(qcol + r);
    };
    typename COLS::raw_t nraw[NANT];
    double2 nc = {0.0, 0.0};
    {
        const int r0 = 2 * (int)threadIdx.x;
        if (r0 < R) {
#pragma unroll
            for (int k = 0; k < NANT; k++) nraw[k] = cols.raw(k, r0);
            nc = ldq(r0);
        }
    }
    for (int r = 2 * (int)threadIdx.x; r < R; r += 2 * BLOCK) {
        const bool second = (r + 1 < R);
        double2 v[NANT];
        double2 c = nc;
#pragma unroll
        for (int k = 0; k < NANT; k++) v[k] = cols.decode(k, nraw[k]);        // decode first, then refill (see sweep_gba_q)
        if (r + 2 * BLOCK < R) {
#pragma unroll
            for (int k = 0; k < NANT; k++) nraw[k] = cols.raw(k, r + 2 * BLOCK);
            nc = ldq(r + 2 * BLOCK);
        }
        if (WITH_Q) {
            double d0 = q1[0] - v[0].x, d1 = q1[0] - v[0].y;
            double a0 = d0 * d0, a1 = d1 * d1;
#pragma unroll
            for (int k = 1; k < NANT; k++) {
                d0 = q1[k] - v[k].x; d1 = q1[k] - v[k].y;
                a0 = __fma_rn(d0, d0, a0); a1 = __fma_rn(d1, d1, a1);
            }
            if (!second) { a1 = NO_RULE_STATE_PART; c.y = 0.0; }
            double tw0, tw1;
            q_pair(a0, a1, c, (unsigned)r, pk, qbest, qv, qw, tw0, tw1);
        }
        double s0, s1;
        {
            double d0 = qs[0] - v[0].x, d1 = qs[0] - v[0].y;
            s0 = d0 * d0; s1 = d1 * d1;
#pragma unroll
            for (int k = 1; k < NS; k++) {
                d0 = qs[k] - v[k].x; d1 = qs[k] - v[k].y;
                s0 = __fma_rn(d0, d0, s0); s1 = __fma_rn(d1, d1, s1);
            }
        }
        double2 va = v[NS];
        if (!second) { s1 = NO_RULE_STATE_PART; va.y = 0.0; c.y = 0.0; }
        note_state_hits(s.hit, s.ave, A, s0, s1, va, (unsigned)r);
#pragma unroll
        for (int g = 0; g < AMAX / G; g++) {
            if (G * g < A) {
#pragma unroll
                for (int j = 0; j < G; j++) concl_pair_nohit(av[G * g + j], va, s0, s1, c, pk, sv[G * g + j], sw[G * g + j]);
            }
        }
    }
    if (WITH_Q) {
        qres->hit = blk_min<BLOCK>(qbest, red);
        qres->vagc = blk_sum<BLOCK>(qv, red);
        qres->ws = blk_sum<BLOCK>(qw, red);
        qres->tracked = false;
    }
    // the 2 A butterflies in passes of 12 sums (one pass for all of them would need 48 more VGPRs than the sweep has; padded actions
    // ride along)
#pragma unroll
    for (int a0 = 0; a0 < AMAX; a0 += 12) {
        if (a0 < A) {
            double tv[12], tw[12];
#pragma unroll
            for (int i = 0; i < 12; i++) { tv[i] = sv[a0 + i]; tw[i] = sw[a0 + i]; }
            wave_sum_f64_n(tv);
            wave_sum_f64_n(tw);
#pragma unroll
            for (int i = 0; i < 12; i++) { sv[a0 + i] = tv[i]; sw[a0 + i] = tw[i]; }
        }
    }
#pragma unroll
    for (int a = 0; a < AMAX; a++)
        if (a < A && lane == 0) { s.v[wave][a] = sv[a]; s.w[wave][a] = sw[a]; }
    __syncthreads();
    if ((int)threadIdx.x < A) {
        const int a = threadIdx.x;
        double tv = s.v[0][a], tw = s.w[0][a];
        for (int w = 1; w < GbaScratch<AMAX, BLOCK>::WAVES; w++) { tv = tv + s.v[w][a]; tw = tw + s.w[w][a]; }
        const unsigned th = s.hit[a];
        s.actconc[a] = (th != FRIRL_HIP_NO_HIT) ? qcol[th] : tv / tw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int a = 1; a < A; a++) if (s.actconc[best] < s.actconc[a]) best = a;   // strict <: first maximum wins
        s.best = best;
    }
    __syncthreads();
    return s.best;
}

#ifndef FRIRL_WIDE_SYNC
#define FRIRL_WIDE_SYNC 8
#endif
static constexpr int WIDE_SYNC = FRIRL_WIDE_SYNC;      // iterations between the barriers of sweep_gba_wide (0 = none; power of two)

// (A branch-free form of the conclusion terms -- weight forced to 0 for exact hits / slots past the end instead of `if` -- was measured at
//  cartpole's 21-action step: 2.73 -> 2.88 ms with the index store: the selects cost more VALU issue than the branches.  What did help:
//  plain instead of non-temporal loads (the four action-parallel waves read the same lines: 84 % L1 hits), a barrier every 8 iterations
//  that keeps the waves inside the L1 window, and 168 instead of 128 VGPRs (no spills; only three workgroups fit a CU beside 40 KB of
//  tables anyway): 22.8 -> 19.9 ms per step of 32 768 x 32 768, f64 columns 6.4 -> 3.0 ms per 4096 environments; PMC: VALU busy 70 %.)
// Many-action form of the greedy sweep (A > 8, e.g. cartpole's 21 actions).  Keeping A accumulator pairs per
// lane costs ~250 VGPRs at A = 21 (one wave per SIMD, every dependent FP64 chain exposed).  Here the ACTIONS
// are split over the waves of the workgroup instead: wave w evaluates actions [w*ag, (w+1)*ag) with
// ag = ceil(A / WAVES) <= AG for ALL rules (64 lanes x 2 rules per iteration); the waves read the same rule
// columns almost simultaneously, so HBM still sees each rule once (L1/L2 absorb the repeats) while the
// register footprint drops to AG accumulator pairs.  The optional Q(s,a) sums of the fused episode step are
// spread over the waves by iteration.  Sums per action are one wave butterfly (deterministic).
template <int NANT, int AG, int AMAX, int BLOCK, bool WITH_Q, bool TRACK = false, class COLS, class POW>
__device__ int sweep_gba_wide(const COLS &cols, const double *__restrict__ qcol, int R, const double (&qs)[NANT - 1 > 0 ? NANT - 1 : 1],
                              const double (&q1)[NANT], POW p, int A, GbaScratch<AMAX, BLOCK> &s, BlockRed<BLOCK> &red, QResult *qres, double track_thr = 0.0,
                              SpreadCand *slot = nullptr)
{
    constexpr int NS = NANT - 1;
    constexpr int WAVES = BLOCK / FRIRL_WAVE;
    if (TRACK) slot[threadIdx.x].clear();          // `slot` = the workgroup's slot array, one entry per lane
    track_thr = wave_uniform(track_thr * SPREAD_PREFILTER_SLACK);
    // `wave` as a scalar: the compiler cannot see that threadIdx.x / 64 is wave-uniform and would mask every per-wave decision with exec
    const int lane = threadIdx.x & (FRIRL_WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / FRIRL_WAVE));
    // actions per wave: as even as A allows (21 over 4 waves: 6 + 5 + 5 + 5, not 6 + 6 + 6 + 3).  Every action is summed by ONE wave over
    // all rules in rule order, so results do not depend on the assignment.  (Rotating the wave that carries the extra action with a
    // hash of blockIdx -- the waves of a workgroup land on the four SIMDs in launch order -- measured no difference: 2.41 vs 2.42 ms.)
    const int vw = wave;
    const int a_lo = A / WAVES, a_rem = A % WAVES;
    const int a_begin = vw * a_lo + (vw < a_rem ? vw : a_rem);
    const int na = a_lo + (vw < a_rem ? 1 : 0);        // actions of this wave (<= AG)
    double sv[AG], sw[AG], av[AG];
    unsigned sh[AG];
#pragma unroll
    for (int j = 0; j < AG; j++) { sv[j] = 0.0; sw[j] = 0.0; sh[j] = FRIRL_HIP_NO_HIT; av[j] = (j < na) ? s.ave[a_begin + j] : 0.0; }
    unsigned qbest = FRIRL_HIP_NO_HIT;
    double qv = 0.0, qw = 0.0;
    int it = 0;
    // software prefetch of the next pair of rules (as in sweep_gba_q): 4 waves per SIMD here, every s_waitcnt shows
    typename COLS::raw_t nraw[NANT];
    double2 nc = {0.0, 0.0};
    if (2 * lane < R) {
#pragma unroll
        for (int k = 0; k < NANT; k++) nraw[k] = cols.raw_shared(k, 2 * lane);
        nc = load_col2_shared(qcol + 2 * lane);
    }
    double T = 0.0;
    const int r_lim = (WITH_Q && TRACK) ? wave_uniform_limit(R) : R;      // see sweep_gba_q
    // The rule loop, specialised for the number of actions NA of this wave (a generic lambda instantiated for 0..AG and selected once
    // per wave): with the run-time bound `j < na` inside the loop every action sits behind its own exec-mask test, which also keeps the
    // scheduler from interleaving the FP64 chains of different actions.  Waves of one workgroup may run different instances; they meet
    // at the same barriers (same trip count).
    const auto pk = pin_pow(p);
    auto rule_loop = [&](auto na_c) {
    constexpr int NA = decltype(na_c)::value;
    for (int r = 2 * lane; r < r_lim; r += 2 * FRIRL_WAVE, it++) {
        // the waves read the same lines: a barrier every few iterations keeps them within the window the vector L1 still holds
        // (trip count and `it` are the same for every wave of the workgroup)
        if (WIDE_SYNC > 0 && (it & (WIDE_SYNC - 1)) == WIDE_SYNC - 1) __syncthreads();
        const bool live = !(WITH_Q && TRACK) || r < R;
        const bool second = (r + 1 < R);
        double2 v[NANT];
        double2 c = {0.0, 0.0};
        if (live) {
            typename COLS::raw_t raw[NANT];
#pragma unroll
            for (int k = 0; k < NANT; k++) raw[k] = nraw[k];
            c = nc;
            if (r + 2 * FRIRL_WAVE < R) {
#pragma unroll
                for (int k = 0; k < NANT; k++) nraw[k] = cols.raw_shared(k, r + 2 * FRIRL_WAVE);
                nc = load_col2_shared(qcol + r + 2 * FRIRL_WAVE);
            }
#pragma unroll
            for (int k = 0; k < NANT; k++) v[k] = cols.decode(k, raw[k]);
        }
        if (WITH_Q && (it % WAVES) == wave) {           // wave-uniform: the Q(s,a) sums are spread over the waves by iteration
            double tw0 = 0.0, tw1 = 0.0;
            if (live) {
                double d0 = q1[0] - v[0].x, d1 = q1[0] - v[0].y;
                double a0 = d0 * d0, a1 = d1 * d1;
#pragma unroll
                for (int k = 1; k < NANT; k++) {
                    d0 = q1[k] - v[k].x; d1 = q1[k] - v[k].y;
                    a0 = __fma_rn(d0, d0, a0); a1 = __fma_rn(d1, d1, a1);
                }
                if (!second) { a1 = NO_RULE_STATE_PART; c.y = 0.0; }
                q_pair(a0, a1, c, (unsigned)r, pk, qbest, qv, qw, tw0, tw1);
            }
            if (TRACK) spread_track(slot, T, track_thr, tw0, tw1, (unsigned)r, qw);
        }
        if (live) {
            double s0, s1;
            {
                double d0 = qs[0] - v[0].x, d1 = qs[0] - v[0].y;
                s0 = d0 * d0; s1 = d1 * d1;
#pragma unroll
                for (int k = 1; k < NS; k++) {
                    d0 = qs[k] - v[k].x; d1 = qs[k] - v[k].y;
                    s0 = __fma_rn(d0, d0, s0); s1 = __fma_rn(d1, d1, s1);
                }
            }
            const double2 va = v[NS];
            if (wave_no_state_hit(second, s0, s1)) {
#pragma unroll
                for (int j = 0; j < NA; j++) concl_pair_nohit(av[j], va, s0, s1, c, pk, sv[j], sw[j]);
            } else {
#pragma unroll
                for (int j = 0; j < NA; j++) {
                    const double e0 = av[j] - va.x, e1 = av[j] - va.y;
                    const double d0 = __fma_rn(e0, e0, s0), d1 = __fma_rn(e1, e1, s1);
                    if (d0 == 0.0) sh[j] = min(sh[j], (unsigned)r);
                    else { const double wi = shepard_w(d0, pk); sv[j] = __fma_rn(wi, c.x, sv[j]); sw[j] = sw[j] + wi; }
                    if (second) {
                        if (d1 == 0.0) sh[j] = min(sh[j], (unsigned)(r + 1));
                        else { const double wi = shepard_w(d1, pk); sv[j] = __fma_rn(wi, c.y, sv[j]); sw[j] = sw[j] + wi; }
                    }
                }
            }
        }
    }
    };
    static_assert(AG == 8, "the dispatch below enumerates 0..8 actions per wave");
    switch (na) {
        case 0: rule_loop(std::integral_constant<int, 0>()); break;
        case 1: rule_loop(std::integral_constant<int, 1>()); break;
        case 2: rule_loop(std::integral_constant<int, 2>()); break;
        case 3: rule_loop(std::integral_constant<int, 3>()); break;
        case 4: rule_loop(std::integral_constant<int, 4>()); break;
        case 5: rule_loop(std::integral_constant<int, 5>()); break;
        case 6: rule_loop(std::integral_constant<int, 6>()); break;
        case 7: rule_loop(std::integral_constant<int, 7>()); break;
        default: rule_loop(std::integral_constant<int, 8>()); break;
    }
#pragma unroll
    for (int j = 0; j < AG; j++) {
        if (j < na) {
            const double tv = wave_sum_f64(sv[j]), tw = wave_sum_f64(sw[j]);
            const unsigned th = wave_min_u32(sh[j]);
            if (lane == 0) s.actconc[a_begin + j] = (th != FRIRL_HIP_NO_HIT) ? qcol[th] : tv / tw;
        }
    }
    if (WITH_Q) {
        qres->hit = blk_min<BLOCK>(qbest, red);
        qres->vagc = blk_sum<BLOCK>(qv, red);
        qres->ws = blk_sum<BLOCK>(qw, red);
        qres->tracked = TRACK;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int a = 1; a < A; a++) if (s.actconc[best] < s.actconc[a]) best = a;
        s.best = best;
    }
    __syncthreads();
    return s.best;
}

}  // namespace frirl
