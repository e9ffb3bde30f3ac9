// device_common.h -- shared device helpers of the gfx950 FRIRL/FIVE kernels.
// CDNA4 only: 64-lane wavefronts, 256-thread workgroups (4 waves, one per SIMD).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/frirl_hip.h"

#define FRIRL_WAVE 64
#define FRIRL_BLOCK 256
#define FRIRL_WAVES_PER_BLOCK (FRIRL_BLOCK / FRIRL_WAVE)

namespace frirl {

// Nearest index on a fixed-step universe -- reference src/inl/min.inl:71-92
// (get_vag_abs_min_i_fixres): C truncation of (point - u[0]) / div, clamp, then the nearer of
// u[low], u[low+1] with ties to `low`.  The reference's read of u[low+1] one past the row when
// low == len-1 is guarded here (preserve the result, not the stray read).
__device__ __forceinline__ unsigned snap_index(const double *__restrict__ uni, int len, double point, double div)
{
    const int low = (int)((point - uni[0]) / div);
    if (low < 0) return 0u;
    if (low >= len) return (unsigned)(len - 1);
    if (low + 1 >= len) return (unsigned)low;
    const double d1 = uni[low] - point;
    const double d2 = uni[low + 1] - point;
    return (fabs(d1) <= fabs(d2)) ? (unsigned)low : (unsigned)(low + 1);
}

// Step of universe k -- reference src/five/FIVEInit.c:244-248 (udivs = (u_last - u_first) / (U-1)).
__device__ __forceinline__ double universe_div(const double *__restrict__ uni, int U)
{
    return (uni[U - 1] - uni[0]) / (double)(U - 1);
}

// VE value of the observation in dimension k: ve[k][snap(x_k)]  (five_rule_distance.c:75,80).
__device__ __forceinline__ double observe_ve(const double *__restrict__ u, const double *__restrict__ ve, int U, int k, double xk)
{
    const double *uni = u + (size_t)k * U;
    return ve[(size_t)k * U + snap_index(uni, U, xk, universe_div(uni, U))];
}

// Two 16-bit universe indices packed in one word -> the LDS byte addresses of their table entries, ONE instruction each:
// v_mad_u32_u16 takes the low or the high half of the word (op_sel), multiplies by 8 and adds the table's LDS address (a scalar
// operand).  The compiler's own sequence is extract (v_and / v_bfe) + v_lshl_add: 4 instead of 2 vector instructions per pair of
// antecedents, ~10 % of the acrobot step's instructions (tools/exp/mad_u16.hip checks the encoding on the device).
#ifndef FRIRL_DECODE_MAD
#define FRIRL_DECODE_MAD 1
#endif
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ double2 lds_table_pair(const double *tab_k, uint32_t w)
{
    double2 v;
#if FRIRL_DECODE_MAD
    const uint32_t base = (uint32_t)(uintptr_t)(lds_cdouble *)tab_k;
    uint32_t a0, a1;
    asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[0,0,0,0]" : "=v"(a0) : "v"(w), "s"(base));
    asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(w), "s"(base));
    v.x = *(lds_cdouble *)(uintptr_t)a0;
    v.y = *(lds_cdouble *)(uintptr_t)a1;
#else
    v.x = tab_k[w & 0xFFFFu];
    v.y = tab_k[w >> 16];
#endif
    return v;
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, off, FRIRL_WAVE);
        v = (o < v) ? o : v;
    }
    return v;
}

// Block-wide minimum (all threads get the result); `scratch` holds FRIRL_WAVES_PER_BLOCK words.
__device__ __forceinline__ unsigned block_min_u32(unsigned v, unsigned *scratch)
{
    v = wave_min_u32(v);
    const int lane = threadIdx.x & (FRIRL_WAVE - 1), wave = threadIdx.x / FRIRL_WAVE;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    unsigned m = scratch[0];
#pragma unroll
    for (int w = 1; w < FRIRL_WAVES_PER_BLOCK; w++) m = (scratch[w] < m) ? scratch[w] : m;
    __syncthreads();
    return m;
}

// Fixed-shape sum: butterfly inside the wave (every lane ends with the same bits), then the
// wave partials are added in wave order.  Deterministic run to run; no float atomics.
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_xor(v, off, FRIRL_WAVE);
    return v;
}

// N independent sums in ONE pass of the butterfly: the N cross-lane moves of a step are issued back to back and waited for once.
// One after the other (wave_sum_f64 in a loop) every step pays the full ds_bpermute latency -- ~12 us for the 42 sums of a 21-action
// greedy sweep, measured as the largest part of a single-rule-base step at ~200 rules.  Same additions in the same order per value.
template <int N>
__device__ __forceinline__ void wave_sum_f64_n(double (&v)[N])
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double t[N];
#pragma unroll
        for (int i = 0; i < N; i++) t[i] = __shfl_xor(v[i], off, FRIRL_WAVE);
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = v[i] + t[i];
    }
}

__device__ __forceinline__ double block_sum_f64(double v, double *scratch)
{
    v = wave_sum_f64(v);
    const int lane = threadIdx.x & (FRIRL_WAVE - 1), wave = threadIdx.x / FRIRL_WAVE;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double s = scratch[0];
#pragma unroll
    for (int w = 1; w < FRIRL_WAVES_PER_BLOCK; w++) s = s + scratch[w];
    __syncthreads();
    return s;
}

}  // namespace frirl

// ---- host-side helpers shared by the extern "C" entry points ---------------------------------
namespace frirl_host {
void set_error(const char *fmt, ...);
int check_device();                       // 0 or FRIRL_HIP_ENODEV
int check_launch(const char *what);       // hipGetLastError -> code
int check_tables(const frirl_hip_tables *t);
int check_rulebases(const frirl_hip_tables *t, const frirl_hip_rulebases *b);
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// The library-owned batches switch to their own device; the caller's current device is put back on every exit path.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// Experiment / test switches (frirl_hip_set_option).  Defaults (0 / -1 = the shipped configuration) are read ONCE from
// the FRIRL_HIP_* environment variables when the library first needs them, never per launch.
struct Options {
    int no_uidx;          // 1: ignore the 16-bit index mirror, stream the f64 columns           (FRIRL_HIP_NO_UIDX)
    int rd_unroll;        // rule-distance scan: column sets in flight per lane, 0 = shipped     (FRIRL_HIP_RD_UNROLL)
    int rd_chunk;         // rule-distance scan: rules per workgroup, 0 = shipped                (FRIRL_HIP_RD_CHUNK)
    int rd_nt;            // rule-distance scan: non-temporal variant, -1 = shipped              (FRIRL_HIP_RD_NT)
    int rd_persist;       // compressed rule-distance scan: -1 = by table size, 0 = one workgroup per item, 1 = persistent (FRIRL_HIP_RD_PERSIST)
    int rd_order;         // rule-distance scan item order: 0 = chunk index fastest (shipped), 1 = environment fastest (FRIRL_HIP_RD_ORDER)
    int step_wave;        // episode step: 1 = one wave per environment, 0 = 256 threads, -1 = by shape (FRIRL_HIP_STEP_WAVE)
    int step_track;       // episode step: spread candidates tracked in the fused sweep: 1 / 0, -1 = large rule bases only (FRIRL_HIP_STEP_TRACK)
    int lanes_slices;     // lane groups: rule slices per conclusion, 0 = by shape               (FRIRL_HIP_LANES_SLICES)
    int lanes_wpe;        // lane groups: waves per SIMD, 0 = by shape                           (FRIRL_HIP_LANES_WPE)
    int rollout_group;    // shared-base roll-out: lanes per environment, 0 = by shape           (FRIRL_HIP_ROLLOUT_GROUP)
    int rollout_slices;   // shared-base roll-out: rule slices, 0 = by shape                     (FRIRL_HIP_ROLLOUT_SLICES)
    int rollout_resident; // shared-base roll-out: 0 = never the LDS-resident queue-fed form (rollout.hip), -1 = by shape (FRIRL_HIP_ROLLOUT_RESIDENT)
    int rollout_cap;      // resident roll-out: steps before a long episode is parked for the latency form, 0 = max_steps / 6 (FRIRL_HIP_ROLLOUT_CAP)
    int rollout_pair;     // resident roll-out, one environment per wave: 0 = the one-wave form instead of sweeper + speculative stepper (FRIRL_HIP_ROLLOUT_PAIR)
    int rollout_wps;      // resident roll-out: persistent waves per SIMD, 1 ... 4 (0 = 2, or 4 for queue-fed launches) (FRIRL_HIP_ROLLOUT_WPS)
    int learn_slices;     // persistent learner: lanes per agent 4 / 16 / 64, 0 = by the number of live agents (FRIRL_HIP_LEARN_SLICES)
    int learn_alone;      // learner launch plan: cost factor (tenths) of a wave that has its SIMD to itself, 0 = 20 (FRIRL_HIP_LEARN_ALONE)
    int learn_persistent; // 0: frirl_hip_learn_supported answers no (callers fall back to one episode per launch) (FRIRL_HIP_LEARN_PERSISTENT)
    int multi_loopback;   // 1: frirl_hip_multi_create builds LOGICAL shards on the current device with the loop-back transport (tests) (FRIRL_HIP_MULTI_LOOPBACK)
    int mirror_sync;      // single-agent fused step: 1 = wait with hipStreamSynchronize instead of polling the completion flag (FRIRL_HIP_MIRROR_SYNC)
    int no_many;          // 9..24 actions: 1 = action-parallel waves (sweep_gba_wide) instead of all actions in registers (FRIRL_HIP_NO_MANY)
};
const Options &opts();
}  // namespace frirl_host
