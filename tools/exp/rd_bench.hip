// rd_bench.hip -- stand-alone A/B harness for variants of the compressed rule-distance scan (round 2).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o rd_bench rd_bench.hip
//   ./rd_bench <cfg2|cfg3|cfg4|cfg5> [E-override]
// Every variant is checked bit for bit against the round-1 kernel (V0) on the same inputs before it is timed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define NO_HIT 0xFFFFFFFFu

__device__ __forceinline__ unsigned snap_index(const double *__restrict__ uni, int len, double point, double div)
{
    const int low = (int)((point - uni[0]) / div);
    if (low < 0) return 0u;
    if (low >= len) return (unsigned)(len - 1);
    if (low + 1 >= len) return (unsigned)low;
    const double d1 = uni[low] - point, d2 = uni[low + 1] - point;
    return (fabs(d1) <= fabs(d2)) ? (unsigned)low : (unsigned)(low + 1);
}
__device__ __forceinline__ double observe_ve(const double *__restrict__ u, const double *__restrict__ ve, int U, int k, double xk)
{
    const double *uni = u + (size_t)k * U;
    return ve[(size_t)k * U + snap_index(uni, U, xk, (uni[U - 1] - uni[0]) / (double)(U - 1))];
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)v, off, 64); v = (o < v) ? o : v; }
    return v;
}

// ---- V0: the round-1 kernel (one workgroup per (env, 2048-rule chunk), VE table refilled per workgroup) ---------------
template <int NANT, int UNROLL, int BLOCK>
__global__ __launch_bounds__(BLOCK) void v0_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U, const uint16_t *__restrict__ uidx,
                                                    const int32_t *__restrict__ nrules, int maxR, const double *__restrict__ x, double *__restrict__ dists,
                                                    uint32_t *__restrict__ hit, int rules_per_block)
{
    extern __shared__ double tab_s[];
    const int e = blockIdx.x;
    const int R = nrules[e];
    const int r0 = blockIdx.y * rules_per_block;
    if (r0 >= R) return;
    int r_end = r0 + rules_per_block;
    if (r_end > R) r_end = R;
    __shared__ double q_s[NANT];
    __shared__ unsigned red_s[BLOCK / 64];
    for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    if (threadIdx.x < NANT) q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, x[(size_t)e * NANT + threadIdx.x]);
    __syncthreads();
    double q[NANT];
#pragma unroll
    for (int k = 0; k < NANT; k++) q[k] = q_s[k];
    const uint16_t *__restrict__ base = uidx + (size_t)e * NANT * maxR;
    double *__restrict__ out = dists + (size_t)e * maxR;
    unsigned best = NO_HIT;
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
                double d0 = q[0] - tab_s[w[j][0] & 0xFFFFu], d1 = q[0] - tab_s[w[j][0] >> 16];
                double a0 = d0 * d0, a1 = d1 * d1;
#pragma unroll
                for (int k = 1; k < NANT; k++) {
                    d0 = q[k] - tab_s[k * U + (w[j][k] & 0xFFFFu)];
                    d1 = q[k] - tab_s[k * U + (w[j][k] >> 16)];
                    const double s0 = d0 * d0, s1 = d1 * d1;
                    a0 = a0 + s0;
                    a1 = a1 + s1;
                }
                double2 d;
                d.x = __dsqrt_rn(a0);
                d.y = __dsqrt_rn(a1);
                __builtin_nontemporal_store(d.x, out + rr);
                __builtin_nontemporal_store(d.y, out + rr + 1);
                if (d.y == 0.0 && rr + 1 < R) best = min(best, (unsigned)(rr + 1));
                if (d.x == 0.0) best = min(best, (unsigned)rr);
            }
        }
    }
    best = wave_min_u32(best);
    if ((threadIdx.x & 63) == 0) red_s[threadIdx.x / 64] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = red_s[0];
        for (int w = 1; w < BLOCK / 64; w++) m = red_s[w] < m ? red_s[w] : m;
        if (m != NO_HIT) atomicMin(&hit[e], m);
    }
}

// ---- V1: persistent workgroups over (env, chunk) items ----------------------------------------------------------------
// SQ: the LDS table holds sq[k][i] = (q_k - ve[k][i])^2 of the CURRENT environment (same sub, same mul => same bits), rebuilt
//     when the workgroup moves to another environment; the per-rule work is NANT gathers + NANT-1 adds + sqrt.
// !SQ: the LDS table is the VE table, filled once per workgroup.
// W: rules per lane per column load (2 = one u32, 4 = one u64).  UNR column sets in flight.  PF: the loads of the next
// batch are issued before the arithmetic of the current one.
template <int W> struct IdxVec;
template <> struct IdxVec<2> { using type = uint32_t; };
template <> struct IdxVec<4> { using type = uint2; };
template <> struct IdxVec<8> { using type = uint4; };

template <int W>
__device__ __forceinline__ typename IdxVec<W>::type ld_idx(const uint16_t *p)
{
    using T = typename IdxVec<W>::type;
    if constexpr (W == 2) return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p));
    else if constexpr (W == 4) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
        T v; v.x = __builtin_nontemporal_load(q); v.y = __builtin_nontemporal_load(q + 1); return v;
    } else {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
        T v; v.x = __builtin_nontemporal_load(q); v.y = __builtin_nontemporal_load(q + 1); v.z = __builtin_nontemporal_load(q + 2); v.w = __builtin_nontemporal_load(q + 3); return v;
    }
}
template <int W>
__device__ __forceinline__ unsigned idx_of(const typename IdxVec<W>::type &v, int i)
{
    if constexpr (W == 2) return (i == 0) ? (v & 0xFFFFu) : (v >> 16);
    else if constexpr (W == 4) { const uint32_t w = (i < 2) ? v.x : v.y; return (i & 1) ? (w >> 16) : (w & 0xFFFFu); }
    else { const uint32_t w = (i < 2) ? v.x : (i < 4 ? v.y : (i < 6 ? v.z : v.w)); return (i & 1) ? (w >> 16) : (w & 0xFFFFu); }
}

template <int NANT, int BLOCK, int W, int UNR, bool SQ, bool PF>
__global__ __launch_bounds__(BLOCK) void v1_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U, const uint16_t *__restrict__ uidx,
                                                    const int32_t *__restrict__ nrules, int maxR, const double *__restrict__ x, double *__restrict__ dists,
                                                    uint32_t *__restrict__ hit, int chunk, int chunks_per_env, int nitems)
{
    extern __shared__ double tab_s[];            // [NANT][U]
    __shared__ double q_s[NANT];
    __shared__ unsigned red_s[BLOCK / 64];
    using V = typename IdxVec<W>::type;
    constexpr int STEP = BLOCK * W;              // rules per column set of the workgroup
    int cur_e = -1;
    double q[NANT];
    if (!SQ) {
        for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    }
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int e = item / chunks_per_env, c = item - e * chunks_per_env;
        const int R = nrules[e];
        const int r0 = c * chunk;
        if (r0 >= R) continue;                   // uniform
        int r_end = r0 + chunk;
        if (r_end > R) r_end = R;
        const uint16_t *__restrict__ base = uidx + (size_t)e * NANT * maxR;
        double *__restrict__ out = dists + (size_t)e * maxR;
        // first batch of index loads: independent of the table, issued before it is (re)built
        V w[UNR][NANT];
        int r = r0 + W * (int)threadIdx.x;
#pragma unroll
        for (int j = 0; j < UNR; j++) {
            const int rr = r + j * STEP;
            if (rr < r_end) {
#pragma unroll
                for (int k = 0; k < NANT; k++) w[j][k] = ld_idx<W>(base + (size_t)k * maxR + rr);
            }
        }
        if (e != cur_e) {
            __syncthreads();                     // everyone is done with the previous table / q_s
            if (threadIdx.x < NANT) q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, x[(size_t)e * NANT + threadIdx.x]);
            __syncthreads();
            if (SQ) {
                for (int i = threadIdx.x; i < NANT * U; i += BLOCK) {
                    const int k = i / U;
                    const double d = q_s[k] - ve[i];
                    tab_s[i] = d * d;
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int k = 0; k < NANT; k++) q[k] = q_s[k];
            }
            cur_e = e;
        }
        unsigned best = NO_HIT;
        for (; r < r_end; r += STEP * UNR) {
            V cw[UNR][NANT];
#pragma unroll
            for (int j = 0; j < UNR; j++)
#pragma unroll
                for (int k = 0; k < NANT; k++) cw[j][k] = w[j][k];
            if (PF) {
                const int rn = r + STEP * UNR;
#pragma unroll
                for (int j = 0; j < UNR; j++) {
                    const int rr = rn + j * STEP;
                    if (rr < r_end) {
#pragma unroll
                        for (int k = 0; k < NANT; k++) w[j][k] = ld_idx<W>(base + (size_t)k * maxR + rr);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < UNR; j++) {
                const int rr = r + j * STEP;
                if (rr < r_end) {
                    double a[W];
#pragma unroll
                    for (int i = 0; i < W; i++) {
                        if (SQ) {
                            double s = tab_s[idx_of<W>(cw[j][0], i)];
#pragma unroll
                            for (int k = 1; k < NANT; k++) s = s + tab_s[k * U + idx_of<W>(cw[j][k], i)];
                            a[i] = s;
                        } else {
                            double d = q[0] - tab_s[idx_of<W>(cw[j][0], i)];
                            double s = d * d;
#pragma unroll
                            for (int k = 1; k < NANT; k++) {
                                d = q[k] - tab_s[k * U + idx_of<W>(cw[j][k], i)];
                                const double t = d * d;
                                s = s + t;
                            }
                            a[i] = s;
                        }
                        a[i] = __dsqrt_rn(a[i]);
                    }
#pragma unroll
                    for (int i = 0; i < W; i += 2) {            // same write set as V0: whole 16-byte pairs below r_end (rows are even-sized)
                        if (W == 2 || rr + i < r_end) {
                            __builtin_nontemporal_store(a[i], out + rr + i);
                            __builtin_nontemporal_store(a[i + 1], out + rr + i + 1);
                        }
                    }
#pragma unroll
                    for (int i = W - 1; i >= 0; i--)
                        if (a[i] == 0.0 && rr + i < R) best = min(best, (unsigned)(rr + i));
                }
            }
            if (!PF) {
                const int rn = r + STEP * UNR;
#pragma unroll
                for (int j = 0; j < UNR; j++) {
                    const int rr = rn + j * STEP;
                    if (rr < r_end) {
#pragma unroll
                        for (int k = 0; k < NANT; k++) w[j][k] = ld_idx<W>(base + (size_t)k * maxR + rr);
                    }
                }
            }
        }
        best = wave_min_u32(best);
        if (best != NO_HIT && (threadIdx.x & 63) == 0) atomicMin(&hit[e], best);     // rare: only waves that saw a hit
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------
struct Cfg { const char *name; int nant, U, R, E; };
static const Cfg CFGS[] = {{"cfg2", 3, 41, 8192, 8192}, {"cfg3", 5, 1001, 32768, 32768}, {"cfg4", 5, 41, 65536, 8192}, {"cfg5", 16, 1001, 262144, 64}};

static uint64_t sm64(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

__global__ void fill_idx(uint16_t *p, size_t n, int U, uint64_t seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = (uint16_t)(z % (uint64_t)U);
    }
}
__global__ void cmp_kernel(const uint64_t *a, const uint64_t *b, size_t n, unsigned long long *bad)
{
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}

struct Ctx {
    Cfg c; int maxR;
    double *u, *ve, *x, *d_ref, *d_out;
    uint16_t *uidx; int32_t *nrules; uint32_t *hit_ref, *hit;
    unsigned long long *bad;
    size_t nd;
};

template <typename F>
static void run_variant(Ctx &C, const char *name, size_t bytes_moved, F launch)
{
    CK(hipMemset(C.d_out, 0xFF, C.nd * 8));
    CK(hipMemset(C.hit, 0xFF, sizeof(uint32_t) * C.c.E));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemset(C.bad, 0, 8));
    cmp_kernel<<<1024, 256>>>((const uint64_t *)C.d_ref, (const uint64_t *)C.d_out, C.nd, C.bad);
    unsigned long long bad = 0;
    CK(hipMemcpy(&bad, C.bad, 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> h1(C.c.E), h0(C.c.E);
    CK(hipMemcpy(h1.data(), C.hit, 4 * C.c.E, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h0.data(), C.hit_ref, 4 * C.c.E, hipMemcpyDeviceToHost));
    size_t hbad = 0;
    for (int e = 0; e < C.c.E; e++) hbad += h0[e] != h1[e];
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int rep = 0; rep < 5; rep++) {
        const int N = 10;
        CK(hipEventRecord(e0));
        for (int i = 0; i < N; i++) { CK(hipMemsetAsync(C.hit, 0xFF, sizeof(uint32_t) * C.c.E)); launch(); }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms / N);
    }
    std::sort(ts.begin(), ts.end());
    const double med = ts[ts.size() / 2];
    printf("%-44s %8.4f ms (min %8.4f)  %7.1f GB/s  frac %.3f  %s\n", name, med, ts[0], bytes_moved / med / 1e6, bytes_moved / med / 1e6 / 8000.0,
           (bad || hbad) ? "MISMATCH" : "ok");
    if (bad || hbad) printf("    !! %llu distance words and %zu hit words differ from V0\n", bad, hbad);
    fflush(stdout);
}

template <int NANT, int BLOCK, int W, int UNR, bool SQ, bool PF>
static void v1(Ctx &C, int wg_per_cu, int chunk_target, size_t bytes)
{
    const int tab = 8 * NANT * C.c.U;
    auto k = v1_kernel<NANT, BLOCK, W, UNR, SQ, PF>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, tab));
    const int gran = BLOCK * W * UNR;
    int chunk = chunk_target <= 0 ? C.maxR : chunk_target;
    chunk = ((chunk + gran - 1) / gran) * gran;
    const int cpe = (C.maxR + chunk - 1) / chunk;
    const int nitems = cpe * C.c.E;
    int grid = 256 * wg_per_cu;
    if (grid > nitems) grid = nitems;
    char name[128];
    snprintf(name, sizeof name, "V1 B%d W%d U%d %s %s wg/cu %d chunk %d", BLOCK, W, UNR, SQ ? "SQ" : "VE", PF ? "PF" : "--", wg_per_cu, chunk);
    run_variant(C, name, bytes, [&] { hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), tab, 0, C.u, C.ve, C.c.U, C.uidx, C.nrules, C.maxR, C.x, C.d_out, C.hit, chunk, cpe, nitems); });
}

template <int NANT>
static void run_cfg(Ctx &C)
{
    const size_t bytes = (size_t)C.c.E * C.c.R * (2 * NANT + 8);
    const int tab = 8 * NANT * C.c.U;
    // reference = V0 as shipped in round 1
    {
        CK(hipMemset(C.hit_ref, 0xFF, 4 * C.c.E));
        CK(hipMemset(C.d_ref, 0xFF, C.nd * 8));
        if (tab <= 48 * 1024) {
            const int rpb = 2048;
            dim3 g(C.c.E, (C.maxR + rpb - 1) / rpb);
            hipLaunchKernelGGL((v0_kernel<NANT, (NANT <= 8 ? 4 : 2), 256>), g, dim3(256), tab, 0, C.u, C.ve, C.c.U, C.uidx, C.nrules, C.maxR, C.x, C.d_ref, C.hit_ref, rpb);
        } else {
            auto k = v0_kernel<NANT, 2, 1024>;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, tab));
            const int rpb = 32768;
            dim3 g(C.c.E, (C.maxR + rpb - 1) / rpb);
            hipLaunchKernelGGL(k, g, dim3(1024), tab, 0, C.u, C.ve, C.c.U, C.uidx, C.nrules, C.maxR, C.x, C.d_ref, C.hit_ref, rpb);
        }
        CK(hipDeviceSynchronize());
    }
    if (tab <= 48 * 1024) {
        const int rpb = 2048;
        dim3 g(C.c.E, (C.maxR + rpb - 1) / rpb);
        run_variant(C, "V0 round-1 (B256 U4 chunk 2048)", bytes, [&] { hipLaunchKernelGGL((v0_kernel<NANT, (NANT <= 8 ? 4 : 2), 256>), g, dim3(256), tab, 0, C.u, C.ve, C.c.U, C.uidx, C.nrules, C.maxR, C.x, C.d_out, C.hit, rpb); });
    } else {
        auto k = v0_kernel<NANT, 2, 1024>;
        const int rpb = 32768;
        dim3 g(C.c.E, (C.maxR + rpb - 1) / rpb);
        run_variant(C, "V0 round-1 (B1024 U2 chunk 32768)", bytes, [&] { hipLaunchKernelGGL(k, g, dim3(1024), tab, 0, C.u, C.ve, C.c.U, C.uidx, C.nrules, C.maxR, C.x, C.d_out, C.hit, rpb); });
    }
    const int big = tab > 64 * 1024;
    if (!big) {
        const int wmax = tab > 16 * 1024 ? 3 : 8;
        for (int wg : {wmax, wmax > 4 ? 4 : 2}) {
            v1<NANT, 256, 2, 4, false, false>(C, wg, 0, bytes);
            v1<NANT, 256, 2, 4, true, false>(C, wg, 0, bytes);
            v1<NANT, 256, 2, 4, true, true>(C, wg, 0, bytes);
            v1<NANT, 256, 2, 2, true, true>(C, wg, 0, bytes);
            v1<NANT, 256, 4, 2, true, false>(C, wg, 0, bytes);
            v1<NANT, 256, 4, 2, true, true>(C, wg, 0, bytes);
            v1<NANT, 256, 4, 1, true, true>(C, wg, 0, bytes);
            v1<NANT, 256, 8, 1, true, true>(C, wg, 0, bytes);
        }
        v1<NANT, 512, 2, 4, true, true>(C, tab > 16 * 1024 ? 2 : 4, 0, bytes);
        v1<NANT, 512, 4, 2, true, true>(C, tab > 16 * 1024 ? 2 : 4, 0, bytes);
        v1<NANT, 1024, 4, 2, true, true>(C, tab > 16 * 1024 ? 1 : 2, 0, bytes);
        v1<NANT, 256, 2, 4, true, true>(C, wmax, 8192, bytes);       // several chunks per environment
    } else {
        for (int chunk : {32768, 65536, 0}) {
            v1<NANT, 1024, 2, 2, false, false>(C, 1, chunk, bytes);
            v1<NANT, 1024, 2, 2, true, false>(C, 1, chunk, bytes);
            v1<NANT, 1024, 2, 2, true, true>(C, 1, chunk, bytes);
            v1<NANT, 1024, 4, 1, true, true>(C, 1, chunk, bytes);
            v1<NANT, 512, 4, 1, true, true>(C, 1, chunk, bytes);
        }
    }
}

int main(int argc, char **argv)
{
    const char *want = argc > 1 ? argv[1] : "cfg4";
    Ctx C;
    bool found = false;
    for (const Cfg &c : CFGS) if (!strcmp(c.name, want)) { C.c = c; found = true; }
    if (!found) { fprintf(stderr, "unknown config %s\n", want); return 2; }
    if (argc > 2) C.c.E = atoi(argv[2]);
    const int nant = C.c.nant, U = C.c.U, E = C.c.E, R = C.c.R;
    C.maxR = R;
    std::vector<double> u((size_t)nant * U), ve((size_t)nant * U), x((size_t)E * nant);
    uint64_t s = 42;
    for (int k = 0; k < nant; k++) {
        const double div = 2.0 * (k + 1) / (U - 1);
        double acc = 0;
        for (int i = 0; i < U; i++) {
            u[(size_t)k * U + i] = -(U - 1) * div / 2 + div * i;
            if (i) acc += div * (0.5 + (double)(sm64(s) >> 11) / 9007199254740992.0);
            ve[(size_t)k * U + i] = acc;
        }
    }
    for (size_t i = 0; i < x.size(); i++) {
        const int k = i % nant;
        x[i] = u[(size_t)k * U] + (u[(size_t)k * U + U - 2] - u[(size_t)k * U]) * ((double)(sm64(s) >> 11) / 9007199254740992.0);
    }
    C.nd = (size_t)E * C.maxR;
    CK(hipMalloc(&C.u, u.size() * 8)); CK(hipMalloc(&C.ve, ve.size() * 8)); CK(hipMalloc(&C.x, x.size() * 8));
    CK(hipMalloc(&C.d_ref, C.nd * 8)); CK(hipMalloc(&C.d_out, C.nd * 8));
    CK(hipMalloc(&C.uidx, (size_t)E * nant * C.maxR * 2)); CK(hipMalloc(&C.nrules, 4 * E)); CK(hipMalloc(&C.hit_ref, 4 * E)); CK(hipMalloc(&C.hit, 4 * E));
    CK(hipMalloc(&C.bad, 8));
    CK(hipMemcpy(C.u, u.data(), u.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(C.ve, ve.data(), ve.size() * 8, hipMemcpyHostToDevice));
    fill_idx<<<4096, 256>>>(C.uidx, (size_t)E * nant * C.maxR, U, 777);
    std::vector<int32_t> nr(E, R);
    for (int e = 0; e < E; e += 97) nr[e] = R - 1 - (e % 5000);          // some ragged / odd counts
    CK(hipMemcpy(C.nrules, nr.data(), 4 * E, hipMemcpyHostToDevice));
    // a few exact hits: query = universe point of an existing rule
    std::vector<uint16_t> row(nant);
    for (int e = 0; e < E; e += 13) {
        const int r = (int)(sm64(s) % (uint64_t)(nr[e] > 0 ? nr[e] : 1));
        for (int k = 0; k < nant; k++) {
            CK(hipMemcpy(&row[k], C.uidx + ((size_t)e * nant + k) * C.maxR + r, 2, hipMemcpyDeviceToHost));
            x[(size_t)e * nant + k] = u[(size_t)k * U + row[k]];
        }
        if (e > 13 * 40) break;
    }
    CK(hipMemcpy(C.x, x.data(), x.size() * 8, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    printf("== %s: nant %d U %d R %d E %d (moved bytes per launch %.3f GB, table %d B)\n", C.c.name, nant, U, R, E, (double)E * R * (2 * nant + 8) / 1e9, 8 * nant * U);
    switch (nant) {
        case 3: run_cfg<3>(C); break;
        case 5: run_cfg<5>(C); break;
        case 16: run_cfg<16>(C); break;
    }
    return 0;
}
