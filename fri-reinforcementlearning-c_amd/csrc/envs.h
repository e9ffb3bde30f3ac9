// envs.h -- the three demo environments as device functions, plus the FMA-free trig they use.
//
// Dynamics / reward / observation snap restate the reference's example callbacks:
//   mountaincar examples/mountaincar/mountaincar.c:37-74, 77-95, 98-125
//   cartpole    examples/cartpole/cartpole.c:35-77, 79-112, 114-168
//   acrobot     examples/acrobot/acrobot.c:31-130, 133-162, 165-192
// The reference calls glibc sin/cos, whose bits the device cannot reproduce (and which differ between
// glibc versions).  The batched environments therefore use the portable polynomial
// sin/cos below: plain IEEE mul/add only (built with -ffp-contract=off), the same source text as
// oracle orc_sin / orc_cos, so host checker and device agree bit for bit.
#pragma once

#include "device_common.h"

namespace frirl {

#define FRIRL_PI 3.14159265358979323846264338327   /* the literal every reference example defines */

__device__ __forceinline__ double k_sin(double x)   // |x| <= pi/4
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    double r = S6;
    r = r * z; r = r + S5;
    r = r * z; r = r + S4;
    r = r * z; r = r + S3;
    r = r * z; r = r + S2;
    r = r * z; r = r + S1;
    double t = z * x;
    t = t * r;
    return x + t;
}

__device__ __forceinline__ double k_cos(double x)   // |x| <= pi/4
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    double r = C6;
    r = r * z; r = r + C5;
    r = r * z; r = r + C4;
    r = r * z; r = r + C3;
    r = r * z; r = r + C2;
    r = r * z; r = r + C1;
    r = r * z;
    r = r * z;
    const double h = 0.5 * z;
    const double w = 1.0 - h;
    double e = (1.0 - w) - h;
    e = e + r;
    return w + e;
}

__device__ __forceinline__ double trig_reduce(double x, int &quad)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11, P3 = 2.02226624871116645580e-21;
    double fn = x * INV_PIO2;
    fn = (fn >= 0.0) ? floor(fn + 0.5) : -floor(0.5 - fn);
    const double a = fn * P1, b = fn * P2, c = fn * P3;
    double r = x - a;
    r = r - b;
    r = r - c;
    const long long n = (long long)fn;
    quad = (int)(n & 3);
    return r;
}

__device__ __forceinline__ double p_sin(double x)
{
    int q;
    const double r = trig_reduce(x, q);
    switch (q) {
        case 0: return k_sin(r);
        case 1: return k_cos(r);
        case 2: return -k_sin(r);
        default: return -k_cos(r);
    }
}

__device__ __forceinline__ double p_cos(double x)
{
    int q;
    const double r = trig_reduce(x, q);
    switch (q) {
        case 0: return k_cos(r);
        case 1: return -k_sin(r);
        case 2: return -k_cos(r);
        default: return k_sin(r);
    }
}

// round-half-away-from-zero like C round() (device round() is the same function; spelled out to keep
// the host checker and the device on identical arithmetic)
__device__ __forceinline__ double c_round(double x) { return round(x); }

// ---- do_action ------------------------------------------------------------------------------
__device__ __forceinline__ void env_do_action(int kind, double a, const double *s, double *ns)
{
    if (kind == FRIRL_HIP_ENV_MOUNTAINCAR) {
        const double pos = s[0], vel = s[1];
        double v1 = (vel + (0.001 * a) + (-0.0025 * p_cos(3.0 * pos))) * 0.999;
        if (v1 < -0.07) v1 = -0.07;
        if (v1 > +0.07) v1 = +0.07;
        double p1 = pos + v1;
        if (p1 <= -1.5) { p1 = -1.5; v1 = 0.0; }
        ns[0] = p1; ns[1] = v1;
    } else if (kind == FRIRL_HIP_ENV_CARTPOLE) {
        const double x = s[0], xd = s[1], th = s[2], thd = s[3];
        const double g = 9.8, mc = 1.0, mp = 0.1, mt = mc + mp, len = 0.5, pml = mp * len;
        const double fmag = 10.0, tau = 0.02, fourthirds = 4.0 / 3.0;
        const double force = a * fmag;
        const double sn = p_sin(th), cs = p_cos(th);
        const double temp = (force + pml * thd * thd * sn) / mt;
        const double thacc = (g * sn - cs * temp) / (len * (fourthirds - mp * cs * cs / mt));
        const double xacc = temp - pml * thacc * cs / mt;
        ns[0] = x + tau * xd;
        ns[1] = xd + tau * xacc;
        ns[2] = th + tau * thd;
        ns[3] = thd + tau * thacc;
    } else {
        const double vmax1 = 4 * FRIRL_PI, vmax2 = 9 * FRIRL_PI;
        const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0, I2 = 1.0, g = 9.8, dt = 0.05;
        const double l1sq = l1 * l1, lc1sq = lc1 * lc1, lc2sq = lc2 * lc2;
        double t1 = s[0], t2 = s[1], t1d = s[2], t2d = s[3];
        const double c2 = p_cos(t2), s2 = p_sin(t2);
        const double d1 = m1 * lc1sq + m2 * (l1sq + lc2sq + 2 * l1 * lc2 * c2) + I1 + I2;
        const double d2 = m2 * (lc2sq + l1 * lc2 * c2) + I2;
        const double phi2 = m2 * lc2 * g * p_cos(t1 + t2 - FRIRL_PI / 2);
        const double phi1 = -m2 * l1 * lc2 * t2d * s2 * (t2d - 2 * t1d) + (m1 * lc1 + m2 * l1) * g * p_cos(t1 - (FRIRL_PI / 2)) + phi2;
        double acc2 = (a + phi1 * (d2 / d1) - m2 * l1 * lc2 * t1d * t1d * s2 - phi2);
        acc2 = acc2 / (m2 * lc2sq + I2 - (d2 * d2 / d1));
        const double acc1 = -(d2 * acc2 + phi1) / d1;
        for (int i = 0; i < 4; i++) {
            t1d = t1d + acc1 * dt;
            if (t1d < -vmax1) t1d = -vmax1;
            if (t1d > vmax1) t1d = vmax1;
            t1 = t1 + t1d * dt;
            t2d = t2d + acc2 * dt;
            if (t2d < -vmax2) t2d = -vmax2;
            if (t2d > vmax2) t2d = vmax2;
            t2 = t2 + t2d * dt;
        }
        if (t1 < -FRIRL_PI) t1 = -FRIRL_PI;
        if (t1 > FRIRL_PI) t1 = FRIRL_PI;
        if (t2 < -FRIRL_PI) t2 = -FRIRL_PI;
        if (t2 > FRIRL_PI) t2 = FRIRL_PI;
        ns[0] = t1; ns[1] = t2; ns[2] = t1d; ns[3] = t2d;
    }
}

// ---- get_reward -----------------------------------------------------------------------------
__device__ __forceinline__ void env_get_reward(int kind, const double *s, double &r, int &f)
{
    if (kind == FRIRL_HIP_ENV_MOUNTAINCAR) {
        r = -10; f = 0;
        if (s[0] >= 0.45) { r = 1000; f = 1; }
    } else if (kind == FRIRL_HIP_ENV_CARTPOLE) {
        const double x = s[0], th = s[2], thd = s[3];
        const double deg45 = FRIRL_PI / 4;
        if ((x < -4.0) || (x > 4.0) || (th < (-1 * deg45)) || (th > deg45)) { r = -10000 - 50 * fabs(x) - 100 * fabs(th); f = 1; }
        else { r = 10 - 1000 * th * th - 5 * fabs(x) - 10 * thd; f = 0; }
    } else {
        const double y1 = 0.0 - p_cos(s[0]);
        const double y2 = y1 - p_cos(s[1]);
        const double goal = 0.0 + 1.0;
        r = -10; f = 0;
        if (y2 >= goal) { r = 1000; f = 1; }
    }
}

// ---- quantize_observations ------------------------------------------------------------------
// grid: [nant][FRIRL_HIP_MAX_GRID] possible values per dim; generic rule = round((s+|v0|)/div) clamped.
__device__ __forceinline__ void env_quantize(int kind, int ns_len, const double *__restrict__ grid, const int32_t *grid_len,
                                             const double *grid_div, const double *s, double *q)
{
    if (kind == FRIRL_HIP_ENV_CARTPOLE) {
        const double deg12 = FRIRL_PI / 15, deg3 = FRIRL_PI / 60;
        double q0 = s[0], q1 = c_round(s[1]), q2 = floor(s[2] / deg3) * deg3, q3 = s[3];
        if (q0 < 0) q0 = -1;
        if (q0 > 0) q0 = 1;
        if (q1 < -1) q1 = -1;
        if (q1 > 1) q1 = 1;
        if (q2 > deg12) q2 = deg12;
        if (q2 < (-1 * deg12)) q2 = -1 * deg12;
        if (q3 < 0) q3 = -1;
        if (q3 > 0) q3 = 1;
        q[0] = q0; q[1] = q1; q[2] = q2; q[3] = q3;
    } else {
        for (int i = 0; i < ns_len; i++) {
            const double *v = grid + (size_t)i * FRIRL_HIP_MAX_GRID;
            int where = (int)c_round((s[i] + fabs(v[0])) / grid_div[i]);
            if (where < 0) where = 0;
            else if (where > grid_len[i] - 1) where = grid_len[i] - 1;
            q[i] = v[where];
        }
    }
}

// frirl_check_possible_states (reference src/frirl/frirl_check_possible_states.c:96-122 with
// hit_between_possible_places :53-88): snap an antecedent of a would-be new rule to the allowed grid;
// ties go to the upper neighbour.  The grid-refinement branch (:68-75) is unreachable on uniform grids.
__device__ __forceinline__ double check_possible_states(double obs, const double *__restrict__ v, int n)
{
    int i = 0;
    bool found = false;
    for (; i < n; i++) if (obs < v[i]) { found = true; break; }
    if (!found) return v[n - 1];
    if (obs < v[0]) return v[0];
    i--;
    const double rel = obs - v[i], rel_next = v[i + 1] - obs;
    return (rel < rel_next) ? v[i] : v[i + 1];
}

// Counter-based per-environment random stream: SplitMix64 finaliser over (seed, env, episode, step, draw).
// Stateless, so the trajectory of environment e does not depend on how environments are sharded over GPUs.
__device__ __forceinline__ double rng_unit(uint64_t seed, uint64_t env, uint32_t episode, uint32_t step, uint32_t draw)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ULL * ((env << 32) | episode) + 0xD1B54A32D192ED03ULL * (((uint64_t)step << 8) | draw);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// frirl_e_greedy_selection (reference src/frirl/frirl_e_greedy_selection.c:21-37): greedy when no_random == 1 or
// epsilon == 0 or the draw exceeds epsilon; otherwise a uniformly drawn action (the reference's index can equal A,
// one past the last action; clamped here -- SURVEY Appendix C "fix in the batched path").
__device__ __forceinline__ int e_greedy(const frirl_hip_agent &ag, int greedy, uint32_t env, uint32_t episode, uint32_t step)
{
    if (ag.no_random == 1 || ag.epsilon == 0.0) return greedy;
    const uint64_t gid = ag.env_id_base + env;
    if (rng_unit(ag.seed, gid, episode, step, 0) > ag.epsilon) return greedy;
    int a = (int)round(rng_unit(ag.seed, gid, episode, step, 1) * ag.A);
    return a >= ag.A ? ag.A - 1 : a;
}

}  // namespace frirl
