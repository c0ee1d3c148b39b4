/*
 * pt_oracle_math.h — fp32 math of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * Two flavours, selected by ORACLE_LIBM before inclusion of pt_oracle_core.inc:
 *
 *   ORACLE_LIBM=1  "host semantics": what the reference's headers mean when they
 *                  are compiled as host C++ (cutil_math.h:28-54 fallbacks:
 *                  fminf/fmaxf are ternaries, rsqrtf = 1/sqrtf; torrey.cuh:70-78
 *                  max/min templates are ternaries; sin/cos/pow come from glibc;
 *                  pow(x,5) is double pow).  Used only to reproduce the numbers
 *                  SURVEY.md §8c recorded from a host build of the reference.
 *
 *   ORACLE_LIBM=0  "deterministic": every transcendental is spelled out with
 *                  + - * / sqrt only (no libm, no FMA contraction), so a gfx950
 *                  kernel can be bit-identical.  This is the checker for the HIP
 *                  path.  max/min follow IEEE maxNum/minNum (what CUDA's device
 *                  max()/fmaxf resolve to, and what v_max_f32 does).
 *
 * The deterministic functions are an independent restatement of the ones in
 * pathtracer_cuda_interactive_amd/csrc/pt_math.h; tests/test_math_parity.py checks
 * the two agree bit-for-bit.  Build with -ffp-contract=off.
 */
#ifndef PT_ORACLE_MATH_H
#define PT_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

/* torrey.cuh:33-35 */
#define O_PI     ((float)3.14159265358979323846)
#define O_TWOPI  ((float)2.0 * O_PI)

/* ---- IEEE maxNum / minNum without libm calls ---- */
static inline float o_fmax_ieee(float a, float b) { return (a > b || b != b) ? a : b; }
static inline float o_fmin_ieee(float a, float b) { return (a < b || b != b) ? a : b; }
static inline float o_max_tern(float a, float b) { return a > b ? a : b; }
static inline float o_min_tern(float a, float b) { return a < b ? a : b; }

/* ---- deterministic sin/cos (Cody–Waite reduction by pi/2 + cephes minimax polynomials) ---- */
static inline void o_det_sincosf(float x, float* sn, float* cs) {
    float fq = floorf(x * 0.636619772367581343f + 0.5f);
    int q = (int)fq;
    float r = x - fq * 1.5703125f;
    r = r - fq * 4.837512969970703125e-4f;
    r = r - fq * 7.54978995489188216e-8f;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    switch (q & 3) {
        case 0: *sn = ps;  *cs = pc;  break;
        case 1: *sn = pc;  *cs = -ps; break;
        case 2: *sn = -ps; *cs = -pc; break;
        default: *sn = -pc; *cs = ps; break;
    }
}

/* ---- deterministic powf for x >= 0 (evaluated in double with + - * / only) ---- */
static inline float o_det_powf(float xf, float yf) {
    if (yf == 0.0f) return 1.0f;
    if (!(xf > 0.0f)) return 0.0f;             /* domain of the hot path: x in [0,1], y > 0 */
    if (xf == 1.0f) return 1.0f;
    double x = (double)xf;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = s2 * (0.33333333333333331 + s2 * (0.20000000000000001 + s2 * (0.14285714285714285 +
               s2 * (0.1111111111111111 + s2 * (0.090909090909090912 + s2 * (0.076923076923076927 +
               s2 * 0.066666666666666666))))));
    double lnx = (double)e * 0.69314718055994529 + (2.0 * s + 2.0 * s * p);
    double t = (double)yf * lnx;
    if (t < -104.0) return 0.0f;
    if (t > 88.8) return INFINITY;
    double kf = floor(t * 1.4426950408889634 + 0.5);
    double r = (t - kf * 0.693147180369123816490) - kf * 1.90821492927058770002e-10;
    double er = 1.0 + r * (1.0 + r * (0.5 + r * (0.16666666666666666 + r * (0.041666666666666664 +
                r * (0.0083333333333333332 + r * (0.0013888888888888889 + r * (0.00019841269841269841 +
                r * (2.4801587301587302e-05 + r * (2.7557319223985893e-06 + r * (2.7557319223985888e-07 +
                r * 2.505210838544172e-08))))))))));
    int k = (int)kf;
    uint64_t sb = (uint64_t)(k + 1023) << 52;
    double sc;
    memcpy(&sc, &sb, 8);
    return (float)(er * sc);
}

static inline float o_det_pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }

/* ---- PCG32 (pcg.h:16-57) ---- */
typedef struct { uint64_t state, inc; } o_pcg32;

static inline uint32_t o_pcg_next(o_pcg32* rng) {
    uint64_t oldstate = rng->state;
    rng->state = oldstate * 6364136223846793005ULL + (rng->inc | 1);
    uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}
static inline o_pcg32 o_pcg_init(uint64_t stream_id, uint64_t seed) {
    o_pcg32 s;
    s.state = 0U;
    s.inc = (stream_id << 1u) | 1u;
    o_pcg_next(&s);
    s.state += seed;
    o_pcg_next(&s);
    return s;
}
static inline float o_pcg_float(o_pcg32* rng) {   /* [0,1) — pcg.h:50-57 */
    union { uint32_t u; float f; } x;
    x.u = (o_pcg_next(rng) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}

#endif
