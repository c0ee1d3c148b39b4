// pt_math.h — deterministic fp32 arithmetic shared by every kernel of the path tracer.
//
// Contract: only IEEE-754 correctly rounded + - * / sqrt, no
// FMA contraction (-ffp-contract=off), no fast-math, denormals kept.  Under that
// contract a gfx950 lane and an x86-64 core produce the same bits, which is what
// lets the image be compared bit-for-bit with the CPU oracle: a path tracer
// amplifies a 1-ulp difference into a different path.
//
// Replaces, for the hot path: cutil_math.h:295-425 (float3 ops), CUDA sinf/cosf/powf
// (scene.h:342,353-354,350,397,402), pow(x,5) (scene.h:335), curand_uniform -> pcg.h:16-57.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_HD __host__ __device__ __forceinline__

namespace ptm {

struct V3 { float x, y, z; };

PT_HD V3 mk(float x, float y, float z) { return V3{x, y, z}; }
PT_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_HD V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
PT_HD V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_HD V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// cutil_math.h:401-405 with rsqrtf spelled as 1/sqrt (IEEE on both sides)
PT_HD V3 normalize(V3 v) { float inv_len = 1.0f / __builtin_sqrtf(dot(v, v)); return v * inv_len; }

// IEEE maxNum / minNum (v_max_f32 / v_min_f32)
PT_HD float fmax2(float a, float b) { return __builtin_fmaxf(a, b); }
PT_HD float fmin2(float a, float b) { return __builtin_fminf(a, b); }
PT_HD float max_elem(V3 a) { return fmax2(fmax2(a.x, a.y), a.z); }        // radiance.cuh:14-16
PT_HD float clamp01(float f) { return fmax2(0.0f, fmin2(f, 1.0f)); }      // cutil_math.h:65-68

constexpr float kPi = float(3.14159265358979323846);                        // torrey.cuh:33
constexpr float kTwoPi = float(2.0) * kPi;                                  // torrey.cuh:35

// sin & cos of x >= 0 (used with x = 2*pi*u): Cody–Waite reduction by pi/2, cephes minimax polynomials.
PT_HD void sincos_det(float x, float& sn, float& cs) {
    float fq = __builtin_floorf(x * 0.636619772367581343f + 0.5f);
    int q = (int)fq;
    float r = x - fq * 1.5703125f;
    r = r - fq * 4.837512969970703125e-4f;
    r = r - fq * 7.54978995489188216e-8f;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    float s0 = (q & 1) ? pc : ps;
    float c0 = (q & 1) ? ps : pc;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}

PT_HD double bits_to_f64(uint64_t b) { return __builtin_bit_cast(double, b); }
PT_HD uint64_t f64_to_bits(double d) { return __builtin_bit_cast(uint64_t, d); }

// x^y for x >= 0, evaluated in fp64 with + - * / only (Phong lobe: scene.h:350,397,402).
PT_HD float pow_det(float xf, float yf) {
    if (yf == 0.0f) return 1.0f;
    if (!(xf > 0.0f)) return 0.0f;
    if (xf == 1.0f) return 1.0f;
    uint64_t bits = f64_to_bits((double)xf);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = bits_to_f64((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = s2 * (0.33333333333333331 + s2 * (0.20000000000000001 + s2 * (0.14285714285714285 +
               s2 * (0.1111111111111111 + s2 * (0.090909090909090912 + s2 * (0.076923076923076927 +
               s2 * 0.066666666666666666))))));
    double lnx = (double)e * 0.69314718055994529 + (2.0 * s + 2.0 * s * p);
    double t = (double)yf * lnx;
    if (t < -104.0) return 0.0f;
    if (t > 88.8) return __builtin_inff();
    double kf = __builtin_floor(t * 1.4426950408889634 + 0.5);
    double r = (t - kf * 0.693147180369123816490) - kf * 1.90821492927058770002e-10;
    double er = 1.0 + r * (1.0 + r * (0.5 + r * (0.16666666666666666 + r * (0.041666666666666664 +
                r * (0.0083333333333333332 + r * (0.0013888888888888889 + r * (0.00019841269841269841 +
                r * (2.4801587301587302e-05 + r * (2.7557319223985893e-06 + r * (2.7557319223985888e-07 +
                r * 2.505210838544172e-08))))))))));
    int k = (int)kf;
    double sc = bits_to_f64((uint64_t)(k + 1023) << 52);
    return (float)(er * sc);
}

PT_HD float pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }

// ---- PCG32 XSH-RR with streams (pcg.h:16-57) ----
struct Pcg { uint64_t state, inc; };

PT_HD uint32_t pcg_next(Pcg& r) {
    uint64_t old = r.state;
    r.state = old * 6364136223846793005ULL + (r.inc | 1);
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31));
}
PT_HD Pcg pcg_init(uint64_t stream, uint64_t seed) {
    Pcg s;
    s.state = 0;
    s.inc = (stream << 1u) | 1u;
    pcg_next(s);
    s.state += seed;
    pcg_next(s);
    return s;
}
PT_HD float pcg_float(Pcg& r) {   // [0,1)
    uint32_t u = (pcg_next(r) >> 9) | 0x3f800000u;
    return __builtin_bit_cast(float, u) - 1.0f;
}

}  // namespace ptm
