/*
 * pt_oracle.h — CPU oracle for the path-tracing hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (pathtracer_cuda_interactive_amd/, libpt_hip.so) never links, imports or calls it.
 *
 * What it is: a plain-C restatement of the reference's render -> radiance ->
 * intersect path (main.cu:30-52, radiance.cuh:21-79, scene.h:176-301,333-464,
 * shape.cuh:107-215, bbox.cuh:35-61, camera.cuh:45-50, frame.h:17-64,
 * cutil_math.h:295-425, pcg.h:16-57), on the reference's own data layout, fed
 * through the same pt_scene_desc / pt_render_params the C ABI takes.
 *
 * Pinning status (see DESIGN.md "Oracle"): the reference cannot be built in this
 * image (every first-party header pulls <cuda_runtime.h>/<curand_kernel.h>, and
 * stand-in headers are not allowed), and it ships no tests or golden vectors.
 * The oracle is pinned against the values SURVEY.md §8c recorded from a host
 * build of the reference's headers (PCG known-answer vectors; scene1 / cbox
 * image statistics and pixel bit patterns under per-pixel-sequential PCG +
 * glibc math): tests/test_oracle_pins.py.  Parity with the literal CUDA/cuRAND
 * binary is unpinned (XORWOW stream not reproducible).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/pt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { PT_ORACLE_MATH_DET = 0, PT_ORACLE_MATH_LIBM = 1 };
/* PER_SAMPLE: stream = pixel*stride + sample (the build's contract);
 * PER_PIXEL : one stream per pixel consumed across all spp, exactly as
 *             main.cu:36-47 uses its curandState (used for the SURVEY pins). */
enum { PT_ORACLE_RNG_PER_SAMPLE = 0, PT_ORACLE_RNG_PER_PIXEL = 1 };

typedef struct pt_oracle_opts {
    int32_t math_mode;
    int32_t rng_mode;
    int32_t threads;        /* <=0: all online cores */
    int32_t accumulate;     /* 1: write the un-normalised SUM of samples (render_progressive) */
} pt_oracle_opts;

typedef struct pt_oracle_counters {
    uint64_t paths, segments;
    uint64_t inner_pops, leaf_tri, leaf_sphere, valid_hits, closer_hits, closer_tri;
    uint64_t rng_draws, emit;
    uint64_t term_miss, term_rr, term_absorb, term_maxdepth;
    uint64_t max_stack, stack_overflow;
    uint64_t shadow_rays, nee_hits;     /* PT_RENDER_NEE: occlusion queries traced / light samples that arrived */
    double   seconds;       /* wall time of the pixel loop only */
    int32_t  threads_used;
    int32_t  pad;
    /* optional event trace (single-threaded use only): 'I' inner visit, 'L' leaf test, 'S' shaded hit, 'M' miss, 'E' path end */
    char*    trace;
    uint64_t trace_len, trace_cap;
} pt_oracle_counters;

/* Renders rows selected by params (same packing as pt_render) into fb (host). */
int pt_oracle_render(const pt_scene_desc* scene, const pt_render_params* params,
                     const pt_oracle_opts* opts, float* fb, pt_oracle_counters* counters);

/* Renders only the listed pixels (x,y pairs) — bounded CPU-baseline samples and spot checks. */
int pt_oracle_render_pixels(const pt_scene_desc* scene, const pt_render_params* params,
                            const pt_oracle_opts* opts, const int32_t* xy, int n,
                            float* out_rgb, pt_oracle_counters* counters);

/* Event trace of the listed pixels (x,y pairs), single-threaded: fills buf with the I/L/S/M/E events of every path in
 * order; returns the number of bytes written in *len (truncated at cap).  For schedule simulations (tools/). */
int pt_oracle_trace_pixels(const pt_scene_desc* scene, const pt_render_params* params, const int32_t* xy, int n,
                           char* buf, uint64_t cap, uint64_t* len);

/* Closest hit for explicit rays: n x {org[3],dir[3],tnear,tfar} -> {t,u,v}, prim (-1 miss). */
int pt_oracle_intersect(const pt_scene_desc* scene, const float* rays, int n, int math_mode,
                        float* out_tuv, int32_t* out_prim);

/* Work of intersect() per ray, primitive tests left out: inner pops, leaves reached, and an order-independent hash of the
 * ids of the primitives whose leaves were reached (the reference never prunes, so this set does not depend on hits). */
int pt_oracle_intersect_work(const pt_scene_desc* scene, const float* rays, int n, int math_mode, uint32_t* out_inner,
                             uint32_t* out_leaf, uint64_t* out_leaf_set);

/* TEST HOOK: mutations of the estimator (bit mask; 0 = none = the reference's algorithm).  Used only to demonstrate that the
 * statistical pin on the reference's screenshots fails for a wrong estimator (tests/test_reference_images.py).  Process-wide,
 * not thread-safe against a running render.  Returns the previous mask.
 *   DIFFUSE_NO_INV_PI  eval_brdf's diffuse value without the 1/pi (scene.h:370-375), pdf unchanged
 *   RR_NO_WEIGHT       Russian-roulette survivors keep their throughput (radiance.cuh:68-74 without the division)
 *   SCHLICK_POW4       (1 - cos)^4 instead of ^5 in schlick_fresnel (scene.h:333-336)
 *   RR_FLOOR_QUARTER   CONTROL: roulette floor 0.25 instead of 0.5 — another estimator with the SAME expectation */
enum { PT_ORACLE_MUT_DIFFUSE_NO_INV_PI = 1, PT_ORACLE_MUT_RR_NO_WEIGHT = 2, PT_ORACLE_MUT_SCHLICK_POW4 = 4,
       PT_ORACLE_MUT_RR_FLOOR_QUARTER = 8 };
int pt_oracle_set_mutation(int bits);

/* math KATs: op 0 sincos(x)->(out0=sin,out1=cos); op 1 powf(x,y)->out0; math_mode as above */
int pt_oracle_math(int op, int math_mode, const float* x, const float* y, float* out0, float* out1, int n);
/* n_draws uint32 + float outputs of init_pcg32(stream, seed); also returns state/inc after init */
int pt_oracle_pcg(uint64_t stream, uint64_t seed, int n_draws, uint32_t* out_u32, float* out_f32,
                  uint64_t* state_inc /* [2] */);

#ifdef __cplusplus
}
#endif
#endif
