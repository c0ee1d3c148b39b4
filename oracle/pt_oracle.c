/*
 * pt_oracle.c — CPU oracle driver.  TEST INFRASTRUCTURE ONLY (see pt_oracle.h).
 * Build: gcc -O2 -std=gnu11 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread
 */
#define _GNU_SOURCE
#include "pt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "pt_oracle_math.h"

typedef struct { float x, y, z; } v3;
static inline v3 v3make(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline v3 v3ld(const float* p) { return v3make(p[0], p[1], p[2]); }
/* cutil_math.h:300-372 */
static inline v3 v3add(v3 a, v3 b) { return v3make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3sub(v3 a, v3 b) { return v3make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3mul(v3 a, v3 b) { return v3make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3scale(v3 a, float s) { return v3make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3neg(v3 a) { return v3make(-a.x, -a.y, -a.z); }
/* cutil_math.h:383-393 */
static inline float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3cross(v3 a, v3 b) {
    return v3make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

typedef struct { v3 org, dir; float tnear, tfar; } o_ray;                 /* ray.h */
typedef struct {                                                          /* intersection.h */
    v3 position, geometric_normal, shading_normal;
    float distance;
    float bu, bv;
    int material_id, area_light_id;
} o_isect;
typedef pt_oracle_counters o_counters;
#define O_TRACE(c, ch) do { if ((c)->trace && (c)->trace_len < (c)->trace_cap) (c)->trace[(c)->trace_len++] = (ch); } while (0)

/* TEST HOOK (tests/test_reference_images.py): deliberately WRONG variants of the estimator, to show that the pin on the
 * reference's screenshots can fail.  0 = the reference's algorithm; never set outside that test. */
static int g_oracle_mutation = 0;
int pt_oracle_set_mutation(int bits) { int old = g_oracle_mutation; g_oracle_mutation = bits; return old; }

#define ORACLE_LIBM 0
#define NS det
#include "pt_oracle_core.inc"
#undef ORACLE_LIBM
#undef NS

#define ORACLE_LIBM 1
#define NS libm
#include "pt_oracle_core.inc"
#undef ORACLE_LIBM
#undef NS

/* ---------------------------------------------------------------- driver */

static void counters_add(o_counters* a, const o_counters* b) {
    a->paths += b->paths; a->segments += b->segments;
    a->inner_pops += b->inner_pops; a->leaf_tri += b->leaf_tri; a->leaf_sphere += b->leaf_sphere;
    a->valid_hits += b->valid_hits; a->closer_hits += b->closer_hits; a->closer_tri += b->closer_tri;
    a->rng_draws += b->rng_draws; a->emit += b->emit;
    a->term_miss += b->term_miss; a->term_rr += b->term_rr; a->term_absorb += b->term_absorb;
    a->term_maxdepth += b->term_maxdepth;
    if (b->max_stack > a->max_stack) a->max_stack = b->max_stack;
    a->stack_overflow += b->stack_overflow;
    a->shadow_rays += b->shadow_rays; a->nee_hits += b->nee_hits;
}

static int validate(const pt_scene_desc* sc, const pt_render_params* p) {
    if (!sc || !p) return PT_ERR_INVALID_ARG;
    if (p->width <= 0 || p->height <= 0 || p->spp <= 0) return PT_ERR_INVALID_ARG;
    if (sc->num_nodes <= 0 || sc->root < 0 || sc->root >= sc->num_nodes) return PT_ERR_BAD_SCENE;
    return PT_OK;
}

typedef struct {
    const pt_scene_desc* sc;
    const pt_render_params* p;
    pt_oracle_opts opts;
    int max_depth, rr_depth;
    /* work list: either rows or explicit pixels */
    const int* rows; int n_rows;
    const int32_t* xy; int n_xy;
    float* out;
    atomic_int next;
    o_counters* per_thread;
} job_t;

typedef struct { job_t* job; int tid; } targ_t;

static void* worker(void* a_) {
    targ_t* a = (targ_t*)a_;
    job_t* J = a->job;
    o_counters cnt;
    memset(&cnt, 0, sizeof cnt);
    const int W = J->p->width;
    const int libm = J->opts.math_mode == PT_ORACLE_MATH_LIBM;
    if (J->rows) {
        for (;;) {
            int r = atomic_fetch_add(&J->next, 1);
            if (r >= J->n_rows) break;
            int j = J->rows[r];
            float* dst = J->out + (size_t)r * W * 3;
            for (int i = 0; i < W; i++) {
                if (libm) render_pixel_libm(J->sc, J->p, J->opts.rng_mode, i, j, J->max_depth, J->rr_depth,
                                            J->opts.accumulate, dst + 3 * i, &cnt);
                else render_pixel_det(J->sc, J->p, J->opts.rng_mode, i, j, J->max_depth, J->rr_depth,
                                      J->opts.accumulate, dst + 3 * i, &cnt);
            }
        }
    } else {
        const int CH = 16;
        for (;;) {
            int b = atomic_fetch_add(&J->next, CH);
            if (b >= J->n_xy) break;
            int e = b + CH < J->n_xy ? b + CH : J->n_xy;
            for (int k = b; k < e; k++) {
                int i = J->xy[2 * k], j = J->xy[2 * k + 1];
                if (libm) render_pixel_libm(J->sc, J->p, J->opts.rng_mode, i, j, J->max_depth, J->rr_depth,
                                            J->opts.accumulate, J->out + 3 * (size_t)k, &cnt);
                else render_pixel_det(J->sc, J->p, J->opts.rng_mode, i, j, J->max_depth, J->rr_depth,
                                      J->opts.accumulate, J->out + 3 * (size_t)k, &cnt);
            }
        }
    }
    J->per_thread[a->tid] = cnt;
    return NULL;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int run_job(job_t* J, pt_oracle_counters* counters) {
    int nt = J->opts.threads;
    if (nt <= 0) {
        /* default: the CPUs this process may run on, capped (GPU boxes expose 256 hardware threads behind a
         * 16-CPU quota and a task-count guard; oversubscribing buys nothing) */
        cpu_set_t set;
        nt = (sched_getaffinity(0, sizeof set, &set) == 0) ? CPU_COUNT(&set) : (int)sysconf(_SC_NPROCESSORS_ONLN);
        if (nt > 32) nt = 32;
    }
    if (nt < 1) nt = 1;
    if (nt > 256) nt = 256;
    pthread_t* th = (pthread_t*)calloc((size_t)nt, sizeof(pthread_t));
    targ_t* ta = (targ_t*)calloc((size_t)nt, sizeof(targ_t));
    J->per_thread = (o_counters*)calloc((size_t)nt, sizeof(o_counters));
    atomic_init(&J->next, 0);
    double t0 = now_s();
    if (nt == 1) {
        ta[0].job = J; ta[0].tid = 0;
        worker(&ta[0]);
    } else {
        int started = 0;
        for (int t = 0; t < nt; t++) {
            ta[t].job = J; ta[t].tid = t;
            if (pthread_create(&th[t], NULL, worker, &ta[t]) != 0) break;   /* task limit hit: go on with fewer threads */
            started++;
        }
        if (started < nt) {                    /* the work list is shared: the calling thread drains what is left */
            ta[started].job = J; ta[started].tid = started;
            worker(&ta[started]);
        }
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    }
    double t1 = now_s();
    if (counters) {
        memset(counters, 0, sizeof *counters);
        for (int t = 0; t < nt; t++) counters_add(counters, &J->per_thread[t]);
        counters->seconds = t1 - t0;
        counters->threads_used = nt;
    }
    free(th); free(ta); free(J->per_thread);
    return PT_OK;
}

int pt_oracle_render(const pt_scene_desc* sc, const pt_render_params* p, const pt_oracle_opts* opts,
                     float* fb, pt_oracle_counters* counters) {
    int st = validate(sc, p);
    if (st) return st;
    if (!fb || !opts) return PT_ERR_INVALID_ARG;
    int rb = p->row_begin, re = p->row_end;
    if (rb == 0 && re == 0) re = p->height;
    int step = p->row_stride > 1 ? p->row_stride : 1;
    if (rb < 0 || re > p->height || rb > re) return PT_ERR_INVALID_ARG;
    int n_rows = 0;
    for (int j = rb; j < re; j += step) n_rows++;
    int* rows = (int*)malloc(sizeof(int) * (size_t)(n_rows > 0 ? n_rows : 1));
    int k = 0;
    for (int j = rb; j < re; j += step) rows[k++] = j;
    job_t J;
    memset(&J, 0, sizeof J);
    J.sc = sc; J.p = p; J.opts = *opts;
    J.max_depth = p->max_depth > 0 ? p->max_depth : 50;
    J.rr_depth = p->rr_depth >= 0 ? p->rr_depth : 5;
    J.rows = rows; J.n_rows = n_rows; J.out = fb;
    int rc = run_job(&J, counters);
    free(rows);
    return rc;
}

int pt_oracle_render_pixels(const pt_scene_desc* sc, const pt_render_params* p, const pt_oracle_opts* opts,
                            const int32_t* xy, int n, float* out_rgb, pt_oracle_counters* counters) {
    int st = validate(sc, p);
    if (st) return st;
    if (!xy || !out_rgb || !opts || n < 0) return PT_ERR_INVALID_ARG;
    for (int k = 0; k < n; k++)
        if (xy[2 * k] < 0 || xy[2 * k] >= p->width || xy[2 * k + 1] < 0 || xy[2 * k + 1] >= p->height)
            return PT_ERR_INVALID_ARG;
    job_t J;
    memset(&J, 0, sizeof J);
    J.sc = sc; J.p = p; J.opts = *opts;
    J.max_depth = p->max_depth > 0 ? p->max_depth : 50;
    J.rr_depth = p->rr_depth >= 0 ? p->rr_depth : 5;
    J.xy = xy; J.n_xy = n; J.out = out_rgb;
    return run_job(&J, counters);
}

int pt_oracle_intersect(const pt_scene_desc* sc, const float* rays, int n, int math_mode,
                        float* out_tuv, int32_t* out_prim) {
    if (!sc || !rays || !out_tuv || !out_prim || n < 0) return PT_ERR_INVALID_ARG;
    if (sc->num_nodes <= 0 || sc->root < 0 || sc->root >= sc->num_nodes) return PT_ERR_BAD_SCENE;
    if (math_mode == PT_ORACLE_MATH_LIBM) return intersect_rays_libm(sc, rays, n, out_tuv, out_prim);
    return intersect_rays_det(sc, rays, n, out_tuv, out_prim);
}

int pt_oracle_intersect_work(const pt_scene_desc* sc, const float* rays, int n, int math_mode, uint32_t* out_inner,
                             uint32_t* out_leaf, uint64_t* out_leaf_set) {
    if (!sc || !rays || !out_inner || !out_leaf || !out_leaf_set || n < 0) return PT_ERR_INVALID_ARG;
    if (sc->num_nodes <= 0 || sc->root < 0 || sc->root >= sc->num_nodes) return PT_ERR_BAD_SCENE;
    if (math_mode == PT_ORACLE_MATH_LIBM) return intersect_work_rays_libm(sc, rays, n, out_inner, out_leaf, out_leaf_set);
    return intersect_work_rays_det(sc, rays, n, out_inner, out_leaf, out_leaf_set);
}

int pt_oracle_math(int op, int math_mode, const float* x, const float* y, float* out0, float* out1, int n) {
    for (int i = 0; i < n; i++) {
        if (op == 0) {
            float s, c;
            if (math_mode == PT_ORACLE_MATH_LIBM) { s = sinf(x[i]); c = cosf(x[i]); }
            else o_det_sincosf(x[i], &s, &c);
            out0[i] = s; out1[i] = c;
        } else if (op == 1) {
            out0[i] = math_mode == PT_ORACLE_MATH_LIBM ? powf(x[i], y[i]) : o_det_powf(x[i], y[i]);
        } else return PT_ERR_INVALID_ARG;
    }
    return PT_OK;
}

int pt_oracle_pcg(uint64_t stream, uint64_t seed, int n_draws, uint32_t* out_u32, float* out_f32,
                  uint64_t* state_inc) {
    o_pcg32 a = o_pcg_init(stream, seed);
    if (state_inc) { state_inc[0] = a.state; state_inc[1] = a.inc; }
    o_pcg32 b = a;
    for (int i = 0; i < n_draws; i++) {
        if (out_u32) out_u32[i] = o_pcg_next(&a);
        if (out_f32) out_f32[i] = o_pcg_float(&b);
    }
    return PT_OK;
}

int pt_oracle_trace_pixels(const pt_scene_desc* sc, const pt_render_params* p, const int32_t* xy, int n,
                           char* buf, uint64_t cap, uint64_t* len) {
    int st = validate(sc, p);
    if (st) return st;
    if (!xy || !buf || !len) return PT_ERR_INVALID_ARG;
    o_counters cnt;
    memset(&cnt, 0, sizeof cnt);
    cnt.trace = buf; cnt.trace_cap = cap;
    const int max_depth = p->max_depth > 0 ? p->max_depth : 50, rr_depth = p->rr_depth >= 0 ? p->rr_depth : 5;
    for (int k = 0; k < n; k++) {
        float rgb[3];
        render_pixel_det(sc, p, PT_ORACLE_RNG_PER_SAMPLE, xy[2 * k], xy[2 * k + 1], max_depth, rr_depth, 0, rgb, &cnt);
    }
    *len = cnt.trace_len;
    return PT_OK;
}
