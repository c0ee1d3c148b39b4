// pt_kernels.h — the gfx950 kernels of the path tracer (included only by pt_api.hip).
//
//   trace_kernel_v2<RES,PRUNE,STATS,THRESH,INNER,MINW,SPEC>   persistent wavefront path tracer with decoupled
//        traversal / shading scheduling (default).  Replaces render + setup_rand (main.cu:30-62) and all they call.
//   trace_kernel<LDS_SCENE,PRUNE,STATS>   the simpler segment-synchronous schedule (option "kernel" = 1).
//   resolve_kernel     ordered per-pixel sum of the per-sample radiances (main.cu:47,50 / 72-86).
//   intersect_kernel, math_kernel   test hooks behind pt_debug_*.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "pt_layout.h"
#include "pt_math.h"
#include "pt_trace.h"

namespace ptk {

using namespace ptl;


#ifndef PT_BLOCK
#define PT_BLOCK 256
#endif
constexpr int kBlock = PT_BLOCK;     // 4 waves (PT_BLOCK: block-size experiments)
constexpr uint32_t kMaxChunk = 256;  // most work items a wave reserves per atomic (RenderDev::chunk)

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}


// ---- work distribution ---------------------------------------------------------------------------------------
// The frame is cut into rp.num_regions bands of rows.  A wave first serves the band that belongs to the XCD it runs
// on (HW_REG_XCC_ID), so the rays an XCD's L2 sees start in 1/8 of the image; when that band is exhausted it steals
// from the next ones.  Each band has its own work counter (128 B apart); a wave reserves rp.chunk items per atomic
// (256 for full frames; down to 64 for small launches so that every resident wave gets work).
// Placement only affects speed: every work item is rendered exactly once whatever XCD picks it up.
constexpr int kCounterStride = 32;   // uint32 slots between region counters (one 128-B line each)

struct WorkFeed {
    uint32_t cur, end;       // reserved chunk [cur, end) of region-local item ids
    uint32_t region;
    uint32_t tried;          // regions found exhausted so far
    bool exhausted;
};

__device__ __forceinline__ uint32_t region_rows(const RenderDev& rp, uint32_t region) {
    const uint32_t first = region * (uint32_t)rp.rows_per_region;
    const uint32_t nrows = (uint32_t)rp.num_rows;
    return first >= nrows ? 0u : min((uint32_t)rp.rows_per_region, nrows - first);
}

__device__ __forceinline__ void feed_init(WorkFeed& f, const RenderDev& rp) {
    f.cur = 0; f.end = 0; f.tried = 0; f.exhausted = false;
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(6164) & 15u;     // hwreg(HW_REG_XCC_ID, 0, 4)
    f.region = xcc % (uint32_t)rp.num_regions;
}

// Wave-uniform.  Makes sure a non-empty chunk is reserved unless every region is exhausted.
// (Issuing the next reservation ahead of time, to take the atomic's round trip off the critical path, was measured and
// rejected: the pending result costs more than it hides — cbox +5.5 %, profiles/r02_tune_round36_*.log.)
__device__ __forceinline__ void feed_reserve(WorkFeed& f, const RenderDev& rp, uint32_t* work_counters, int lane) {
    while (f.cur >= f.end && !f.exhausted) {
        const uint32_t total = region_rows(rp, f.region) * (uint32_t)rp.width * (uint32_t)rp.spp_pass;
        const uint32_t chunk = rp.chunk;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&work_counters[f.region * kCounterStride], chunk);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base < total) {
            f.cur = base;
            f.end = min(base + chunk, total);
        } else if (++f.tried >= (uint32_t)rp.num_regions) {
            f.exhausted = true;
        } else {
            f.region = (f.region + 1) % (uint32_t)rp.num_regions;
        }
    }
}

struct PathStart {
    ptd::Ray ray;
    ptm::Pcg rng;
    uint32_t sample_index;   // slot in the sample-major scratch buffer
};

// main.cu:32-44 for region-local work item `item` of `region`: (sample, pixel) -> PCG stream, jitter, primary ray.
__device__ __forceinline__ PathStart start_path(const RenderDev& rp, uint32_t region, uint32_t item) {
    const uint32_t npix_r = region_rows(rp, region) * (uint32_t)rp.width;
    uint32_t s_local, rr;
    int i;
    if (rp.row_major) {
        // all samples of a row before the next row: the waves of an XCD work on a few rows of its band at a time, so the
        // primary rays (and what they hit) of one moment come from a small part of the scene
        const uint32_t row_s = fastdiv(item, rp.div_width);                    // item / width = row * spp + sample
        i = (int)(item - row_s * (uint32_t)rp.width);
        rr = fastdiv(row_s, rp.div_spp);                                       // / spp_pass
        s_local = row_s - rr * (uint32_t)rp.spp_pass;
    } else {
        s_local = fastdiv(item, (int)region == rp.short_region ? rp.div_npix_last : rp.div_npix_full);   // item / npix_r
        const uint32_t pix_r = item - s_local * npix_r;
        rr = fastdiv(pix_r, rp.div_width);                                                                // pix_r / width
        i = (int)(pix_r - rr * (uint32_t)rp.width);
    }
    const uint32_t local_row = region * (uint32_t)rp.rows_per_region + rr;
    const int j = rp.row_begin + (int)local_row * rp.row_step;
    const uint64_t pixel_index = (uint64_t)j * (uint64_t)rp.width + (uint64_t)i;
    const uint64_t stream = pixel_index * (uint64_t)rp.stream_stride + (uint64_t)(rp.sample_base + (int)s_local);
    PathStart ps;
    ps.rng = ptm::pcg_init(stream, rp.seed);
    const float ru = ptm::pcg_float(ps.rng);
    const float u = ((float)i + ru) / (float)rp.width;
    const float rv = ptm::pcg_float(ps.rng);
    const float v = ((float)j + rv) / (float)rp.height;
    ps.ray = ptd::primary_ray(rp, u, v);
    ps.sample_index = s_local * rp.npix + local_row * (uint32_t)rp.width + (uint32_t)i;
    return ps;
}

__device__ __forceinline__ void stage_to_lds(void* dst, const void* src, uint32_t bytes) {
    float4* d = reinterpret_cast<float4*>(dst);
    const float4* s = reinterpret_cast<const float4*>(src);
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}

// Node placement in LDS: kLdsNodeStride bytes between consecutive nodes of a table.  64 = densely packed (4 bank
// positions for ds_read_b128), 80 = 16 bank positions.  Set by -DPT_LDS_NODE_STRIDE for experiments.
#ifndef PT_LDS_NODE_STRIDE
#define PT_LDS_NODE_STRIDE 64
#endif
constexpr uint32_t kLdsNodeStride = PT_LDS_NODE_STRIDE;
__device__ __forceinline__ void stage_nodes_to_lds(void* dst, const DNode* src, uint32_t n) {
    const float4* s = reinterpret_cast<const float4*>(src);
    for (uint32_t i = threadIdx.x; i < n * 4; i += blockDim.x)
        *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(dst) + (i >> 2) * kLdsNodeStride + (i & 3) * 16) = s[i];
}

// The scene as this kernel instantiation sees it: staged into LDS by the whole workgroup (small scenes) or read in
// place from global memory.  Must be called by every thread of the block (it contains a __syncthreads()).
// RES = residency of the scene: 0 global memory, 1 staged in LDS, 2 staged in LDS with 8 ray-octant node tables,
// 3 global memory with the top of the tree cached in LDS.
template <int RES>
__device__ __forceinline__ ptd::SceneView make_scene_view(const SceneDev& scn, const LdsPlan& lp, unsigned char* smem) {
    ptd::SceneView sv;
    sv.top_nodes = nullptr;
    sv.top_count = 0;
    if (RES == 1 || RES == 2) {
        const uint32_t oct_pitch = oct_table_pitch((uint32_t)scn.num_nodes, kLdsNodeStride);
        if (RES == 2) {
            for (uint32_t o = 0; o < 8; o++)
                stage_nodes_to_lds(smem + lp.nodes_off + o * oct_pitch, scn.nodes_oct + (size_t)o * scn.num_nodes, (uint32_t)scn.num_nodes);
        } else {
            stage_nodes_to_lds(smem + lp.nodes_off, scn.nodes, (uint32_t)scn.num_nodes);
        }
        stage_to_lds(smem + lp.prims_off, scn.prims, (uint32_t)scn.num_prims * sizeof(DPrim));
        stage_to_lds(smem + lp.normals_off, scn.normals, (uint32_t)scn.num_prims * sizeof(DNormals));
        stage_to_lds(smem + lp.mats_off, scn.materials, (uint32_t)scn.num_materials * sizeof(DMaterial));
        stage_to_lds(smem + lp.emis_off, scn.emission, (uint32_t)scn.num_emission * sizeof(DEmission));
        __syncthreads();
        sv.nodes = reinterpret_cast<const DNode*>(smem + lp.nodes_off);
        sv.prims = reinterpret_cast<const DPrim*>(smem + lp.prims_off);
        sv.normals = reinterpret_cast<const DNormals*>(smem + lp.normals_off);
        sv.materials = reinterpret_cast<const DMaterial*>(smem + lp.mats_off);
        sv.emission = reinterpret_cast<const DEmission*>(smem + lp.emis_off);
        sv.node_stride = kLdsNodeStride;
        sv.oct_stride = RES == 2 ? oct_pitch : 0u;
    } else {
        if (RES == 3) {              // top of the tree (nodes [0, top_count), breadth-first) staged next to the stacks
            stage_to_lds(smem + lp.nodes_off, scn.nodes, lp.top_count * (uint32_t)sizeof(DNode));
            __syncthreads();
            sv.top_nodes = reinterpret_cast<const DNode*>(smem + lp.nodes_off);
            sv.top_count = lp.top_count;
        }
        sv.nodes = scn.nodes; sv.prims = scn.prims; sv.normals = scn.normals;
        sv.materials = scn.materials; sv.emission = scn.emission;
        sv.node_stride = sizeof(DNode);
        sv.oct_stride = 0;
    }
    sv.lights = scn.lights;
    sv.num_emission = scn.num_emission;
    sv.root_ref = scn.root_ref;
    sv.fixed_order = scn.fixed_order;
    sv.ref_nodes = scn.ref_nodes; sv.ref_path = scn.ref_path; sv.ref_anc = scn.ref_anc; sv.ref_levels = scn.ref_levels;
    sv.bg = ptm::mk(scn.bg[0], scn.bg[1], scn.bg[2]);
    return sv;
}

// pt_counters: per-wave sums, one atomic per wave and counter.  Same-address atomics serialise in L2 (~5 ns each:
// 12 k of them at the end of a 6144-wave launch were ~60 us of every frame), so the sums are spread over kCounterSlots
// slots of one 128-B line each, picked by workgroup id; the host adds the slots up (pt_get_counters).
// Slot layout (uint64): [0] paths [1] segments [2] node visits [3] leaf tests [4..11] schedule diagnostics
// [12] segments traced on the caller's tree in reference order (exact traversal on the internal tree: zero direction component).
// After the slots: launch timeline [kTimelineBase + 0..7] and two 128-bin histograms (STATS builds only).
constexpr int kCounterSlots = 64;
constexpr int kSlotStride = 16;
constexpr int kTimelineBase = kCounterSlots * kSlotStride;
constexpr int kNumCounters = kTimelineBase + 8 + 2 * 128;

__device__ __forceinline__ unsigned long long* counter_slot(unsigned long long* counters) {
    return counters + (blockIdx.x % kCounterSlots) * kSlotStride;
}

template <bool STATS>
__device__ __forceinline__ void flush_counters(unsigned long long* counters, int lane, uint32_t n_paths, uint32_t n_segs,
                                               const ptd::TravStats& st) {
    const unsigned long long a = wave_sum(n_paths), b = wave_sum(n_segs);
    const unsigned long long c = STATS ? wave_sum(st.nodes) : 0ull, d = STATS ? wave_sum(st.leaves) : 0ull;
    if (lane == 0) {
        unsigned long long* slot = counter_slot(counters);
        atomicAdd(&slot[0], a);
        atomicAdd(&slot[1], b);
        if (STATS) { atomicAdd(&slot[2], c); atomicAdd(&slot[3], d); }
    }
}

template <bool LDS_SCENE, bool PRUNE, bool STATS>
__global__ __launch_bounds__(kBlock) void trace_kernel(SceneDev scn, RenderDev rp, LdsPlan lp,
                                                       float4* __restrict__ samples,
                                                       uint32_t* __restrict__ work_counter,
                                                       unsigned long long* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ptd::SceneView sv = make_scene_view<(LDS_SCENE ? 1 : 0)>(scn, lp, smem);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int32_t* stk = reinterpret_cast<int32_t*>(smem + lp.stack_off) + (size_t)wave * scn.stack_cap * 64 + lane;

    WorkFeed feed;
    feed_init(feed, rp);
    bool alive = false;
    ptd::Ray ray;
    ptm::V3 L = ptm::mk(0, 0, 0), T = ptm::mk(1, 1, 1);
    ptm::Pcg rng;
    rng.state = 0; rng.inc = 1;
    int depth = 0;
    uint32_t my_w = 0;
    uint32_t n_paths = 0, n_segs = 0;
    ptd::TravStats st;
    st.nodes = 0; st.leaves = 0;
    ray.org = ptm::mk(0, 0, 0); ray.dir = ptm::mk(0, 0, 1); ray.tnear = 0; ray.tfar = 0;

    for (;;) {
        // ---- refill dead lanes with new paths
        const unsigned long long need = __ballot(!alive);
        if (need) {
            feed_reserve(feed, rp, work_counter, lane);
            const uint32_t avail = feed.end - feed.cur;
            if (avail) {
                const uint32_t rank = lane_rank(need);
                const uint32_t n = (uint32_t)__popcll(need);
                if (!alive && rank < avail) {
                    const PathStart ps = start_path(rp, feed.region, feed.cur + rank);
                    ray = ps.ray;
                    rng = ps.rng;
                    my_w = ps.sample_index;
                    L = ptm::mk(0, 0, 0);
                    T = ptm::mk(1, 1, 1);
                    depth = 0;
                    alive = true;
                    n_paths++;
                }
                feed.cur += min(n, avail);
            }
        }
        if (!__any(alive)) {
            if (feed.exhausted) break;
            continue;
        }
        // ---- one path segment per live lane (radiance.cuh:24-75)
        if (alive) {
            n_segs++;
            const ptd::Hit h = ptd::intersect<PRUNE, STATS>(sv, ray, stk, st);
            bool cont = false;
            if (h.prim < 0) {
                L = L + T * sv.bg;                          // radiance.cuh:27-30
            } else {
                const ptd::Surface sf = ptd::make_surface<false>(sv, ray, h);
                cont = ptd::shade_and_bounce<false>(sv, sf, ray, rng, L, T, depth, rp.rr_depth);
                depth++;
                if (depth >= rp.max_depth) cont = false;
            }
            if (!cont) {
                samples[my_w] = make_float4(L.x, L.y, L.z, 0.0f);
                alive = false;
            }
        }
    }
    // ---- work counters (pt_counters)
    flush_counters<STATS>(counters, lane, n_paths, n_segs, st);
}

// trace_kernel_v2 — same work, different schedule: traversal is decoupled from shading.
// Every lane runs a small state machine {traversing | waiting for the scheduler phase}.  The wave keeps
// executing traversal steps (a burst of inner-node visits and leaf tests per iteration, shaped by INNER: see the
// burst code below) for the lanes that still traverse; lanes whose traversal finished wait until at least THRESH lanes can make progress in the
// scheduler phase, which then (1) shades the finished segments, (2) refills dead lanes with new paths and
// (3) starts the next traversal — so the traversal loop runs with mostly full waves instead of draining to
// the slowest ray of every segment.  Per-lane arithmetic is untouched: results stay bit-identical.
// SPEC = specialisation on scene content: 0 generic, 1 no spheres, 2 no spheres and only diffuse materials.
// RES = where the scene lives (make_scene_view): 0 global memory, 1 LDS, 2 LDS with octant node tables, 3 global memory
// with the top of the tree cached in LDS.  MINW = minimum waves per SIMD the register allocation must allow.
// NEE = built with next-event estimation (PT_RENDER_NEE): a lane alternates between closest-hit traversals and the
// any-hit traversal of its light sample's shadow ray; the same step functions serve both.
template <int RES, bool PRUNE, bool STATS, int THRESH, int INNER, int MINW, int SPEC, bool NEE = false>
__global__ __launch_bounds__(kBlock, (kBlock > 256 ? 1 : MINW)) void trace_kernel_v2(SceneDev scn, RenderDev rp, LdsPlan lp,
                                                          float4* __restrict__ samples,
                                                          uint32_t* __restrict__ work_counter,
                                                          unsigned long long* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // launch timeline (STATS builds only): 100 MHz constant clock at entry, after staging, when the wave first finds the
    // work feed dry, at exit — reduced over waves into counters[kTimelineBase..] (see flush below)
    const unsigned long long tl_entry = STATS ? wall_clock64() : 0ull;
    const ptd::SceneView sv = make_scene_view<RES>(scn, lp, smem);
    const unsigned long long tl_staged = STATS ? wall_clock64() : 0ull;
    unsigned long long tl_dry = 0;

    // LDS-resident scenes are small enough for 16-bit node / primitive references on the stack
    constexpr bool TRI_ONLY = SPEC >= 1, DIFFUSE_ONLY = SPEC >= 2;
    using STK = typename std::conditional<RES == 1 || RES == 2, int16_t, int32_t>::type;
    constexpr int32_t DONE = ptd::done_value<STK>();
    // INNER >= 1000 (internal tree only — exact traversal there may test leaves in ANY order): a lane that reaches a leaf sets
    // it aside (`pend`) and goes on with the next stack entry; the set-aside leaves are tested in the burst's leaf steps, when
    // many lanes have one, instead of every lane stalling on its leaf until the burst gets there.
    constexpr bool POSTPONE = INNER >= 1000;
    constexpr int BURST = INNER >= 1000 ? INNER - 1000 : INNER;
    int32_t pend = DONE;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    STK* stk = reinterpret_cast<STK*>(smem + lp.stack_off) + (size_t)wave * scn.stack_cap * 64 + lane;

    WorkFeed feed;
    feed_init(feed, rp);
    bool alive = false;
    ptd::Ray ray;
    ray.org = ptm::mk(0, 0, 0); ray.dir = ptm::mk(0, 0, 1); ray.tnear = 0; ray.tfar = 0;
    ptd::Trav tv;
    tv.inv = ptm::mk(1, 1, 1);
    tv.best.t = 0; tv.best.u = 0; tv.best.v = 0; tv.best.prim = -1;
    tv.cur = DONE; tv.sp = 1; tv.node_off = 0; tv.redo = false;
    ptd::stack_init(stk);
    ptm::V3 L = ptm::mk(0, 0, 0), T = ptm::mk(1, 1, 1);
    ptm::Pcg rng;
    rng.state = 0; rng.inc = 1;
    int depth = 0;
    uint32_t my_w = 0;
    uint32_t n_paths = 0, n_segs = 0;
    ptd::TravStats st;
    st.nodes = 0; st.leaves = 0;
    // exact traversal on the internal tree (scn.fallback): rays with a zero direction component use the caller's tree and a
    // global-memory stack column of their own (the LDS columns belong to the lanes that are still traversing)
    const bool fbk = scn.fallback != 0;                       // wave-uniform
    ptd::SceneView sv_ref = sv;
    sv_ref.nodes = scn.ref_nodes; sv_ref.root_ref = scn.ref_root_ref; sv_ref.node_stride = sizeof(DNode);
    sv_ref.oct_stride = 0; sv_ref.top_nodes = nullptr; sv_ref.top_count = 0; sv_ref.fixed_order = 0;
    int32_t* redo_stk = scn.redo_stack + ((size_t)(blockIdx.x * (kBlock / 64) + wave) * (size_t)scn.redo_cap) * 64 + lane;
    uint32_t n_redo = 0;
    // next-event estimation: the lane is tracing the shadow ray of its light sample; what it resumes with afterwards
    ptd::NeeState nee;
    nee.count_emission = true; nee.want_shadow = false;
    nee.ls.wl = ptm::mk(0, 0, 1); nee.ls.tfar = 0; nee.ls.contrib = ptm::mk(0, 0, 0);
    bool in_shadow = false, cont_after = false;
    ptm::V3 next_dir = ptm::mk(0, 0, 1);

    // schedule diagnostics (STATS builds only; wave-uniform, kept in scalar registers):
    // [4] loop iterations  [5] scheduler phases  [6] lanes served by scheduler phases
    // [7] inner steps executed  [8] lanes active in them  [9] leaf steps executed  [10] lanes active in them
    // [11] lane-slots idle-waiting (finished traversal or empty) summed over traversal steps (per counter slot)
    unsigned long long dg_iter = 0, dg_sched = 0, dg_sched_lanes = 0, dg_in = 0, dg_in_lanes = 0, dg_lf = 0, dg_lf_lanes = 0, dg_wait = 0;

    for (;;) {
        const bool idle = tv.cur == DONE && (!POSTPONE || pend == DONE);
        const unsigned long long idle_mask = __ballot(idle);
        if (STATS) dg_iter++;
        const bool work_left = !(feed.exhausted && feed.cur >= feed.end);        // wave-uniform
        if (STATS && !work_left && tl_dry == 0) tl_dry = wall_clock64();
        const int n_pend = __popcll(__ballot(idle && (alive || work_left)));
        // drain (no work left to refill with): fewer than THRESH lanes may be alive at all.  Scenes in global memory serve
        // the finished segments as soon as a quarter of the live lanes wait instead of waiting for the slowest traversal
        // (bunny: -1 % at 64 spp, -15 % for 2-spp frames); LDS-resident scenes gain nothing from it and keep the plain rule
        int thresh = THRESH;
        if ((RES == 0 || RES == 3) && !work_left) thresh = min(THRESH, max(1, (__popcll(__ballot(alive)) + 3) / 4));
        if (n_pend >= thresh || idle_mask == ~0ull) {
            if (n_pend == 0) break;          // every lane idle, no live path, no work left
            if (STATS) { dg_sched++; dg_sched_lanes += (unsigned)n_pend; }
            // (1) finish the segments whose traversal completed (radiance.cuh:26-75)
            if (fbk && idle && alive && tv.redo) {
                // the closest hit of this ray depends on the visit order: trace it again the reference's way, to completion
                ptd::TravStats st_redo;
                st_redo.nodes = 0; st_redo.leaves = 0;
                tv.best = ptd::intersect<false, false>(sv_ref, ray, redo_stk, st_redo);
                tv.redo = false;
                if (STATS) n_redo++;
            }
            if (idle && alive) {
                bool cont = false;
                if (NEE && in_shadow) {
                    // the shadow ray of the last bounce's light sample has been traced: add it if it arrived, then go on
                    // with the bounce that was prepared at the same time (or end the path there)
                    if (tv.best.prim < 0) L = L + nee.ls.contrib;
                    in_shadow = false;
                    cont = cont_after;
                    if (cont) { ray.dir = next_dir; ray.tnear = 1e-4f; ray.tfar = FLT_MAX; }
                } else if (tv.best.prim < 0) {
                    L = L + T * sv.bg;
                } else {
                    const ptd::Surface sf = ptd::make_surface<TRI_ONLY>(sv, ray, tv.best);
                    cont = ptd::shade_and_bounce<DIFFUSE_ONLY, NEE>(sv, sf, ray, rng, L, T, depth, rp.rr_depth, NEE ? &nee : nullptr);
                    depth++;
                    if (depth >= rp.max_depth) cont = false;
                    if (NEE && nee.want_shadow) {
                        in_shadow = true;
                        cont_after = cont;
                        next_dir = ray.dir;
                        ray.dir = nee.ls.wl; ray.tnear = 1e-4f; ray.tfar = nee.ls.tfar;     // ray.org = the hit point already
                        cont = true;
                    }
                }
                if (!cont) {
                    samples[my_w] = make_float4(L.x, L.y, L.z, 0.0f);
                    alive = false;
                }
            }
            // (2) refill dead lanes (main.cu:32-44)
            const unsigned long long need = __ballot(idle && !alive);
            if (need) {
                feed_reserve(feed, rp, work_counter, lane);
                const uint32_t avail = feed.end - feed.cur;
                if (avail) {
                    const uint32_t rank = lane_rank(need);
                    const uint32_t n = (uint32_t)__popcll(need);
                    if (idle && !alive && rank < avail) {
                        const PathStart ps = start_path(rp, feed.region, feed.cur + rank);
                        ray = ps.ray;
                        rng = ps.rng;
                        my_w = ps.sample_index;
                        L = ptm::mk(0, 0, 0);
                        T = ptm::mk(1, 1, 1);
                        depth = 0;
                        alive = true;
                        n_paths++;
                        if (NEE) { nee.count_emission = true; in_shadow = false; }
                    }
                    feed.cur += min(n, avail);
                }
            }
            // (3) start the next traversal (scene.h:247-256)
            if (idle && alive) {
                ptd::trav_begin(sv, ray, tv);
                if (!(NEE && in_shadow)) n_segs++;          // "segments" = intersect() calls; shadow rays are not counted
                // 1/d infinite on an axis: 0 * inf in the slab test is outside the argument that lets another tree stand in
                // for the caller's (pt_api.hip: validate_and_build) -> straight to the reference-order rerun
                if (fbk && !(__builtin_isfinite(tv.inv.x) && __builtin_isfinite(tv.inv.y) && __builtin_isfinite(tv.inv.z))) {
                    tv.redo = true;
                    tv.cur = DONE;
                }
            }
        }
        // ---- traversal burst
        if (POSTPONE) {
            constexpr int REPS = BURST >= 100 ? BURST / 100 : 1;
            constexpr int N_IN = BURST >= 100 ? (BURST / 10) % 10 : BURST;
            constexpr int N_LF = BURST >= 100 ? BURST % 10 : 1;
#pragma unroll
            for (int r = 0; r < REPS; r++) {
#pragma unroll
                for (int k = 0; k < N_IN; k++) {
                    if (STATS) {
                        const int n_in = __popcll(__ballot(tv.cur >= 0));
                        if (n_in) { dg_in++; dg_in_lanes += (unsigned)n_in; dg_wait += (unsigned)__popcll(__ballot(tv.cur == DONE && pend == DONE)); }
                    }
                    if (tv.cur >= 0) {
                        if (STATS) st.nodes++;
                        ptd::inner_step<PRUNE, RES == 2, STK, (RES == 3 ? 1 : 0)>(sv, ray.org, tv, stk);
                        if (tv.cur < 0 && tv.cur != DONE && pend == DONE) {       // a leaf: set it aside, take the next entry
                            pend = tv.cur;
                            tv.sp--;
                            tv.cur = stk[tv.sp * 64];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < N_LF; k++) {
                    if (STATS) {
                        const int n_lf = __popcll(__ballot(pend != DONE));
                        if (n_lf) { dg_lf++; dg_lf_lanes += (unsigned)n_lf; dg_wait += (unsigned)__popcll(__ballot(tv.cur == DONE && pend == DONE)); }
                    }
                    if (pend != DONE) {
                        if (STATS) st.leaves++;
                        ptd::leaf_test<TRI_ONLY>(sv, ray, tv, pend, fbk);
                        pend = DONE;
                        if (NEE && in_shadow && tv.best.prim >= 0) { tv.cur = DONE; tv.sp = 1; }      // occluded: done
                        if (tv.cur < 0 && tv.cur != DONE) {                       // the lane was blocked on a second leaf
                            pend = tv.cur;
                            tv.sp--;
                            tv.cur = stk[tv.sp * 64];
                        }
                    }
                }
            }
        } else if (INNER < 0) {
            // "vote" schedule: each step runs the step kind (inner-node visit or leaf test) that more lanes wait for
#pragma unroll
            for (int k = 0; k < -INNER; k++) {
                const bool at_inner = tv.cur >= 0;
                const bool at_leaf = tv.cur < 0 && tv.cur != DONE;
                const int n_in = __popcll(__ballot(at_inner));
                const int n_lf = __popcll(__ballot(at_leaf));
                if (STATS && (n_in | n_lf)) {
                    dg_wait += (unsigned)(64 - n_in - n_lf);
                    if (n_in >= n_lf) { dg_in++; dg_in_lanes += (unsigned)n_in; } else { dg_lf++; dg_lf_lanes += (unsigned)n_lf; }
                }
                if (n_in >= n_lf) {
                    if (at_inner) {
                        if (STATS) st.nodes++;
                        ptd::inner_step<PRUNE, RES == 2, STK, (RES == 3 ? 1 : 0)>(sv, ray.org, tv, stk);
                    }
                } else if (at_leaf) {
                    if (STATS) st.leaves++;
                    ptd::leaf_step<STK, TRI_ONLY, NEE>(sv, ray, tv, stk, NEE && in_shadow, fbk);
                }
            }
        } else {
            // fixed burst: REPS x (N_IN inner steps, N_LF leaf steps).  INNER = 1..9 means (INNER, 1) x 1; INNER >= 100
            // encodes REPS*100 + N_IN*10 + N_LF (e.g. 231 = two rounds of 3 inner + 1 leaf between scheduler checks).
            constexpr int REPS = INNER >= 100 ? INNER / 100 : 1;
            constexpr int N_IN = INNER >= 100 ? (INNER / 10) % 10 : (INNER > 0 ? INNER : 1);
            constexpr int N_LF = INNER >= 100 ? INNER % 10 : 1;
#pragma unroll
            for (int r = 0; r < REPS; r++) {
#pragma unroll
                for (int k = 0; k < N_IN; k++) {
                    if (STATS) {
                        const int n_in = __popcll(__ballot(tv.cur >= 0));
                        if (n_in) { dg_in++; dg_in_lanes += (unsigned)n_in; dg_wait += (unsigned)__popcll(__ballot(tv.cur == DONE)); }
                    }
                    if (tv.cur >= 0) {
                        if (STATS) st.nodes++;
                        ptd::inner_step<PRUNE, RES == 2, STK, (RES == 3 ? 1 : 0)>(sv, ray.org, tv, stk);
                    }
                }
#pragma unroll
                for (int k = 0; k < N_LF; k++) {
                    if (STATS) {
                        const int n_lf = __popcll(__ballot(tv.cur < 0 && tv.cur != DONE));
                        if (n_lf) { dg_lf++; dg_lf_lanes += (unsigned)n_lf; dg_wait += (unsigned)__popcll(__ballot(tv.cur == DONE)); }
                    }
                    if (tv.cur < 0 && tv.cur != DONE) {
                        if (STATS) st.leaves++;
                        ptd::leaf_step<STK, TRI_ONLY, NEE>(sv, ray, tv, stk, NEE && in_shadow, fbk);
                    }
                }
            }
        }
    }
    flush_counters<STATS>(counters, lane, n_paths, n_segs, st);
    const unsigned long long redo_sum = STATS ? wave_sum(n_redo) : 0ull;
    if (STATS && lane == 0) {
        const unsigned long long tl_exit = wall_clock64();
        unsigned long long* slot = counter_slot(counters);
        atomicAdd(&slot[4], dg_iter); atomicAdd(&slot[5], dg_sched); atomicAdd(&slot[6], dg_sched_lanes);
        atomicAdd(&slot[7], dg_in); atomicAdd(&slot[8], dg_in_lanes); atomicAdd(&slot[9], dg_lf);
        atomicAdd(&slot[10], dg_lf_lanes); atomicAdd(&slot[11], dg_wait);
        atomicAdd(&slot[12], redo_sum);
        // timeline, in 10-ns ticks: [0] ~(earliest entry)  [1] latest staged  [2] ~(earliest dry)  [3] latest dry
        // [4] latest exit  [5] sum over waves of (exit - dry)  [6] sum of (dry - staged)  [7] waves
        unsigned long long* tl = counters + kTimelineBase;
        if (tl_dry == 0) tl_dry = tl_exit;
        atomicMax(&tl[0], ~tl_entry); atomicMax(&tl[1], tl_staged); atomicMax(&tl[2], ~tl_dry);
        atomicMax(&tl[3], tl_dry); atomicMax(&tl[4], tl_exit); atomicAdd(&tl[5], tl_exit - tl_dry);
        atomicAdd(&tl[6], tl_dry - tl_staged); atomicAdd(&tl[7], 1ull);
        // histograms over waves, 100-us bins after the wave's own entry: [8..135] feed found dry, [136..263] exit
        atomicAdd(&tl[8 + min(127ull, (tl_dry - tl_entry) / 10000ull)], 1ull);
        atomicAdd(&tl[136 + min(127ull, (tl_exit - tl_entry) / 10000ull)], 1ull);
    }
}

// mode 0: fb = (prev + sum) * scale   (prev = accum if !first)        [final pass of pt_render]
// mode 1: accum = prev + sum          (prev = accum if !first else 0) [intermediate pass]
// mode 2: accum = first ? sum : accum + sum, sum started from zero    [render_progressive, main.cu:72-86]
__global__ __launch_bounds__(256) void resolve_kernel(const float4* __restrict__ samples, float* __restrict__ accum,
                                                      float* __restrict__ fb, uint32_t npix, int spp_pass,
                                                      int mode, int first, float scale) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    ptm::V3 sum = ptm::mk(0, 0, 0);
    if (mode != 2 && !first) sum = ptm::mk(accum[3 * (size_t)pix], accum[3 * (size_t)pix + 1], accum[3 * (size_t)pix + 2]);
    for (int s = 0; s < spp_pass; s++) {
        const float4 v = samples[(size_t)s * npix + pix];
        sum = sum + ptm::mk(v.x, v.y, v.z);                 // main.cu:47 color += radiance(...)
    }
    if (mode == 0) {
        sum = sum * scale;                                  // main.cu:50 color / float(spp) == color * (1/spp)
        fb[3 * (size_t)pix] = sum.x; fb[3 * (size_t)pix + 1] = sum.y; fb[3 * (size_t)pix + 2] = sum.z;
    } else if (mode == 1) {
        accum[3 * (size_t)pix] = sum.x; accum[3 * (size_t)pix + 1] = sum.y; accum[3 * (size_t)pix + 2] = sum.z;
    } else {
        if (!first) {
            sum = ptm::mk(accum[3 * (size_t)pix], accum[3 * (size_t)pix + 1], accum[3 * (size_t)pix + 2]) + sum;
        }
        accum[3 * (size_t)pix] = sum.x; accum[3 * (size_t)pix + 1] = sum.y; accum[3 * (size_t)pix + 2] = sum.z;
    }
}

template <bool PRUNE>
__global__ __launch_bounds__(kBlock) void intersect_kernel(SceneDev scn, const float* __restrict__ rays, int n,
                                                           float* __restrict__ out_tuv, int32_t* __restrict__ out_prim,
                                                           int32_t* __restrict__ reruns) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ptd::SceneView sv;
    sv.nodes = scn.nodes; sv.prims = scn.prims; sv.normals = scn.normals;
    sv.materials = scn.materials; sv.emission = scn.emission; sv.lights = scn.lights;
    sv.top_nodes = nullptr; sv.top_count = 0;
    sv.node_stride = sizeof(DNode);
    sv.num_emission = scn.num_emission; sv.root_ref = scn.root_ref; sv.fixed_order = scn.fixed_order;
    sv.ref_nodes = scn.ref_nodes; sv.ref_path = scn.ref_path; sv.ref_anc = scn.ref_anc; sv.ref_levels = scn.ref_levels;
    sv.bg = ptm::mk(0, 0, 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cap = scn.fallback && scn.redo_cap > scn.stack_cap ? scn.redo_cap : scn.stack_cap;   // one column serves both trees
    int32_t* stk = reinterpret_cast<int32_t*>(smem) + (size_t)wave * cap * 64 + lane;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    ptd::Ray r;
    r.org = ptm::mk(rays[8 * k], rays[8 * k + 1], rays[8 * k + 2]);
    r.dir = ptm::mk(rays[8 * k + 3], rays[8 * k + 4], rays[8 * k + 5]);
    r.tnear = rays[8 * k + 6];
    r.tfar = rays[8 * k + 7];
    ptd::TravStats st;
    ptd::Hit h;
    if (scn.fallback) {                                   // internal tree; the caller's for rays with a zero direction component
        ptd::SceneView sv_ref = sv;
        sv_ref.nodes = scn.ref_nodes; sv_ref.root_ref = scn.ref_root_ref; sv_ref.fixed_order = 0;
        bool rerun;
        h = ptd::intersect_any_tree<PRUNE>(sv, sv_ref, r, stk, rerun);
        if (rerun) atomicAdd(reruns, 1);
    } else {
        h = ptd::intersect<PRUNE, false>(sv, r, stk, st);
    }
    out_prim[k] = h.prim;
    out_tuv[3 * k] = h.prim < 0 ? 0.0f : h.t;
    out_tuv[3 * k + 1] = h.prim < 0 ? 0.0f : h.u;
    out_tuv[3 * k + 2] = h.prim < 0 ? 0.0f : h.v;
}

// Inner-node visits of a fixed set of probe rays through the tree `scn` points at (pt_api.hip: validate_and_build chooses
// between the caller's tree and the internal one by this count).  Ray k starts on primitive hash(k) mod N — its centroid,
// or the point of a sphere facing the direction — and leaves into a uniform direction: the shape of a segment after a bounce.
// Both trees in one launch: blockIdx.y picks the tree and its counter (two launches of 128 blocks each ran one after the other
// on half a chip).
__global__ __launch_bounds__(kBlock) void probe_kernel(SceneDev scn0, SceneDev scn1, unsigned long long* __restrict__ visits2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SceneDev& scn = blockIdx.y ? scn1 : scn0;
    unsigned long long* visits = visits2 + blockIdx.y;
    ptd::SceneView sv;
    sv.nodes = scn.nodes; sv.prims = scn.prims; sv.normals = scn.normals;
    sv.materials = scn.materials; sv.emission = scn.emission; sv.lights = scn.lights;
    sv.top_nodes = nullptr; sv.top_count = 0;
    sv.node_stride = sizeof(DNode);
    sv.num_emission = scn.num_emission; sv.root_ref = scn.root_ref; sv.fixed_order = scn.fixed_order;
    sv.ref_nodes = scn.ref_nodes; sv.ref_path = scn.ref_path; sv.ref_anc = scn.ref_anc; sv.ref_levels = scn.ref_levels;
    sv.bg = ptm::mk(0, 0, 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t* stk = reinterpret_cast<int32_t*>(smem) + (size_t)wave * scn.stack_cap * 64 + lane;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    ptm::Pcg rng = ptm::pcg_init(k, 0x9e3779b9u);
    const uint32_t prim = (uint32_t)(((unsigned long long)ptm::pcg_next(rng) * (unsigned long long)scn.num_prims) >> 32);
    const float z = 1.0f - 2.0f * ptm::pcg_float(rng);
    const float phi = 6.2831853f * ptm::pcg_float(rng);
    const float rr = __builtin_sqrtf(__builtin_fmaxf(0.0f, 1.0f - z * z));
    ptd::Ray r;
    float sn, cs;
    ptm::sincos_det(phi, sn, cs);                        // own sin / cos: the same probe rays, hence the same choice of tree, on every ROCm build
    r.dir = ptm::mk(rr * cs, rr * sn, z);
    const DPrim* pr = scn.prims + prim;
    const float4 a = ptd::ld4(pr, 0), b = ptd::ld4(pr, 1), c = ptd::ld4(pr, 2);
    if (__builtin_bit_cast(int32_t, c.y) < 0) {          // sphere: center + radius * dir
        r.org = ptm::mk(a.x + a.w * r.dir.x, a.y + a.w * r.dir.y, a.z + a.w * r.dir.z);
    } else {
        r.org = ptm::mk((a.x + a.w + b.z) * (1.0f / 3.0f), (a.y + b.x + b.w) * (1.0f / 3.0f), (a.z + b.y + c.x) * (1.0f / 3.0f));
    }
    r.tnear = 1e-4f;
    r.tfar = FLT_MAX;
    ptd::TravStats st;
    st.nodes = 0; st.leaves = 0;
    (void)ptd::intersect<false, true>(sv, r, stk, st);
    const unsigned long long sum = wave_sum((unsigned long long)st.nodes);
    if (lane == 0) atomicAdd(visits, sum);
}

__global__ void math_kernel(int op, const float* __restrict__ x, const float* __restrict__ y,
                            float* __restrict__ o0, float* __restrict__ o1, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (op == 0) {
        float s, c;
        ptm::sincos_det(x[k], s, c);
        o0[k] = s; o1[k] = c;
    } else if (op == 1) {
        o0[k] = ptm::pow_det(x[k], y[k]);
    } else {
        // PCG: stream = bits of x[k], seed = bits of y[k]; out0 = 1st float draw, out1 = 2nd
        ptm::Pcg r = ptm::pcg_init((uint64_t)__builtin_bit_cast(uint32_t, x[k]), (uint64_t)__builtin_bit_cast(uint32_t, y[k]));
        o0[k] = ptm::pcg_float(r);
        o1[k] = ptm::pcg_float(r);
    }
}


}  // namespace ptk
