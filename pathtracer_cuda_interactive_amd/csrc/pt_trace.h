// pt_trace.h — device functions of the path tracer: BVH traversal, primitive tests,
// hit-record assembly, BSDF sampling/evaluation.  One ray per lane (wave64).
//
// Reference functions restated here (every arithmetic expression keeps the reference's
// operand order so that results are bit-identical to the CPU oracle, see pt_math.h):
//   intersect                        scene.h:246-301
//   Hit (slab test)                  bbox.cuh:35-61      (1/d hoisted out of the loop: same bits)
//   intersect_triangle               shape.cuh:188-215
//   find_intersection_with_sphere    shape.cuh:135-186, solve_quadratic shape.cuh:110-133
//   find_intersection_with_triangle  scene.h:176-224     (record built once, for the closest hit)
//   sample_brdf / eval_brdf          scene.h:422-464 / 364-412
//   schlick_fresnel, samplers        scene.h:333-357
//   Frame / to_world                 frame.h:17-29,39-43,62-64
#pragma once

#include <float.h>

#include "pt_layout.h"
#include "pt_math.h"

namespace ptd {

using namespace ptm;
using namespace ptl;

struct Ray {
    V3 org, dir;
    float tnear, tfar;
};

struct Hit {
    float t, u, v;
    int32_t prim;      // -1 = miss
};

struct TravStats {
    uint32_t nodes, leaves;
};

// Pointers to the scene arrays as the kernel sees them (LDS or global — the address
// space is a compile-time property of each kernel instantiation).
struct SceneView {
    const DNode* nodes;
    uint32_t node_stride;   // bytes between consecutive nodes (64)
    uint32_t oct_stride;    // bytes between the 8 ray-octant copies of the node table; 0 = one table (see inner_step)
    const DNode* top_nodes; // TOP kernels: LDS copy of nodes [0, top_count), the top levels of the tree in breadth-first order
    uint32_t top_count;
    const DPrim* prims;
    const DNormals* normals;
    const DMaterial* materials;
    const DEmission* emission;
    const DLight* lights;   // global memory (next-event estimation)
    int32_t num_emission;
    int32_t root_ref;
    // the library's internal tree: the LEFT child first, always (see visit_node; the traversal stack is sized for that order)
    int32_t fixed_order;
    // ... and what settles a tie on t the reference's way without leaving that tree (ref_visits_first)
    const DNode* ref_nodes;           // the caller's tree, plain layout, global memory
    const unsigned long long* ref_path;   // [prim] turns from the root to the primitive's leaf, bit k = level k goes right
    const int32_t* ref_anc;           // [prim * ref_levels + k] the inner node of the caller's tree at level k of that path
    int32_t ref_levels;
    V3 bg;
};

__device__ __forceinline__ float4 ld4(const void* p, int i) { return reinterpret_cast<const float4*>(p)[i]; }

// Per-lane traversal state.  It lives in registers across scheduler phases of trace_kernel_v2
// (a lane keeps traversing while other lanes of its wave shade or start new paths).
struct Trav {
    V3 inv;            // 1/dir, hoisted out of Hit() (bbox.cuh:36) — same bits
    Hit best;          // closest hit so far (scene.h:248-249,270-273)
    int32_t cur;       // node reference being visited; kDone = traversal finished
    int32_t sp;        // entries on this lane's LDS stack column
    uint32_t node_off; // byte offset of the node table this ray reads (octant copy), 0 when there is one table
    bool redo;         // traversal of the internal tree: 1/d is infinite on an axis (0 * inf in the slab test lies outside the
                       // argument that lets that tree stand in) — this ray is traced on the caller's tree in the reference's order
};

// The traversal stack lives in LDS, one column per lane (entry k at stk[k*64]).  Its element type STK is int32_t, or
// int16_t for LDS-resident scenes (node and primitive counts far below 32767) to halve the LDS footprint.  The value
// that means "traversal finished" must be representable in STK: done_value<STK>().
template <class STK> __device__ __forceinline__ constexpr int32_t done_value() { return sizeof(STK) == 2 ? -32768 : kDone; }
// Entry 0 of the column holds the done value, so popping an empty stack ends the traversal without an emptiness test.
template <class STK> __device__ __forceinline__ void stack_init(STK* stk) { stk[0] = (STK)done_value<STK>(); }

__device__ __forceinline__ void trav_begin(const SceneView& sv, const Ray& ray, Trav& t) {
    t.inv = mk(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
    t.best.t = FLT_MAX; t.best.u = 0.0f; t.best.v = 0.0f; t.best.prim = -1;
    t.cur = sv.root_ref;           // scene.h:256: the root is pushed without a box test
    t.sp = 1;                      // entry 0 is the kDone sentinel
    t.redo = false;
    // octant = which of the swaps of bbox.cuh:40-55 apply to this ray (inv < 0 per axis)
    const uint32_t oct = (t.inv.x < 0.0f ? 1u : 0u) | (t.inv.y < 0.0f ? 2u : 0u) | (t.inv.z < 0.0f ? 4u : 0u);
    t.node_off = oct * sv.oct_stride;
}

// One inner-node visit (requires t.cur >= 0): test both child boxes, descend into the nearer one,
// push the farther one (scene.h:278-297).  PRUNE=false visits exactly the nodes the reference visits;
// PRUNE=true also skips a child whose box entry lies beyond the closest hit.
// OCT=true: the node table exists in 8 copies, one per ray octant, in which every box is stored as (near planes,
// far planes) for that octant — the swaps of bbox.cuh:40-55 are done once at scene-build time instead of with 12
// selects per visit.  Same arithmetic on the same operands, so the result is bit-identical.
// TOP (scenes read from global memory): the first sv.top_count nodes — the top levels of the tree, which every ray
// visits — are served from an LDS copy, the rest from global memory.  Same bytes either way.
//   TOP = 0  one table (global memory, or LDS for staged scenes)
//   TOP = 1  per lane: LDS copy when cur < top_count, else global memory (both sides may execute in one step)
// (Measured and rejected in round 2, all bit-exact: separate step kinds for LDS-served / global-memory / leaf visits,
// +39..+111 % time; a gathered fetch — node ids compacted through LDS, quad-coalesced LDS-DMA into per-wave tiles —
// +30..+50 %; child references loaded as a dwordx2, +1.5..+4 %: profiles/r02_tune_round34..37_*.log)
// Entry / exit distances of the two child boxes of a plain-layout node (a, b, c = its first three 16-B pieces): bbox.cuh:36-55.
__device__ __forceinline__ void slab_distances(const float4 a, const float4 b, const float4 c, const V3& o, const V3& inv,
                                               float& ltn, float& ltf, float& rtn, float& rtf) {
    const bool sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
    // left box
    float l0x = (a.x - o.x) * inv.x, l1x = (a.w - o.x) * inv.x;
    float l0y = (a.y - o.y) * inv.y, l1y = (b.x - o.y) * inv.y;
    float l0z = (a.z - o.z) * inv.z, l1z = (b.y - o.z) * inv.z;
    ltn = fmax2(fmax2(sx ? l1x : l0x, sy ? l1y : l0y), sz ? l1z : l0z);
    ltf = fmin2(fmin2(sx ? l0x : l1x, sy ? l0y : l1y), sz ? l0z : l1z);
    // right box
    float r0x = (b.z - o.x) * inv.x, r1x = (c.y - o.x) * inv.x;
    float r0y = (b.w - o.y) * inv.y, r1y = (c.z - o.y) * inv.y;
    float r0z = (c.x - o.z) * inv.z, r1z = (c.w - o.z) * inv.z;
    rtn = fmax2(fmax2(sx ? r1x : r0x, sy ? r1y : r0y), sz ? r1z : r0z);
    rtf = fmin2(fmin2(sx ? r0x : r1x, sy ? r0y : r1y), sz ? r0z : r1z);
}

// Everything of an inner visit after the node has been fetched (a, b, c, d = the node's four 16-B pieces).
template <bool PRUNE, bool OCT, class STK>
__device__ __forceinline__ void visit_node(const float4 a, const float4 b, const float4 c, const float4 d, const V3& o, Trav& t, STK* stk,
                                           const bool fixed_order) {
    const V3 inv = t.inv;
    float ltn, ltf, rtn, rtf;
    if (OCT) {
        ltn = fmax2(fmax2((a.x - o.x) * inv.x, (a.y - o.y) * inv.y), (a.z - o.z) * inv.z);
        ltf = fmin2(fmin2((a.w - o.x) * inv.x, (b.x - o.y) * inv.y), (b.y - o.z) * inv.z);
        rtn = fmax2(fmax2((b.z - o.x) * inv.x, (b.w - o.y) * inv.y), (c.x - o.z) * inv.z);
        rtf = fmin2(fmin2((c.y - o.x) * inv.x, (c.z - o.y) * inv.y), (c.w - o.z) * inv.z);
    } else {
        slab_distances(a, b, c, o, inv, ltn, ltf, rtn, rtf);
    }
    bool hl = ltf >= fmax2(0.0f, ltn);
    bool hr = rtf >= fmax2(0.0f, rtn);
    if (PRUNE) {
        // Skip a child whose box entry lies beyond the closest hit.  The 1e-5 relative slack keeps boxes whose
        // entry is within rounding of the hit: a triangle's Moeller-Trumbore t and its (possibly flat) box's slab
        // entry round differently, and without slack one cbox path in ~2e7 lost its true closest hit.
        const float lim = t.best.t + t.best.t * 1e-5f;
        hl = hl && !(ltn > lim);
        hr = hr && !(rtn > lim);
    }
    const int32_t L = __builtin_bit_cast(int32_t, d.x);
    const int32_t R = __builtin_bit_cast(int32_t, d.y);
    // scene.h:281-297: both hit -> visit the nearer box first (on a tie the right one) and keep the other on the
    // stack; one hit -> descend into it; none -> pop.  The stack bottom holds a kDone sentinel (trav_begin), so a
    // pop needs no emptiness test.  Written with selects so that only the push and the pop are masked regions.
    // fixed_order (wave-uniform; the internal tree only, where exact traversal may visit in ANY order — rays whose answer
    // depends on it are rerun on the caller's tree): left first.  The builder puts the child that needs the SHALLOWER stack
    // on the left, which bounds the stack by the tree's Strahler number, <= log2(leaves) + 1, however deep the tree is
    // (pt_api.hip: convert_tree).
    const bool left_first = fixed_order || ltn < rtn;
    const bool both = hl && hr;
    const int32_t next = (hl && (!hr || left_first)) ? L : R;
    if (both) {
        stk[t.sp * 64] = (STK)(left_first ? R : L);
        t.sp++;
    }
    if (hl || hr) {
        t.cur = next;
    } else {
        t.sp--;
        t.cur = stk[t.sp * 64];
    }
}

typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f4v lds_f4v;

// 64-B node at an LDS byte address (explicit LDS address space: left to itself the compiler folds LDS and global sides
// into one flat_load on a selected pointer, which sends every node fetch — also the global ones — down the slower flat path)
template <class P>
__device__ __forceinline__ void ld_node_lds(const P* p, float4& a, float4& b, float4& c, float4& d) {
    const lds_f4v* nd = (const lds_f4v*)p;
    const f4v va = nd[0], vb = nd[1], vc = nd[2], vd = nd[3];
    a = make_float4(va.x, va.y, va.z, va.w); b = make_float4(vb.x, vb.y, vb.z, vb.w);
    c = make_float4(vc.x, vc.y, vc.z, vc.w); d = make_float4(vd.x, vd.y, vd.z, vd.w);
}

template <bool PRUNE, bool OCT, class STK, int TOP = 0>
__device__ __forceinline__ void inner_step(const SceneView& sv, const V3& o, Trav& t, STK* stk) {
    float4 a, b, c, d;
    if (TOP == 1 && (uint32_t)t.cur < sv.top_count) {
        ld_node_lds(reinterpret_cast<const unsigned char*>(sv.top_nodes) + (uint32_t)t.cur * (uint32_t)sizeof(DNode), a, b, c, d);
    } else {
        const void* nd = reinterpret_cast<const unsigned char*>(sv.nodes) + (OCT ? t.node_off : 0u) + (uint32_t)t.cur * sv.node_stride;
        a = ld4(nd, 0);    // lmin.xyz lmax.x      (OCT: lnear.xyz lfar.x)
        b = ld4(nd, 1);    // lmax.yz  rmin.xy     (OCT: lfar.yz  rnear.xy)
        c = ld4(nd, 2);    // rmin.z   rmax.xyz    (OCT: rnear.z  rfar.xyz)
        d = ld4(nd, 3);    // left right - -
    }
    visit_node<PRUNE, OCT, STK>(a, b, c, d, o, t, stk, sv.fixed_order != 0);
}

// Two primitives, both tested by this ray, give the same t: the reference keeps the one it VISITS first (scene.h:270, strict <).
// Its visit order is decided at one place — the lowest node of the caller's tree that has both below it: there both child
// boxes are hit (every box above a tested leaf is), and the reference descends into the nearer one first, on equal entry
// distances the right one (scene.h:281-297 as restated in visit_node).  So: the level where the two root-to-leaf paths part
// (ref_path), the node there (ref_anc), its two entry distances for this ray — the same arithmetic on the same operands as a
// visit of that node.  True iff `b` comes before `a`.
__device__ __forceinline__ bool ref_visits_first(const SceneView& sv, const V3& o, const V3& inv, const int32_t a, const int32_t b) {
    const unsigned long long pa = sv.ref_path[a], pb = sv.ref_path[b];
    const int level = __builtin_ctzll(pa ^ pb);                 // distinct leaves: neither path is a prefix of the other
    const int32_t node = sv.ref_anc[(size_t)a * (size_t)sv.ref_levels + level];
    const DNode* nd = sv.ref_nodes + node;
    float ltn, ltf, rtn, rtf;
    slab_distances(ld4(nd, 0), ld4(nd, 1), ld4(nd, 2), o, inv, ltn, ltf, rtn, rtf);
    const bool left_first = ltn < rtn;
    const bool b_left = ((pb >> level) & 1ull) == 0ull;
    return b_left == left_first;
}

// One leaf visit (requires t.cur < 0 && t.cur != done_value<STK>()): primitive test, keep the hit if strictly closer, pop.
// TRI_ONLY: the scene holds no sphere, the sphere branch is compiled out.
// ANYHIT (kernels built with next-event estimation): a lane tracing a shadow ray (`shadow`) stops at the first hit.
// leaf_test: the primitive test alone, for leaf reference `ref` (= ~primitive); leaf_step: test of t.cur, then pop.
template <bool TRI_ONLY>
__device__ __forceinline__ void leaf_test(const SceneView& sv, const Ray& ray, Trav& t, const int32_t ref, const bool ties) {
    const int32_t prim = ~ref;
    const DPrim* pr = sv.prims + prim;
    const float4 a = ld4(pr, 0);
    const float4 b = ld4(pr, 1);
    const float4 c = ld4(pr, 2);
    const V3 o = ray.org;
    const int32_t info = __builtin_bit_cast(int32_t, c.y);
    if (TRI_ONLY || info >= 0) {
        // Möller–Trumbore, shape.cuh:188-215
        const V3 p0 = mk(a.x, a.y, a.z), p1 = mk(a.w, b.x, b.y), p2 = mk(b.z, b.w, c.x);
        const V3 e1 = p1 - p0;
        const V3 e2 = p2 - p0;
        const V3 s1 = cross(ray.dir, e2);
        const float divisor = dot(s1, e1);
        if (divisor != 0.0f) {
            const float inv_divisor = 1.0f / divisor;
            const V3 s = o - p0;
            const float u = dot(s, s1) * inv_divisor;
            const V3 s2 = cross(s, e1);
            const float v = dot(ray.dir, s2) * inv_divisor;
            const float tt = dot(e2, s2) * inv_divisor;
            if (tt > ray.tnear && tt < ray.tfar && u >= 0.0f && v >= 0.0f && u + v <= 1.0f) {
                bool take = tt < t.best.t;
                if (ties && tt == t.best.t && t.best.prim >= 0) take = ref_visits_first(sv, o, t.inv, t.best.prim, prim);
                if (take) { t.best.t = tt; t.best.u = u; t.best.v = v; t.best.prim = prim; }
            }
        }
    } else {
        // sphere, shape.cuh:135-186
        const V3 center = mk(a.x, a.y, a.z);
        const float radius = a.w;
        const V3 vv = o - center;
        const float A = dot(ray.dir, ray.dir);
        const float B = 2.0f * dot(ray.dir, vv);
        const float C = dot(vv, vv) - radius * radius;
        float t0 = 0.0f, t1 = 0.0f;
        bool ok = true;
        if (A == 0.0f) {
            if (B == 0.0f) ok = false;
            else { t0 = -C / B; t1 = t0; }
        } else {
            const float disc = B * B - 4.0f * A * C;
            if (disc < 0.0f) ok = false;
            else {
                const float rd = __builtin_sqrtf(disc);
                if (B >= 0.0f) { t0 = (-B - rd) / (2.0f * A); t1 = 2.0f * C / (-B - rd); }
                else { t0 = 2.0f * C / (-B + rd); t1 = (-B + rd) / (2.0f * A); }
            }
        }
        if (ok) {
            if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }
            float tt = t0;
            if (t1 >= ray.tnear && t1 < ray.tfar && tt < ray.tnear) tt = t1;
            if (tt >= ray.tnear && tt < ray.tfar) {
                bool take = tt < t.best.t;
                if (ties && tt == t.best.t && t.best.prim >= 0) take = ref_visits_first(sv, o, t.inv, t.best.prim, prim);
                if (take) { t.best.t = tt; t.best.u = 0.0f; t.best.v = 0.0f; t.best.prim = prim; }
            }
        }
    }
}

template <class STK, bool TRI_ONLY, bool ANYHIT = false>
__device__ __forceinline__ void leaf_step(const SceneView& sv, const Ray& ray, Trav& t, STK* stk, const bool shadow = false,
                                          const bool ties = false) {
    leaf_test<TRI_ONLY>(sv, ray, t, t.cur, ties);
    if (ANYHIT && shadow && t.best.prim >= 0) t.sp = 1;      // occluded: drop the rest of the stack, the pop below ends the traversal
    t.sp--;                       // sentinel at the stack bottom: popping an empty stack yields kDone
    t.cur = stk[t.sp * 64];
}

// Closest hit of one ray, run to completion (while-while loop): intersect() of scene.h:246-301.
// `stk` points at this lane's column of the wave's LDS stack (entry k at stk[k*64]).
template <bool PRUNE, bool STATS>
__device__ __forceinline__ Hit intersect(const SceneView& sv, const Ray& ray, int32_t* stk, TravStats& st) {
    Trav t;
    stack_init(stk);
    trav_begin(sv, ray, t);
    while (t.cur != kDone) {
        while (t.cur >= 0) {                    // descend through inner nodes until this lane holds a leaf
            if (STATS) st.nodes++;
            inner_step<PRUNE, false, int32_t>(sv, ray.org, t, stk);
        }
        if (t.cur != kDone) {                   // one primitive test, then pop
            if (STATS) st.leaves++;
            leaf_step<int32_t, false>(sv, ray, t, stk);
        }
    }
    return t.best;
}

// The same closest hit — ties and all — traversed on ANOTHER tree over the same leaf boxes (`sv`): ties on t are settled in
// the caller's visit order inside leaf_step (ref_visits_first), and a ray with a zero direction component is traced on the
// caller's tree (`sv_ref`) instead (pt_api.hip: validate_and_build has the argument).  The trace kernel inlines the same three steps into its
// scheduler; this run-to-completion form serves pt_debug_intersect.  `rerun` reports that the reference order was needed.
// PRUNE: the opt-in pruned traversal takes the same tree and the same tie handling (launch_render), its reruns are exact.
template <bool PRUNE = false>
__device__ __forceinline__ Hit intersect_any_tree(const SceneView& sv, const SceneView& sv_ref, const Ray& ray, int32_t* stk, bool& rerun) {
    Trav t;
    TravStats st;
    st.nodes = 0; st.leaves = 0;
    stack_init(stk);
    trav_begin(sv, ray, t);
    if (!(__builtin_isfinite(t.inv.x) && __builtin_isfinite(t.inv.y) && __builtin_isfinite(t.inv.z))) {
        t.redo = true;
        t.cur = kDone;
    }
    while (t.cur != kDone) {
        while (t.cur >= 0) inner_step<PRUNE, false, int32_t>(sv, ray.org, t, stk);
        if (t.cur != kDone) leaf_step<int32_t, false>(sv, ray, t, stk, false, true);
    }
    rerun = t.redo;
    return t.redo ? intersect<false, false>(sv_ref, ray, stk, st) : t.best;
}

// Surface record of the closest hit (scene.h:186-217 / shape.cuh:168-180).
struct Surface {
    V3 p, n;             // position, shading normal
    int32_t material, light;
};

template <bool TRI_ONLY>
__device__ __forceinline__ Surface make_surface(const SceneView& sv, const Ray& ray, const Hit& h) {
    Surface s;
    const DPrim* pr = sv.prims + h.prim;
    const float4 a = ld4(pr, 0);
    const float4 b = ld4(pr, 1);
    const float4 c = ld4(pr, 2);
    const int32_t info = __builtin_bit_cast(int32_t, c.y);
    s.material = info & 0x7fffffff;
    s.light = __builtin_bit_cast(int32_t, c.z);
    if (TRI_ONLY || info >= 0) {
        const V3 p0 = mk(a.x, a.y, a.z), p1 = mk(a.w, b.x, b.y), p2 = mk(b.z, b.w, c.x);
        const float w = 1.0f - h.u - h.v;
        s.p = p0 * w + p1 * h.u + p2 * h.v;
        const DNormals* nr = sv.normals + h.prim;
        const float4 na = ld4(nr, 0);
        const float4 nb = ld4(nr, 1);
        const float4 nc = ld4(nr, 2);
        const V3 n0 = mk(na.x, na.y, na.z), n1 = mk(na.w, nb.x, nb.y), n2 = mk(nb.z, nb.w, nc.x);
        s.n = normalize(n0 * w + n1 * h.u + n2 * h.v);
    } else {
        const V3 center = mk(a.x, a.y, a.z);
        s.p = ray.org + ray.dir * h.t;
        s.n = normalize(s.p - center);
    }
    return s;
}

// frame.h:17-29 + 62-64: local -> world about unit vector n (Frisvad basis)
__device__ __forceinline__ V3 to_world_about(V3 n, V3 local) {
    V3 fx, fy;
    if (n.z < float(-1 + 1e-6)) {
        fx = mk(0.0f, -1.0f, 0.0f);
        fy = mk(-1.0f, 0.0f, 0.0f);
    } else {
        const float a = 1.0f / (1.0f + n.z);
        const float b = -n.x * n.y * a;
        fx = mk(1.0f - n.x * n.x * a, b, -n.x);
        fy = mk(b, 1.0f - n.y * n.y * a, -n.y);
    }
    return fx * local.x + fy * local.y + n * local.z;
}

__device__ __forceinline__ V3 reflect_about(V3 wi, V3 n) {     // -wi + 2*dot(wi,n)*n
    const float k = 2.0f * dot(wi, n);
    return (-wi) + n * k;
}

__device__ __forceinline__ V3 schlick(V3 F0, float cos_theta) {   // scene.h:333-336
    const float p5 = pow5(1.0f - cos_theta);
    return F0 + (mk(1.0f, 1.0f, 1.0f) - F0) * p5;
}

__device__ __forceinline__ V3 sample_cos_hemisphere(float ux, float uy) {   // scene.h:338-345
    const float phi = kTwoPi * ux;
    const float tmp = __builtin_sqrtf(clamp01(1.0f - uy));
    float sn, cs;
    sincos_det(phi, sn, cs);
    return mk(cs * tmp, sn * tmp, __builtin_sqrtf(clamp01(uy)));
}

__device__ __forceinline__ V3 sample_cos_n_hemisphere(float ux, float uy, float exponent) {   // scene.h:348-357
    const float phi = kTwoPi * ux;
    const float cos_theta = pow_det(uy, 1.0f / (exponent + 1.0f));
    const float sin_theta = __builtin_sqrtf(clamp01(1.0f - cos_theta * cos_theta));
    float sn, cs;
    sincos_det(phi, sn, cs);
    return mk(cs * sin_theta, sn * sin_theta, cos_theta);
}

// ---- next-event estimation (an extension, see include/pt_api.h PT_RENDER_NEE) --------------------------
// Mirrors oracle/pt_oracle_core.inc nee_sample / nee_brdf expression for expression (bit parity).
struct LightSample {
    V3 wl;             // unit direction to the light
    float tfar;        // shadow ray extent
    V3 contrib;        // radiance to add if the shadow ray arrives
};

// eval_brdf (scene.h:364-412) for direction wl, divided by the probability of the lobe the estimator is in
__device__ __forceinline__ V3 nee_brdf(int32_t mtype, V3 refl, float eta, float exponent, V3 n, V3 wi, V3 wl) {
    if (mtype == 0) {
        const float c = fmax2(dot(wl, n), 0.0f) / kPi;
        return refl * c;
    }
    if (mtype == 2) {
        const float q = (eta - 1.0f) / (eta + 1.0f);
        const float F0s = q * q;
        const float p5 = pow5(1.0f - dot(n, wi));
        const V3 F0 = mk(F0s, F0s, F0s);
        const V3 F = F0 + (mk(1.0f, 1.0f, 1.0f) - F0) * p5;
        const float c = fmax2(dot(wl, n), 0.0f) / kPi;
        const V3 value = ((mk(1.0f, 1.0f, 1.0f) - F) * refl) * c;
        return value * (1.0f / (1.0f - F.x));
    }
    if (mtype == 3) {
        const V3 r = (-wi) + n * (2.0f * dot(wi, n));
        const float r_dot = dot(r, wl), n_dot = dot(n, wl);
        if (r_dot > 0.0f && n_dot > 0.0f) {
            const float resp = ((exponent + 1.0f) / (2.0f * kPi)) * pow_det(r_dot, exponent);
            return refl * resp;
        }
    }
    return mk(0.0f, 0.0f, 0.0f);
}

// One light sample at p (normal n faces wi).  Draws: 1 (which light) + 2 (where on an area light).
__device__ __forceinline__ bool nee_sample(const SceneView& sv, int32_t mtype, V3 refl, float eta, float exponent,
                                           V3 p, V3 n, V3 wi, V3 T, Pcg& rng, LightSample& out) {
    const int nl = sv.num_emission;
    if (nl <= 0) return false;
    const float xi = pcg_float(rng);
    int k = (int)(xi * (float)nl);
    if (k > nl - 1) k = nl - 1;
    const float4 l0 = ld4(sv.lights + k, 0);
    const float4 l1 = ld4(sv.lights + k, 1);
    V3 x, le;
    float geom;
    if (__builtin_bit_cast(int32_t, l0.w) == 0) {                  // PT_LIGHT_POINT
        x = mk(l1.x, l1.y, l1.z);
        le = mk(l0.x, l0.y, l0.z);
        geom = 1.0f;
        const V3 dv = x - p;
        const float d2 = dot(dv, dv);
        if (!(d2 > 0.0f)) return false;
        const float d = __builtin_sqrtf(d2);
        const float invd = 1.0f / d;
        const V3 wl = dv * invd;
        const V3 f = nee_brdf(mtype, refl, eta, exponent, n, wi, wl);
        if (!(max_elem(f) > 0.0f)) return false;
        const float w = ((float)nl * geom) / d2;
        out.contrib = T * ((f * le) * w);
        out.wl = wl;
        out.tfar = d;
        return true;
    }
    const float u1 = pcg_float(rng);
    const float u2 = pcg_float(rng);
    const int32_t prim = __builtin_bit_cast(int32_t, l1.w);
    const DPrim* pr = sv.prims + prim;
    const float4 a = ld4(pr, 0);
    const float4 b = ld4(pr, 1);
    const float4 c = ld4(pr, 2);
    // radiance a BSDF-sampled ray would pick up on this primitive (radiance.cuh:35-43 incl. the parsed-light-id quirk)
    const int32_t lid = __builtin_bit_cast(int32_t, c.z);
    if (!(lid >= 0 && lid < sv.num_emission)) return false;
    const float4 e = ld4(sv.emission + lid, 0);
    if (__builtin_bit_cast(int32_t, e.w) == 0) return false;
    le = mk(e.x, e.y, e.z);
    V3 nx, ng;
    float area;
    if (__builtin_bit_cast(int32_t, c.y) < 0) {                    // sphere
        const float z = 1.0f - 2.0f * u1;
        const float r = __builtin_sqrtf(clamp01(1.0f - z * z));
        const float phi = kTwoPi * u2;
        float sn, cs;
        sincos_det(phi, sn, cs);
        nx = mk(r * cs, r * sn, z);
        ng = nx;
        x = mk(a.x, a.y, a.z) + nx * a.w;
        area = (4.0f * kPi) * (a.w * a.w);
    } else {
        const V3 p0 = mk(a.x, a.y, a.z), p1 = mk(a.w, b.x, b.y), p2 = mk(b.z, b.w, c.x);
        const float su = __builtin_sqrtf(u1);
        const float bu = su * (1.0f - u2), bv = su * u2, bw = 1.0f - su;
        x = (p0 * bw + p1 * bu) + p2 * bv;
        const V3 cr = cross(p1 - p0, p2 - p0);
        const float len = __builtin_sqrtf(dot(cr, cr));
        if (!(len > 0.0f)) return false;
        const float invl = 1.0f / len;
        ng = cr * invl;
        area = 0.5f * len;
        const DNormals* nr = sv.normals + prim;
        const float4 na = ld4(nr, 0);
        const float4 nb = ld4(nr, 1);
        const float4 nc = ld4(nr, 2);
        const V3 n0 = mk(na.x, na.y, na.z), n1 = mk(na.w, nb.x, nb.y), n2 = mk(nb.z, nb.w, nc.x);
        nx = normalize((n0 * bw + n1 * bu) + n2 * bv);
    }
    const V3 dv = x - p;
    const float d2 = dot(dv, dv);
    if (!(d2 > 0.0f)) return false;
    const float d = __builtin_sqrtf(d2);
    const float invd = 1.0f / d;
    const V3 wl = dv * invd;
    if (!(dot(-wl, nx) > 0.0f)) return false;                      // radiance.cuh:38: emits towards dot(-ray.dir, n) > 0 only
    const float cosg = __builtin_fabsf(dot(wl, ng));
    geom = cosg * area;
    const V3 f = nee_brdf(mtype, refl, eta, exponent, n, wi, wl);
    if (!(max_elem(f) > 0.0f)) return false;
    const float w = ((float)nl * geom) / d2;
    out.contrib = T * ((f * le) * w);
    out.wl = wl;
    out.tfar = d * (1.0f - 1e-4f);
    return true;
}

// One bounce of radiance() after a hit (radiance.cuh:32-74): emission, BSDF sampling,
// throughput update, next ray, Russian roulette.  Returns false when the path ends.
// DIFFUSE_ONLY: every material of the scene is DIFFUSE; the mirror / plastic / Phong code (and the registers its fp64
// pow needs) is compiled out.
// Per-path state of next-event estimation that lives across scheduler phases (kernels built with NEE only).
struct NeeState {
    bool count_emission;   // emission found by BSDF sampling counts on camera rays and after specular bounces only
    bool want_shadow;      // out: the light sample of this bounce needs its shadow ray traced
    LightSample ls;        // out: that ray and the radiance it carries
};

template <bool DIFFUSE_ONLY, bool NEE = false>
__device__ __forceinline__ bool shade_and_bounce(const SceneView& sv, const Surface& sf, Ray& ray, Pcg& rng,
                                                 V3& L, V3& T, int depth, int rr_depth, NeeState* nee = nullptr) {
    V3 n = sf.n;
    const V3 wi = -ray.dir;
    const float wi_n = dot(wi, n);
    // radiance.cuh:35-43 — lights[] indexed by the parsed light id
    if (sf.light >= 0 && sf.light < sv.num_emission) {
        const float4 e = ld4(sv.emission + sf.light, 0);
        if (__builtin_bit_cast(int32_t, e.w) != 0 && wi_n > 0.0f && (!NEE || nee->count_emission)) L = L + T * mk(e.x, e.y, e.z);
    }
    if (wi_n < 0.0f) n = -n;

    const DMaterial* mp = sv.materials + sf.material;
    const float4 m0 = ld4(mp, 0);
    const float4 m1 = ld4(mp, 1);
    const int32_t mtype = __builtin_bit_cast(int32_t, m0.x);
    const V3 refl = mk(m0.y, m0.z, m0.w);
    bool have_light = false, specular = false;
    if (NEE) {
        // the light sample is drawn first (draw order: light, BSDF, roulette) and used only if the bounce turns out
        // non-specular (PLASTIC decides below); MIRROR has no non-specular lobe
        nee->want_shadow = false;
        if ((DIFFUSE_ONLY || mtype != 1))
            have_light = nee_sample(sv, DIFFUSE_ONLY ? 0 : mtype, refl, m1.x, m1.y, sf.p, n, wi, T, rng, nee->ls);
    }
    V3 wo = mk(0.0f, 0.0f, 0.0f);
    bool ok = true;                                      // false: the BSDF sample carries nothing, the path ends (radiance.cuh:49-63)
    if (DIFFUSE_ONLY || mtype == 0) {                   // DIFFUSE: scene.h:429-433 + 370-375
        const float ux = pcg_float(rng);
        const float uy = pcg_float(rng);
        wo = to_world_about(n, sample_cos_hemisphere(ux, uy));
        const float c = fmax2(dot(wo, n), 0.0f) / kPi;
        const V3 value = refl * c;
        if (!(max_elem(value) > 0.0f && c > 0.0f)) ok = false;
        else T = T * (value * (1.0f / c));
    } else if (mtype == 1) {                            // MIRROR: scene.h:434-438
        specular = true;
        wo = reflect_about(wi, n);
        const V3 F = schlick(refl, dot(n, wo));
        if (!(max_elem(F) > 0.0f)) ok = false;
        else T = T * F;
    } else if (mtype == 2) {                            // PLASTIC: scene.h:439-454 + 379-389
        const float q = (m1.x - 1.0f) / (m1.x + 1.0f);
        const float F0s = q * q;
        const V3 F = schlick(mk(F0s, F0s, F0s), dot(n, wi));
        const float xi = pcg_float(rng);
        if (xi <= F.x) {
            specular = true;
            wo = reflect_about(wi, n);
            // weight (1,1,1): throughput *= 1
            T = T * mk(1.0f, 1.0f, 1.0f);
        } else {
            const float ux = pcg_float(rng);
            const float uy = pcg_float(rng);
            wo = to_world_about(n, sample_cos_hemisphere(ux, uy));
            const float c = fmax2(dot(wo, n), 0.0f) / kPi;
            const V3 value = ((mk(1.0f, 1.0f, 1.0f) - F) * refl) * c;
            const float pdf = (1.0f - F.x) * c;
            if (!(max_elem(value) > 0.0f && pdf > 0.0f)) ok = false;
            else T = T * (value * (1.0f / pdf));
        }
    } else {                                            // PHONG: scene.h:455-460 + 390-408
        const float ux = pcg_float(rng);
        const float uy = pcg_float(rng);
        const V3 r = reflect_about(wi, n);
        wo = to_world_about(r, sample_cos_n_hemisphere(ux, uy, m1.y));
        const float r_dot_wo = dot(r, wo);
        const float n_dot_wo = dot(n, wo);
        if (!(r_dot_wo > 0.0f && n_dot_wo > 0.0f)) {
            ok = false;
        } else {
            const float resp = ((m1.y + 1.0f) / (2.0f * kPi)) * pow_det(r_dot_wo, m1.y);
            const V3 value = refl * resp;
            if (!(max_elem(value) > 0.0f && resp > 0.0f)) ok = false;
            else T = T * (value * (1.0f / resp));
        }
    }
    if (NEE) {
        // the light sample stands whether or not the BSDF sample carries on (the oracle adds it before the absorb test)
        nee->want_shadow = have_light && !specular;
        nee->count_emission = specular;
    }
    ray.org = sf.p;
    if (!ok) return false;
    ray.dir = wo;
    ray.tnear = 1e-4f;
    ray.tfar = FLT_MAX;
    if (depth > rr_depth) {                             // radiance.cuh:68-74
        const float q = fmax2(0.5f, 1.0f - max_elem(T));
        const float xi = pcg_float(rng);
        if (xi < q) return false;
        T = T * (1.0f / (1.0f - q));
    }
    return true;
}

// camera.cuh:45-50
__device__ __forceinline__ Ray primary_ray(const RenderDev& rp, float u, float v) {
    const V3 tl = mk(rp.cam_top_left[0], rp.cam_top_left[1], rp.cam_top_left[2]);
    const V3 hz = mk(rp.cam_horizontal[0], rp.cam_horizontal[1], rp.cam_horizontal[2]);
    const V3 vt = mk(rp.cam_vertical[0], rp.cam_vertical[1], rp.cam_vertical[2]);
    const V3 og = mk(rp.cam_origin[0], rp.cam_origin[1], rp.cam_origin[2]);
    Ray r;
    r.org = og;
    r.dir = normalize(((tl + hz * u) - vt * v) - og);
    r.tnear = 0.0f;
    r.tfar = __builtin_inff();
    return r;
}

}  // namespace ptd
