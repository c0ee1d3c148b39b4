// pt_bvh_build.hip — BVH construction on the GPU, part of libpt_hip.so.
//
// The reference builds its tree on the host: construct_bvh (bvh.cu:16-54) sorts the primitives of every node by
// centroid on the node's largest axis and cuts at the OBJECT median — O(N log^2 N), 10-57 s of start-up on the
// README's machines (README.md:123,132), and a poor tree: it cuts through the middle of the object list wherever that
// falls in space, and exact traversal (scene.h:258-297 never prunes) pays for every box a ray touches.
//
// Here the primitives are put in Morton order once (63-bit codes of the AABB centroids, ties by primitive id; rocPRIM
// radix sort) and stay there; every subtree is a contiguous range of that order, so the box of ANY candidate subtree is
// a range query on a static array — answered by a bottom-up range tree over the sorted leaf boxes (min/max are exact
// and associative: a query returns the same bits as a sequential merge).  Two builders on top of that:
//   PT_BVH_DEVICE_LBVH  Karras 2012: the hierarchy of the Morton keys' common prefixes, every inner node found
//                       independently (one thread per node), boxes by range query — no bottom-up atomics pass
//   PT_BVH_DEVICE_SAH   top-down, level-synchronous: every node picks, among ALL cuts of its range, the one with the
//                       smallest surface-area cost  SA(left) * n_left + SA(right) * n_right  (one thread per cut position,
//                       per-node argmin by 64-bit atomicMin on {cost bits, position}: deterministic); a depth cap of
//                       ceil(log2 N) + 5 levels turns into median cuts where it binds (LDS stack entries cost occupancy)
// Output: the reference's own node layout (pt_bvh_node = BVHNode, bvh.cuh:7-15) on the HOST, so that the caller can
// hand the very same tree to pt_scene_create and to the CPU oracle.  Images on these trees are bit-identical between
// device and oracle (tests/test_device_bvh.py); against the reference tree they differ only where two primitives tie on t.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_internal.h"

namespace {

#define HIPB(expr)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return pt_fail(e_ == hipErrorNoDevice ? PT_ERR_NO_DEVICE : PT_ERR_DEVICE,                \
                           std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

struct Box {
    float lo[3], hi[3];
};

__device__ __forceinline__ Box box_empty() {
    Box b;
    for (int k = 0; k < 3; k++) { b.lo[k] = __builtin_inff(); b.hi[k] = -__builtin_inff(); }
    return b;
}
__device__ __forceinline__ void box_merge(Box& a, const Box& b) {
    for (int k = 0; k < 3; k++) { a.lo[k] = fminf(a.lo[k], b.lo[k]); a.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
}
__device__ __forceinline__ float box_area(const Box& b) {
    const float dx = fmaxf(b.hi[0] - b.lo[0], 0.0f), dy = fmaxf(b.hi[1] - b.lo[1], 0.0f), dz = fmaxf(b.hi[2] - b.lo[2], 0.0f);
    return 2.0f * (dx * dy + dy * dz + dx * dz);
}

struct DevShape {          // pt_shape with the mesh resolved to offsets into the flat vertex / index arrays
    int32_t type;
    float center[3], radius;
    int32_t index_base;    // triangle: first of its 3 indices in `indices`
    int32_t vertex_base;   // triangle: its mesh's first vertex in `positions`
};

// scene.cpp:124-145: one AABB per primitive (sphere: centre -+ radius; triangle: component-wise min / max of its vertices)
__global__ void prim_bounds_kernel(const DevShape* __restrict__ shapes, const float* __restrict__ positions,
                                   const int32_t* __restrict__ indices, int n, Box* __restrict__ boxes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevShape s = shapes[i];
    Box b;
    if (s.type == PT_SHAPE_SPHERE) {
        for (int k = 0; k < 3; k++) { b.lo[k] = s.center[k] - s.radius; b.hi[k] = s.center[k] + s.radius; }
    } else {
        const float* p0 = positions + 3 * (size_t)(s.vertex_base + indices[s.index_base]);
        const float* p1 = positions + 3 * (size_t)(s.vertex_base + indices[s.index_base + 1]);
        const float* p2 = positions + 3 * (size_t)(s.vertex_base + indices[s.index_base + 2]);
        for (int k = 0; k < 3; k++) {
            b.lo[k] = fminf(fminf(p0[k], p1[k]), p2[k]);
            b.hi[k] = fmaxf(fmaxf(p0[k], p1[k]), p2[k]);
        }
    }
    boxes[i] = b;
}

// bounds of the centroids: block partials, then one block over the partials
__global__ __launch_bounds__(256) void centroid_bounds_kernel(const Box* __restrict__ boxes, int n, int from_partials, Box* __restrict__ out) {
    __shared__ Box sh[256];
    Box acc = box_empty();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Box b = boxes[i];
        if (!from_partials)
            for (int k = 0; k < 3; k++) { const float c = (b.hi[k] + b.lo[k]) * 0.5f; b.lo[k] = c; b.hi[k] = c; }
        box_merge(acc, b);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) box_merge(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

__device__ __forceinline__ uint64_t expand_bits21(uint64_t v) {      // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x001f00000000ffffull;
    v = (v | (v << 16)) & 0x001f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// 63-bit Morton code of the centroid (21 bits per axis: one far-away primitive — a ground sphere of radius 100 under a
// unit-sized scene — stretches the centroid bounds by two orders of magnitude, and a 10-bit grid would then see the whole
// rest of the scene in a handful of cells), with the primitive id as the sorted value; the sort is stable, so equal codes
// stay in id order: a total, deterministic order.
__global__ void morton_kernel(const Box* __restrict__ boxes, const Box* __restrict__ cbounds, int n, uint64_t* __restrict__ codes,
                              uint32_t* __restrict__ ids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Box b = boxes[i];
    const Box cb = cbounds[0];
    uint64_t q[3];
    for (int k = 0; k < 3; k++) {
        const float c = (b.hi[k] + b.lo[k]) * 0.5f;
        const float ext = cb.hi[k] - cb.lo[k];
        float t = ext > 0.0f ? (c - cb.lo[k]) / ext : 0.0f;
        t = fminf(fmaxf(t * 2097152.0f, 0.0f), 2097151.0f);
        q[k] = (uint64_t)t;
    }
    codes[i] = (expand_bits21(q[0]) << 2) | (expand_bits21(q[1]) << 1) | expand_bits21(q[2]);
    ids[i] = (uint32_t)i;
}

// ---- range tree over the sorted leaf boxes: tree[P + i] = box of sorted leaf i (empty beyond n), tree[k] = tree[2k] U tree[2k+1]
__global__ void rangetree_leaves_kernel(const Box* __restrict__ boxes, const uint32_t* __restrict__ ids, int n, int P, Box* __restrict__ tree) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    tree[P + i] = i < n ? boxes[ids[i]] : box_empty();
}
__global__ void rangetree_level_kernel(Box* __restrict__ tree, int first, int count) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    Box b = tree[2 * (first + k)];
    box_merge(b, tree[2 * (first + k) + 1]);
    tree[first + k] = b;
}
// union of sorted leaves [a, b)  (a < b)
__device__ __forceinline__ Box range_box(const Box* __restrict__ tree, int P, int a, int b) {
    Box r = box_empty();
    int l = a + P, h = b + P;
    while (l < h) {
        if (l & 1) box_merge(r, tree[l++]);
        if (h & 1) box_merge(r, tree[--h]);
        l >>= 1; h >>= 1;
    }
    return r;
}

__device__ __forceinline__ void write_node(pt_bvh_node* out, int slot, const Box& b, int left, int right, int prim) {
    pt_bvh_node nd;
    for (int k = 0; k < 3; k++) { nd.bmin[k] = b.lo[k]; nd.bmax[k] = b.hi[k]; }
    nd.left = left; nd.right = right; nd.prim = prim;
    out[slot] = nd;
}

// ---- LBVH (Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012) ----------
// Virtual key of sorted position i: {morton code, i} — positions of equal codes are consecutive integers, so a run of
// duplicates becomes a balanced subtree.  delta = length of the common prefix of two virtual keys (-1 outside [0,n)).
__device__ __forceinline__ int lbvh_delta(const uint64_t* __restrict__ codes, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t x = codes[i] ^ codes[j];
    return x ? __builtin_clzll(x) : 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);     // (i != j here)
}

// Output layout: leaves at slots [0, n) in sorted order, inner node i at slot n + i, root = slot n (inner node 0).
__global__ void lbvh_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ ids, const Box* __restrict__ tree, int P, int n,
                            pt_bvh_node* __restrict__ out, int* __restrict__ parent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) write_node(out, i, tree[P + i], -1, -1, (int)ids[i]);
    if (i >= n - 1) return;
    const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    for (int div = 2;; div *= 2) {
        const int t = (l + div - 1) / div;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int left = lo == gamma ? gamma : n + gamma;              // leaf slot or inner slot
    const int right = hi == gamma + 1 ? gamma + 1 : n + gamma + 1;
    parent[left] = n + i;
    parent[right] = n + i;
    if (i == 0) parent[n] = -1;
    write_node(out, n + i, range_box(tree, P, lo, hi + 1), left, right, -1);
}

// depth of the tree, leaves count 1 (computeMaxDepth, bvh.cu:56-65)
__global__ void depth_kernel(const int* __restrict__ parent, int n, int* __restrict__ max_depth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int d = 1;
    for (int p = parent[i]; p >= 0; p = parent[p]) d++;
    atomicMax(max_depth, d);
}

// ---- SAH over the Morton order, top-down, one level per launch -------------------------------------------------------
// Per sorted position i: the node that currently contains it, as (lo, hi, base): the subtree of [lo, hi) owns the output
// slots [base, base + 2(hi-lo) - 1) with its root last (the host builder's post-order layout, scene_build.cpp), so child
// slots follow from the cut without any allocation.  hi == lo + 1 -> the position's leaf has been written: inactive.
struct SahPos {
    int lo, hi, base;
};

constexpr unsigned long long kNoCut = ~0ull;

// level 1: one node [0, n) that owns the slots [0, 2n-1)
__global__ void sah_init_kernel(SahPos* __restrict__ pos, int n, unsigned long long* __restrict__ best) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pos[i] = SahPos{0, n, 0};
    if (i == 0) best[2 * n - 2] = kNoCut;
}

// cost of cutting node [lo,hi) in front of position i, for every i in (lo, hi); argmin into best[root slot]
__global__ __launch_bounds__(256) void sah_cost_kernel(const SahPos* __restrict__ pos, const Box* __restrict__ tree, int P, int n,
                                                       unsigned long long* __restrict__ best) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    SahPos p{0, 0, 0};
    if (i < n) p = pos[i];
    const bool active = i < n && p.hi - p.lo > 1 && i > p.lo;
    unsigned long long key = kNoCut;
    if (active) {
        const Box l = range_box(tree, P, p.lo, i), r = range_box(tree, P, i, p.hi);
        const float cost = box_area(l) * (float)(i - p.lo) + box_area(r) * (float)(p.hi - i);
        // costs are >= 0 (or +inf / NaN for degenerate boxes: NaN's bit pattern sorts last among positives too)
        key = ((unsigned long long)__builtin_bit_cast(uint32_t, cost) << 32) | (uint32_t)i;
    }
    // lanes of a wave mostly sit in the same node at the top levels: one atomic per wave there instead of 64 on one address
    const int slot = p.base + 2 * (p.hi - p.lo) - 2;
    const int slot0 = __builtin_amdgcn_readfirstlane(slot);
    if (__all(!active || slot == slot0)) {
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = o < key ? o : key;
        }
        const unsigned long long any_active = __ballot(active);
        if (any_active && (threadIdx.x & 63) == __builtin_ctzll(any_active)) atomicMin(&best[slot], key);
    } else if (active) {
        atomicMin(&best[slot], key);
    }
}

// apply the cuts: every position moves into its child; the first position of a node writes the node, of a new leaf the leaf
__global__ __launch_bounds__(256) void sah_split_kernel(SahPos* __restrict__ pos, const Box* __restrict__ tree, int P, int n,
                                                        const uint32_t* __restrict__ ids, unsigned long long* __restrict__ best,
                                                        int depth, int max_depth, pt_bvh_node* __restrict__ out,
                                                        int* __restrict__ parent, int* __restrict__ n_active) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SahPos p = pos[i];
    const int cnt = p.hi - p.lo;
    if (cnt <= 1) return;
    const int slot = p.base + 2 * cnt - 2;
    int m = (int)(uint32_t)best[slot];
    // depth cap: once the levels left only just suffice for a balanced subtree, cut at the median (depth counts leaves as 1)
    int need = 0;
    while ((1 << need) < cnt) need++;
    if (best[slot] == kNoCut || depth + need >= max_depth) m = p.lo + cnt / 2;
    const int nl = m - p.lo, nr = p.hi - m;
    const int lbase = p.base, rbase = p.base + 2 * nl - 1;
    const int lslot = lbase + 2 * nl - 2, rslot = rbase + 2 * nr - 2;
    if (i == p.lo) {
        write_node(out, slot, range_box(tree, P, p.lo, p.hi), lslot, rslot, -1);
        parent[lslot] = slot;
        parent[rslot] = slot;
        if (depth == 1) parent[slot] = -1;
    }
    SahPos c;
    if (i < m) c = SahPos{p.lo, m, lbase}; else c = SahPos{m, p.hi, rbase};
    pos[i] = c;
    if (c.hi - c.lo == 1) {
        write_node(out, c.base, tree[P + i], -1, -1, (int)ids[i]);       // a one-leaf subtree owns exactly slot `base`
    } else {
        if (i == c.lo) { best[c.base + 2 * (c.hi - c.lo) - 2] = kNoCut; atomicAdd(n_active, 1); }
    }
}

template <class T>
struct Dev {
    T* p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    int alloc(size_t count) {
        HIPB(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)));
        return PT_OK;
    }
};

// Everything after the primitive boxes: Morton order, range tree, hierarchy, depth, copy back.  d_boxes: n boxes on the device.
int bvh_build_core(const Box* d_boxes, int n, int method, pt_bvh_node* out_nodes, int32_t* out_root, int32_t* out_depth,
                   hipEvent_t e0, hipEvent_t e1, double* out_build_ms) {
    int P = 1;
    while (P < n) P <<= 1;
    const int n_nodes = 2 * n - 1;
    Dev<Box> d_part, d_tree;
    Dev<uint64_t> d_keys, d_keys2; Dev<uint32_t> d_ids, d_ids2; Dev<pt_bvh_node> d_out; Dev<int> d_parent, d_misc; Dev<SahPos> d_sp;
    Dev<unsigned long long> d_best; Dev<unsigned char> d_tmp;
    int rc;
    if ((rc = d_part.alloc(1024 + 1)) || (rc = d_tree.alloc(2 * (size_t)P)) || (rc = d_keys.alloc(n)) ||
        (rc = d_keys2.alloc(n)) || (rc = d_ids.alloc(n)) || (rc = d_ids2.alloc(n)) || (rc = d_out.alloc(n_nodes)) || (rc = d_parent.alloc(n_nodes)) || (rc = d_misc.alloc(4)))
        return rc;
    size_t tmp_bytes = 0;
    HIPB(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys.p, d_keys2.p, d_ids.p, d_ids2.p, (size_t)n, 0, 63, nullptr));
    if ((rc = d_tmp.alloc(tmp_bytes))) return rc;
    if (method == PT_BVH_DEVICE_SAH && ((rc = d_sp.alloc(n)) || (rc = d_best.alloc(n_nodes)))) return rc;

    const int B = 256;
    const auto G = [&](int count) { return dim3((unsigned)((count + B - 1) / B)); };
    const int parts = std::min(1024, (n + B - 1) / B);
    hipLaunchKernelGGL(centroid_bounds_kernel, dim3(parts), dim3(B), 0, nullptr, d_boxes, n, 0, d_part.p + 1);
    hipLaunchKernelGGL(centroid_bounds_kernel, dim3(1), dim3(B), 0, nullptr, d_part.p + 1, parts, 1, d_part.p);
    hipLaunchKernelGGL(morton_kernel, G(n), dim3(B), 0, nullptr, d_boxes, d_part.p, n, d_keys.p, d_ids.p);
    HIPB(hipGetLastError());
    HIPB(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_keys.p, d_keys2.p, d_ids.p, d_ids2.p, (size_t)n, 0, 63, nullptr));
    const uint64_t* keys = d_keys2.p;
    const uint32_t* ids = d_ids2.p;
    hipLaunchKernelGGL(rangetree_leaves_kernel, G(P), dim3(B), 0, nullptr, d_boxes, ids, n, P, d_tree.p);
    for (int first = P / 2; first >= 1; first /= 2)
        hipLaunchKernelGGL(rangetree_level_kernel, G(first), dim3(B), 0, nullptr, d_tree.p, first, first);
    HIPB(hipGetLastError());
    HIPB(hipMemsetAsync(d_misc.p, 0, 4 * sizeof(int), nullptr));

    int root = 0;
    if (method == PT_BVH_DEVICE_LBVH) {
        hipLaunchKernelGGL(lbvh_kernel, G(n), dim3(B), 0, nullptr, keys, ids, d_tree.p, P, n, d_out.p, d_parent.p);
        HIPB(hipGetLastError());
        root = n;
    } else {
        hipLaunchKernelGGL(sah_init_kernel, G(n), dim3(B), 0, nullptr, d_sp.p, n, d_best.p);
        // Depth cap (leaves count 1): ceil(log2 n) + 5.  The kernel keeps one LDS stack entry per level and lane, so a
        // deep tree takes LDS from the top-of-tree cache and, beyond ~30 levels, a resident block per CU; the cap costs
        // little (bunny: 22.6 inner visits per segment uncapped at depth 28, 23.3 capped at 24, 32.1 at 22)
        int lg = 0;
        while ((1 << lg) < n) lg++;
        const int max_depth = std::min(48, std::max(8, lg + 5));
        for (int depth = 1; depth < max_depth; depth++) {
            hipLaunchKernelGGL(sah_cost_kernel, G(n), dim3(B), 0, nullptr, d_sp.p, d_tree.p, P, n, d_best.p);
            HIPB(hipMemsetAsync(d_misc.p + 1, 0, sizeof(int), nullptr));
            hipLaunchKernelGGL(sah_split_kernel, G(n), dim3(B), 0, nullptr, d_sp.p, d_tree.p, P, n, ids, d_best.p, depth, max_depth,
                               d_out.p, d_parent.p, d_misc.p + 1);
            HIPB(hipGetLastError());
            if (depth % 4 == 0) {           // nodes still to be cut after this level?
                int active = 0;
                HIPB(hipMemcpy(&active, d_misc.p + 1, sizeof(int), hipMemcpyDeviceToHost));
                if (active == 0) break;
            }
        }
        root = n_nodes - 1;
    }
    hipLaunchKernelGGL(depth_kernel, G(n_nodes), dim3(B), 0, nullptr, d_parent.p, n_nodes, d_misc.p);
    HIPB(hipGetLastError());
    HIPB(hipEventRecord(e1, nullptr));
    HIPB(hipEventSynchronize(e1));
    float ms = 0;
    HIPB(hipEventElapsedTime(&ms, e0, e1));
    int depth = 0;
    HIPB(hipMemcpy(&depth, d_misc.p, sizeof(int), hipMemcpyDeviceToHost));
    HIPB(hipMemcpy(out_nodes, d_out.p, (size_t)n_nodes * sizeof(pt_bvh_node), hipMemcpyDeviceToHost));
    *out_root = root;
    if (out_depth) *out_depth = depth;
    if (out_build_ms) *out_build_ms = ms;
    return PT_OK;
}

struct EvGuard {
    hipEvent_t a = nullptr, b = nullptr;
    ~EvGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

}  // namespace

extern "C" int pt_bvh_build_device(const pt_scene_desc* d, int method, pt_bvh_node* out_nodes, int32_t* out_root,
                                   int32_t* out_depth, double* out_build_ms) {
    if (!d || !out_nodes || !out_root) return pt_fail(PT_ERR_INVALID_ARG, "null argument");
    if (method != PT_BVH_DEVICE_LBVH && method != PT_BVH_DEVICE_SAH) return pt_fail(PT_ERR_INVALID_ARG, "unknown device BVH method");
    const int n = d->num_shapes;
    if (n <= 0 || !d->shapes) return pt_fail(PT_ERR_BAD_SCENE, "scene has no shapes");
    if (d->num_meshes < 0 || (d->num_meshes > 0 && !d->meshes)) return pt_fail(PT_ERR_BAD_SCENE, "bad mesh array");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return pt_fail(PT_ERR_NO_DEVICE, "no HIP device available: the device BVH builder has no CPU fallback");

    // ---- flatten the meshes: one vertex array, one index array, per-triangle offsets
    std::vector<int32_t> vbase(d->num_meshes), ibase(d->num_meshes);
    size_t nv = 0, ni = 0;
    for (int m = 0; m < d->num_meshes; m++) {
        const pt_mesh& me = d->meshes[m];
        if (me.num_vertices <= 0 || me.num_faces <= 0 || !me.positions || !me.indices) return pt_fail(PT_ERR_BAD_SCENE, "empty mesh");
        vbase[m] = (int32_t)nv; ibase[m] = (int32_t)ni;
        nv += (size_t)me.num_vertices; ni += 3 * (size_t)me.num_faces;
    }
    if (nv > 0x7fffffffull / 3 || ni > 0x7fffffffull) return pt_fail(PT_ERR_BAD_SCENE, "meshes too large for 32-bit offsets");
    std::vector<float> positions(3 * nv);
    std::vector<int32_t> indices(ni);
    for (int m = 0; m < d->num_meshes; m++) {
        const pt_mesh& me = d->meshes[m];
        std::memcpy(positions.data() + 3 * (size_t)vbase[m], me.positions, 12 * (size_t)me.num_vertices);
        std::memcpy(indices.data() + ibase[m], me.indices, 12 * (size_t)me.num_faces);
        for (size_t k = 0; k < 3 * (size_t)me.num_faces; k++)
            if (me.indices[k] < 0 || me.indices[k] >= me.num_vertices) return pt_fail(PT_ERR_BAD_SCENE, "vertex index out of range");
    }
    std::vector<DevShape> shapes(n);
    for (int i = 0; i < n; i++) {
        const pt_shape& s = d->shapes[i];
        DevShape& o = shapes[i];
        std::memset(&o, 0, sizeof o);
        o.type = s.type;
        if (s.type == PT_SHAPE_SPHERE) {
            std::memcpy(o.center, s.center, 12);
            o.radius = s.radius;
        } else if (s.type == PT_SHAPE_TRIANGLE) {
            if (s.mesh_index < 0 || s.mesh_index >= d->num_meshes) return pt_fail(PT_ERR_BAD_SCENE, "triangle mesh index out of range");
            if (s.face_index < 0 || s.face_index >= d->meshes[s.mesh_index].num_faces) return pt_fail(PT_ERR_BAD_SCENE, "triangle face index out of range");
            o.index_base = ibase[s.mesh_index] + 3 * s.face_index;
            o.vertex_base = vbase[s.mesh_index];
        } else {
            return pt_fail(PT_ERR_BAD_SCENE, "unknown shape type");
        }
    }

    if (n == 1) {       // a single leaf is the whole tree
        pt_bvh_node nd{};
        const pt_shape& s = d->shapes[0];
        if (s.type == PT_SHAPE_SPHERE) {
            for (int k = 0; k < 3; k++) { nd.bmin[k] = s.center[k] - s.radius; nd.bmax[k] = s.center[k] + s.radius; }
        } else {
            const float* P3 = positions.data();
            for (int k = 0; k < 3; k++) {
                const float a = P3[3 * (size_t)(shapes[0].vertex_base + indices[shapes[0].index_base]) + k];
                const float b = P3[3 * (size_t)(shapes[0].vertex_base + indices[shapes[0].index_base + 1]) + k];
                const float c = P3[3 * (size_t)(shapes[0].vertex_base + indices[shapes[0].index_base + 2]) + k];
                nd.bmin[k] = std::min(std::min(a, b), c); nd.bmax[k] = std::max(std::max(a, b), c);
            }
        }
        nd.left = nd.right = -1; nd.prim = 0;
        out_nodes[0] = nd; *out_root = 0;
        if (out_depth) *out_depth = 1;
        if (out_build_ms) *out_build_ms = 0.0;
        return PT_OK;
    }

    Dev<DevShape> d_shapes; Dev<float> d_pos; Dev<int32_t> d_idx; Dev<Box> d_boxes;
    int rc;
    if ((rc = d_shapes.alloc(n)) || (rc = d_pos.alloc(positions.size())) || (rc = d_idx.alloc(indices.size())) || (rc = d_boxes.alloc(n)))
        return rc;
    HIPB(hipMemcpy(d_shapes.p, shapes.data(), shapes.size() * sizeof(DevShape), hipMemcpyHostToDevice));
    if (!positions.empty()) HIPB(hipMemcpy(d_pos.p, positions.data(), positions.size() * 4, hipMemcpyHostToDevice));
    if (!indices.empty()) HIPB(hipMemcpy(d_idx.p, indices.data(), indices.size() * 4, hipMemcpyHostToDevice));
    EvGuard evg;
    HIPB(hipEventCreate(&evg.a));
    HIPB(hipEventCreate(&evg.b));
    HIPB(hipEventRecord(evg.a, nullptr));
    hipLaunchKernelGGL(prim_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_shapes.p, d_pos.p, d_idx.p, n, d_boxes.p);
    HIPB(hipGetLastError());
    return bvh_build_core(d_boxes.p, n, method, out_nodes, out_root, out_depth, evg.a, evg.b, out_build_ms);
}
