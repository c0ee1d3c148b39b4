// pt_scene_prep.hip — pieces of pt_scene_create that run on the device instead of in host loops over millions of nodes.
// (The reference does all of its scene preparation on the host: Scene::Scene scene.cpp:11-153, construct_bvh bvh.cu:16-54,
// per-element cudaMemcpy uploads scene.h:102-109; README.md:123,132 reports 10-57 s of it.)
#include <hip/hip_runtime.h>

#include <string>

#include <algorithm>
#include <cstring>
#include <vector>

#include "pt_internal.h"
#include "pt_layout.h"
#include "pt_scene_prep.h"

namespace {

#define HIPP(expr)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return pt_fail(e_ == hipErrorNoDevice ? PT_ERR_NO_DEVICE : PT_ERR_DEVICE,                \
                           std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

// parent[child] = node, with the side in bit 31 (set = the child is the RIGHT one)
__global__ void parents_kernel(const pt_bvh_node* __restrict__ pool, int num_nodes, int32_t* __restrict__ parent) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_nodes) return;
    const pt_bvh_node nd = pool[k];
    if (nd.prim != -1) return;
    parent[nd.left] = k;
    parent[nd.right] = (int32_t)((uint32_t)k | 0x80000000u);
}

// One thread per pool node that is a leaf: up to the root once to learn the depth, up again to write the path top-down.
__global__ void tie_tables_kernel(const pt_bvh_node* __restrict__ pool, int num_nodes, int root, const int32_t* __restrict__ parent,
                                  const int32_t* __restrict__ inner_of_pool, int levels, unsigned long long* __restrict__ path,
                                  int32_t* __restrict__ anc, unsigned int* __restrict__ failed) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_nodes) return;
    const int32_t prim = pool[k].prim;
    if (prim == -1) return;
    int depth = 0;
    for (int v = k; v != root; depth++) {
        if (depth > levels) { atomicAdd(failed, 1u); return; }
        v = (int32_t)((uint32_t)parent[v] & 0x7fffffffu);
    }
    unsigned long long turns = 0ull;
    int level = depth;
    for (int v = k; v != root;) {
        const uint32_t pw = (uint32_t)parent[v];
        level--;
        v = (int32_t)(pw & 0x7fffffffu);
        if (pw & 0x80000000u) turns |= 1ull << level;
        anc[(size_t)prim * (size_t)levels + (size_t)level] = inner_of_pool[v];
    }
    path[prim] = turns;
}

}  // namespace

int ptp::tie_tables_device(const pt_bvh_node* pool_dev, int num_nodes, int root, const int32_t* inner_of_pool_dev, int N, int levels,
                           int32_t* scratch_parent_dev, unsigned long long* path_dev, int32_t* anc_dev) {
    (void)N;
    unsigned int* failed = nullptr;
    HIPP(hipMalloc(reinterpret_cast<void**>(&failed), sizeof(unsigned int)));
    struct Free { void* p; ~Free() { (void)hipFree(p); } } guard{failed};
    HIPP(hipMemsetAsync(failed, 0, sizeof(unsigned int), nullptr));
    const int T = 256, G = (num_nodes + T - 1) / T;
    hipLaunchKernelGGL(parents_kernel, dim3(G), dim3(T), 0, nullptr, pool_dev, num_nodes, scratch_parent_dev);
    hipLaunchKernelGGL(tie_tables_kernel, dim3(G), dim3(T), 0, nullptr, pool_dev, num_nodes, root, scratch_parent_dev, inner_of_pool_dev,
                       levels, path_dev, anc_dev, failed);
    HIPP(hipGetLastError());
    unsigned int f = 0;
    HIPP(hipMemcpy(&f, failed, sizeof f, hipMemcpyDeviceToHost));
    if (f) return pt_fail(PT_ERR_DEVICE, "tie_tables_device: a leaf did not reach the root (internal error)");
    return PT_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// relay_tree_device
// ------------------------------------------------------------------------------------------------------------------------------
namespace {

using ptl::DNode;

enum : unsigned int {             // validation outcomes, in the order the host path reports them
    kErrChildRange = 1u, kErrLeafPrim = 2u, kErrNodeTwice = 4u, kErrPrimTwice = 8u, kErrPrimMissing = 16u, kErrNotATree = 32u,
    kErrTooDeep = 64u,
};
struct RelayCtl {
    unsigned int err;
    int max_level;                // deepest node's level, root = 0 (edges)
    unsigned int not_nested;
    unsigned int pad;
};
constexpr int kWalkLimit = 160;   // a walk to the root longer than this is a cycle (caller's trees: <= 64 levels; own trees: <= 2 log2 n + 18)

__global__ void relay_check_kernel(const pt_bvh_node* __restrict__ pool, int num_nodes, int N, int validate, int32_t* __restrict__ parent,
                                   unsigned int* __restrict__ refs, unsigned int* __restrict__ prim_refs, RelayCtl* __restrict__ ctl) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_nodes) return;
    const pt_bvh_node nd = pool[k];
    if (nd.prim != -1) {
        if (nd.prim < 0 || nd.prim >= N) { atomicOr(&ctl->err, kErrLeafPrim); return; }
        if (validate) atomicAdd(&prim_refs[nd.prim], 1u);
        return;
    }
    if (nd.left < 0 || nd.left >= num_nodes || nd.right < 0 || nd.right >= num_nodes) { atomicOr(&ctl->err, kErrChildRange); return; }
    parent[nd.left] = k;
    parent[nd.right] = (int32_t)((uint32_t)k | 0x80000000u);
    if (validate) { atomicAdd(&refs[nd.left], 1u); atomicAdd(&refs[nd.right], 1u); }
}
__global__ void relay_refs_kernel(const unsigned int* __restrict__ refs, const unsigned int* __restrict__ prim_refs, int num_nodes, int root,
                                  int N, RelayCtl* __restrict__ ctl) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < num_nodes) {
        const unsigned int want = k == root ? 0u : 1u;
        if (refs[k] != want) atomicOr(&ctl->err, refs[k] > want ? kErrNodeTwice : kErrNotATree);
    }
    if (k < N && prim_refs[k] != 1u) atomicOr(&ctl->err, prim_refs[k] > 1u ? kErrPrimTwice : kErrPrimMissing);
}
// level of every node (root 0) by a walk to the root
__global__ void relay_level_kernel(int num_nodes, int root, const int32_t* __restrict__ parent, int32_t* __restrict__ level,
                                   RelayCtl* __restrict__ ctl) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_nodes) return;
    int e = 0;
    for (int v = k; v != root; e++) {
        if (e >= kWalkLimit) { atomicOr(&ctl->err, kErrNotATree); level[k] = -1; return; }
        v = (int32_t)((uint32_t)parent[v] & 0x7fffffffu);
    }
    level[k] = e;
    atomicMax(&ctl->max_level, e);
}
// leaves below and Strahler number, the nodes of one level at a time (deepest first): the children are complete
__global__ void relay_bottom_up_kernel(const pt_bvh_node* __restrict__ pool, int num_nodes, const int32_t* __restrict__ level, int at,
                                       int32_t* __restrict__ leaves, int32_t* __restrict__ need) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_nodes || level[k] != at) return;
    const pt_bvh_node nd = pool[k];
    if (nd.prim != -1) { leaves[k] = 1; need[k] = 0; return; }
    const int32_t a = need[nd.left], b = need[nd.right];
    leaves[k] = leaves[nd.left] + leaves[nd.right];
    need[k] = a == b ? a + 1 : (a > b ? a : b);
}
__device__ __forceinline__ bool relay_swapped(const pt_bvh_node& nd, const int32_t* __restrict__ need, int internal) {
    return internal && need[nd.left] > need[nd.right];
}
// position of every inner node in the DFS pre-order over inner nodes (first child's subtree right after the parent)
__global__ void relay_preorder_kernel(const pt_bvh_node* __restrict__ pool, int num_nodes, int root, const int32_t* __restrict__ parent,
                                      const int32_t* __restrict__ leaves, const int32_t* __restrict__ need, int internal,
                                      int32_t* __restrict__ pre) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_nodes) return;
    if (pool[k].prim != -1) { pre[k] = -1; return; }
    int32_t idx = 0;
    for (int c = k; c != root;) {
        const int p = (int32_t)((uint32_t)parent[c] & 0x7fffffffu);
        const pt_bvh_node pn = pool[p];
        const int first = relay_swapped(pn, need, internal) ? pn.right : pn.left;
        idx += 1 + (c != first ? leaves[first] - 1 : 0);          // a subtree of L leaves holds L - 1 inner nodes
        c = p;
    }
    pre[k] = idx;
}
__device__ __forceinline__ double relay_area(const pt_bvh_node& b) {
    const double x = (double)b.bmax[0] - b.bmin[0], y = (double)b.bmax[1] - b.bmin[1], z = (double)b.bmax[2] - b.bmin[2];
    return 2.0 * (x * y + y * z + z * x);
}
// The `want` inner nodes with the largest boxes, parents before children (pt_api.hip: convert_tree has the argument): the host
// pops a max-heap of (area, -pre-order position) — a total order, so the sequence of picks is simply "the largest entry of the
// frontier, `want` times".  One wave: the frontier sits unsorted in LDS, every pick is a 64-lane argmax over it (at most `want` + 1
// entries), lanes 0 and 1 fetch the picked node's two children in one round trip (an entry carries its children's indices), the
// children that are inner nodes take the freed place and the end of the list.  top[j] = pool index of the j-th pick.
// (First version: one thread and a binary heap in LDS — 2.7 us a pick, 1.4 ms per tree; the heap's dependent LDS steps cost more
// than its global loads.)
constexpr int kHeapCap = 2048;
__global__ __launch_bounds__(64) void relay_top_kernel(const pt_bvh_node* __restrict__ pool, int root, const int32_t* __restrict__ pre, int want,
                                                       int32_t* __restrict__ top, int32_t* __restrict__ top_pre, int32_t* __restrict__ n_top) {
    __shared__ double f_area[kHeapCap];
    __shared__ int32_t f_neg[kHeapCap], f_node[kHeapCap], f_left[kHeapCap], f_right[kHeapCap];
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    int size = 0, picked = 0;                            // wave-uniform
    if (want > 0) {
        if (lane == 0) {
            const pt_bvh_node nd = pool[root];
            f_area[0] = __builtin_huge_val(); f_neg[0] = 0; f_node[0] = root; f_left[0] = nd.left; f_right[0] = nd.right;
        }
        size = 1;
    }
    __syncthreads();
    while (size > 0 && picked < want) {
        double ba = 0.0;
        int32_t bn = 0;
        int bi = -1;
        for (int k = lane; k < size; k += 64) {
            const double a = f_area[k];
            const int32_t g = f_neg[k];
            if (bi < 0 || a > ba || (a == ba && g > bn)) { ba = a; bn = g; bi = k; }
        }
        for (int off = 32; off; off >>= 1) {
            const double oa = __shfl_xor(ba, off);
            const int32_t og = __shfl_xor(bn, off);
            const int oi = __shfl_xor(bi, off);
            if (oi >= 0 && (bi < 0 || oa > ba || (oa == ba && og > bn))) { ba = oa; bn = og; bi = oi; }
        }
        const int32_t node = f_node[bi];
        const int32_t child = lane == 0 ? f_left[bi] : f_right[bi];
        if (lane == 0) { top_pre[picked] = -bn; top[picked] = node; }
        picked++;
        // lanes 0 and 1: the two children's records and pre-order positions
        bool inner = false;
        double c_area = 0.0;
        int32_t c_neg = 0, c_left = 0, c_right = 0;
        if (lane < 2) {
            const pt_bvh_node c = pool[child];
            inner = c.prim == -1;
            c_area = relay_area(c);
            c_neg = -pre[child];
            c_left = c.left; c_right = c.right;
        }
        const bool in0 = __shfl((int)inner, 0) != 0, in1 = __shfl((int)inner, 1) != 0;
        __syncthreads();                                 // every lane has read the frontier
        const int last = size - 1;
        if (!in0 && !in1) {
            if (lane == 0 && bi != last) {               // the pick's place goes to the last entry
                f_area[bi] = f_area[last]; f_neg[bi] = f_neg[last]; f_node[bi] = f_node[last]; f_left[bi] = f_left[last]; f_right[bi] = f_right[last];
            }
            size = last;
        } else {
            // the first inner child takes the pick's place, a second one the end of the list (dropped when the list is full,
            // as the heap version dropped it: `want` is at most kTopNodes = 512 in this library)
            const int at = lane == 0 ? (in0 ? bi : -1) : lane == 1 ? (in0 ? (size < kHeapCap ? size : -1) : bi) : -1;
            if (lane < 2 && inner && at >= 0) {
                f_area[at] = c_area; f_neg[at] = c_neg; f_node[at] = child; f_left[at] = c_left; f_right[at] = c_right;
            }
            if (in0 && in1 && size < kHeapCap) size++;
        }
        __syncthreads();
    }
    if (lane == 0) *n_top = picked;
}
// final DNode index: the picks first, in pick order; the rest keep their pre-order among themselves
__global__ void relay_final_index_kernel(const pt_bvh_node* __restrict__ pool, int num_nodes, const int32_t* __restrict__ pre,
                                         const int32_t* __restrict__ top_sorted_pre, const int32_t* __restrict__ top_rank_of_sorted,
                                         int n_top, int32_t* __restrict__ fin) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_nodes) return;
    if (pool[k].prim != -1) { fin[k] = -1; return; }
    const int32_t p = pre[k];
    int lo = 0, hi = n_top;                     // picks with a pre-order position below p
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (top_sorted_pre[mid] < p) lo = mid + 1; else hi = mid;
    }
    fin[k] = (lo < n_top && top_sorted_pre[lo] == p) ? top_rank_of_sorted[lo] : n_top + (p - lo);
}
__global__ void relay_emit_kernel(const pt_bvh_node* __restrict__ pool, int num_nodes, int root, const int32_t* __restrict__ need,
                                  const int32_t* __restrict__ fin, int internal, DNode* __restrict__ out, float* __restrict__ leaf_boxes,
                                  RelayCtl* __restrict__ ctl) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_nodes) return;
    const pt_bvh_node nd = pool[k];
    if (nd.prim != -1) {
        if (leaf_boxes)
            for (int q = 0; q < 3; q++) { leaf_boxes[6 * (size_t)nd.prim + q] = nd.bmin[q]; leaf_boxes[6 * (size_t)nd.prim + 3 + q] = nd.bmax[q]; }
        return;
    }
    const bool sw = relay_swapped(nd, need, internal);
    const int li = sw ? nd.right : nd.left, ri = sw ? nd.left : nd.right;
    const pt_bvh_node ln = pool[li], rn = pool[ri];
    if (k != root) {
        bool inside = true;
        for (int q = 0; q < 3; q++)
            inside = inside && ln.bmin[q] >= nd.bmin[q] && ln.bmax[q] <= nd.bmax[q] && rn.bmin[q] >= nd.bmin[q] && rn.bmax[q] <= nd.bmax[q];
        if (!inside) atomicAdd(&ctl->not_nested, 1u);
    }
    DNode o;
    for (int q = 0; q < 3; q++) { o.lmin[q] = ln.bmin[q]; o.lmax[q] = ln.bmax[q]; o.rmin[q] = rn.bmin[q]; o.rmax[q] = rn.bmax[q]; }
    o.left = ln.prim != -1 ? ~ln.prim : fin[li];
    o.right = rn.prim != -1 ? ~rn.prim : fin[ri];
    o.pad0 = 0; o.pad1 = 0;
    out[fin[k]] = o;
}

template <class T>
struct TmpBuf {
    T* p = nullptr;
    ~TmpBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T)); }
};

}  // namespace

int ptp::relay_tree_device(const pt_bvh_node* pool_dev, int num_nodes, int root, int N, bool internal, void* dnodes_dev,
                           int32_t* inner_of_pool_dev, float* leaf_boxes_dev, int block_threads, uint32_t top_nodes_max,
                           uint32_t lds_budget_max, RelayResult* out) {
    if (!pool_dev || !dnodes_dev || !inner_of_pool_dev || !out || N < 2 || num_nodes != 2 * N - 1 || root < 0 || root >= num_nodes)
        return pt_fail(PT_ERR_INVALID_ARG, "relay_tree_device: bad argument");
    const int T = 256, G = (num_nodes + T - 1) / T;
    const bool validate = !internal;
    // one allocation for the working arrays (a hipMalloc / hipFree pair each would cost as much as the kernels)
    struct { int32_t* p; } parent, level, leaves, need, pre, top, top_sorted, top_rank, n_top_dev;
    struct { unsigned int* p; } refs, prim_refs;
    struct { RelayCtl* p; } ctl;
    TmpBuf<unsigned char> arena;
    {
        size_t used = 0;
        auto take = [&used](size_t bytes) { const size_t at = used; used += (bytes + 255) & ~(size_t)255; return at; };
        const size_t nn = (size_t)num_nodes;
        const size_t o_parent = take(nn * 4), o_level = take(nn * 4), o_leaves = take(nn * 4), o_need = take(nn * 4), o_pre = take(nn * 4);
        const size_t o_top = take((size_t)top_nodes_max * 4), o_tops = take((size_t)top_nodes_max * 4), o_topr = take((size_t)top_nodes_max * 4), o_ntop = take(4);
        const size_t o_refs = take(validate ? nn * 4 : 4), o_prefs = take(validate ? (size_t)N * 4 : 4), o_ctl = take(sizeof(RelayCtl));
        HIPP(arena.alloc(used));
        unsigned char* b = arena.p;
        parent.p = reinterpret_cast<int32_t*>(b + o_parent); level.p = reinterpret_cast<int32_t*>(b + o_level); leaves.p = reinterpret_cast<int32_t*>(b + o_leaves);
        need.p = reinterpret_cast<int32_t*>(b + o_need); pre.p = reinterpret_cast<int32_t*>(b + o_pre); top.p = reinterpret_cast<int32_t*>(b + o_top);
        top_sorted.p = reinterpret_cast<int32_t*>(b + o_tops); top_rank.p = reinterpret_cast<int32_t*>(b + o_topr); n_top_dev.p = reinterpret_cast<int32_t*>(b + o_ntop);
        refs.p = reinterpret_cast<unsigned int*>(b + o_refs); prim_refs.p = reinterpret_cast<unsigned int*>(b + o_prefs); ctl.p = reinterpret_cast<RelayCtl*>(b + o_ctl);
    }
    HIPP(hipMemsetAsync(ctl.p, 0, sizeof(RelayCtl), nullptr));
    HIPP(hipMemsetAsync(parent.p, 0xff, (size_t)num_nodes * sizeof(int32_t), nullptr));        // "no parent": the walk never leaves the array
    if (validate) {
        HIPP(hipMemsetAsync(refs.p, 0, (size_t)num_nodes * sizeof(unsigned int), nullptr));
        HIPP(hipMemsetAsync(prim_refs.p, 0, (size_t)N * sizeof(unsigned int), nullptr));
    }
    RelayCtl h{};
    auto bad = [&](unsigned int e) {
        const char* msg = (e & kErrChildRange) ? "BVH child index out of range"
                        : (e & kErrLeafPrim) ? "leaf primitive id out of range"
                        : (e & kErrNodeTwice) ? "BVH node referenced twice"
                        : (e & kErrPrimTwice) ? "primitive referenced by two leaves"
                        : (e & kErrNotATree) ? "BVH is not a tree (cycle)"
                        : (e & kErrPrimMissing) ? "primitive not covered by any leaf"
                        : "BVH deeper than the traversal stack (reference cap 64, scene.h:251)";
        return pt_fail(PT_ERR_BAD_SCENE, msg);
    };
    hipLaunchKernelGGL(relay_check_kernel, dim3(G), dim3(T), 0, nullptr, pool_dev, num_nodes, N, validate ? 1 : 0, parent.p, refs.p, prim_refs.p, ctl.p);
    if (validate)
        hipLaunchKernelGGL(relay_refs_kernel, dim3(G), dim3(T), 0, nullptr, refs.p, prim_refs.p, num_nodes, root, N, ctl.p);
    HIPP(hipGetLastError());
    HIPP(hipMemcpy(&h, ctl.p, sizeof h, hipMemcpyDeviceToHost));
    if (h.err) return bad(h.err);                        // nothing below may follow a child index that was not checked
    // the parent word of the root stays 0xffffffff: & 0x7fffffff would be out of range, but no walk goes above the root
    hipLaunchKernelGGL(relay_level_kernel, dim3(G), dim3(T), 0, nullptr, num_nodes, root, parent.p, level.p, ctl.p);
    HIPP(hipGetLastError());
    HIPP(hipMemcpy(&h, ctl.p, sizeof h, hipMemcpyDeviceToHost));
    if (h.err) return bad(h.err);
    const int depth = h.max_level + 1;                   // levels, leaves counting
    if (validate && depth - 1 > 64 - 1) return bad(kErrTooDeep);
    for (int at = h.max_level; at >= 0; at--)
        hipLaunchKernelGGL(relay_bottom_up_kernel, dim3(G), dim3(T), 0, nullptr, pool_dev, num_nodes, level.p, at, leaves.p, need.p);
    hipLaunchKernelGGL(relay_preorder_kernel, dim3(G), dim3(T), 0, nullptr, pool_dev, num_nodes, root, parent.p, leaves.p, need.p,
                       internal ? 1 : 0, pre.p);
    HIPP(hipGetLastError());
    int32_t need_root = 0;
    HIPP(hipMemcpy(&need_root, need.p + root, sizeof need_root, hipMemcpyDeviceToHost));
    const int stack_need = internal ? need_root : std::max(depth - 1, 1);
    // the top of the tree, as many nodes as fit next to the traversal stacks within the largest LDS budget an option may ask for
    const uint32_t stack_bytes = (uint32_t)(block_threads / 64) * (uint32_t)(stack_need + 1) * 64u * 4u;
    int want = (int)std::min<size_t>(top_nodes_max, lds_budget_max > stack_bytes ? (lds_budget_max - stack_bytes) / sizeof(DNode) : 0);
    want = std::min(want, N - 1);
    hipLaunchKernelGGL(relay_top_kernel, dim3(1), dim3(64), 0, nullptr, pool_dev, root, pre.p, want, top.p, top_sorted.p, n_top_dev.p);
    HIPP(hipGetLastError());
    int32_t n_top = 0;
    HIPP(hipMemcpy(&n_top, n_top_dev.p, sizeof n_top, hipMemcpyDeviceToHost));
    if (n_top) {
        // pre-order positions of the picks, sorted, each with its rank among the picks (a few hundred numbers: host side)
        std::vector<int32_t> pre_of(n_top);
        HIPP(hipMemcpy(pre_of.data(), top_sorted.p, (size_t)n_top * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<int32_t> order(n_top);
        for (int j = 0; j < n_top; j++) order[j] = j;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return pre_of[a] < pre_of[b]; });
        std::vector<int32_t> sorted_pre(n_top), rank_of(n_top);
        for (int j = 0; j < n_top; j++) { sorted_pre[j] = pre_of[order[j]]; rank_of[j] = order[j]; }
        HIPP(hipMemcpy(top_sorted.p, sorted_pre.data(), (size_t)n_top * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPP(hipMemcpy(top_rank.p, rank_of.data(), (size_t)n_top * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(relay_final_index_kernel, dim3(G), dim3(T), 0, nullptr, pool_dev, num_nodes, pre.p, top_sorted.p, top_rank.p, n_top,
                       inner_of_pool_dev);
    hipLaunchKernelGGL(relay_emit_kernel, dim3(G), dim3(T), 0, nullptr, pool_dev, num_nodes, root, need.p, inner_of_pool_dev, internal ? 1 : 0,
                       reinterpret_cast<DNode*>(dnodes_dev), leaf_boxes_dev, ctl.p);
    HIPP(hipGetLastError());
    HIPP(hipMemcpy(&h, ctl.p, sizeof h, hipMemcpyDeviceToHost));
    out->depth = depth;
    out->stack_need = stack_need;
    out->top_avail = (uint32_t)n_top;
    out->nested = h.not_nested == 0 ? 1 : 0;
    out->num_inner = N - 1;
    return PT_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// prims_device
// ------------------------------------------------------------------------------------------------------------------------------
namespace {

using ptl::DNormals;
using ptl::DPrim;

struct MeshTab {
    int32_t vertex_base, face_base, num_vertices, num_faces, material_id, area_light_id;
};
enum : unsigned int { kPErrSphereMat = 1u, kPErrMeshIndex = 2u, kPErrFaceIndex = 4u, kPErrVertexIndex = 8u, kPErrShapeType = 16u };

__global__ void prims_kernel(const pt_shape* __restrict__ shapes, int n, const MeshTab* __restrict__ tab, int num_meshes, int num_materials,
                             const float* __restrict__ positions, const float* __restrict__ normals, const int32_t* __restrict__ indices,
                             DPrim* __restrict__ out_p, DNormals* __restrict__ out_n, unsigned int* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pt_shape s = shapes[i];
    DPrim p;
    DNormals nn;
    for (int k = 0; k < 9; k++) { p.v[k] = 0.0f; nn.n[k] = 0.0f; }
    p.info = 0; p.light = 0; p.pad = 0;
    nn.pad[0] = nn.pad[1] = nn.pad[2] = 0.0f;
    if (s.type == PT_SHAPE_SPHERE) {
        if (s.material_id < 0 || s.material_id >= num_materials) { atomicOr(&flags[0], kPErrSphereMat); return; }
        p.v[0] = s.center[0]; p.v[1] = s.center[1]; p.v[2] = s.center[2]; p.v[3] = s.radius;
        p.info = (int32_t)(0x80000000u | (uint32_t)s.material_id);
        p.light = s.area_light_id;
        flags[1] = 1u;                                       // a sphere (same value from every writer)
    } else if (s.type == PT_SHAPE_TRIANGLE) {
        if (s.mesh_index < 0 || s.mesh_index >= num_meshes) { atomicOr(&flags[0], kPErrMeshIndex); return; }
        const MeshTab m = tab[s.mesh_index];
        if (s.face_index < 0 || s.face_index >= m.num_faces) { atomicOr(&flags[0], kPErrFaceIndex); return; }
        const int32_t* idx = indices + 3 * ((size_t)m.face_base + (size_t)s.face_index);
        for (int k = 0; k < 3; k++) {
            const int32_t v = idx[k];
            if (v < 0 || v >= m.num_vertices) { atomicOr(&flags[0], kPErrVertexIndex); return; }
            for (int c = 0; c < 3; c++) {
                p.v[3 * k + c] = positions[3 * ((size_t)m.vertex_base + (size_t)v) + c];
                nn.n[3 * k + c] = normals[3 * ((size_t)m.vertex_base + (size_t)v) + c];
            }
        }
        p.info = m.material_id;
        p.light = m.area_light_id;
    } else {
        atomicOr(&flags[0], kPErrShapeType);
        return;
    }
    out_p[i] = p;
    out_n[i] = nn;
}

}  // namespace

int ptp::prims_device(const pt_scene_desc* d, void* prims_dev, void* normals_dev, int* has_sphere) {
    const int N = d->num_shapes;
    size_t nv = 0, nf = 0;
    std::vector<MeshTab> tab((size_t)std::max(d->num_meshes, 1));
    for (int m = 0; m < d->num_meshes; m++) {
        const pt_mesh& me = d->meshes[m];
        if (nv + (size_t)me.num_vertices > 0x7fffffffu || nf + (size_t)me.num_faces > 0x7fffffffu)
            return pt_fail(PT_ERR_UNSUPPORTED, "more than 2^31 vertices or faces in one scene");
        tab[m] = MeshTab{(int32_t)nv, (int32_t)nf, me.num_vertices, me.num_faces, me.material_id, me.area_light_id};
        nv += (size_t)me.num_vertices;
        nf += (size_t)me.num_faces;
    }
    TmpBuf<pt_shape> shapes;
    TmpBuf<MeshTab> tab_dev;
    TmpBuf<float> pos, nor;
    TmpBuf<int32_t> idx;
    TmpBuf<unsigned int> flags;
    HIPP(shapes.alloc(N)); HIPP(tab_dev.alloc(tab.size())); HIPP(pos.alloc(nv * 3)); HIPP(nor.alloc(nv * 3)); HIPP(idx.alloc(nf * 3)); HIPP(flags.alloc(2));
    HIPP(hipMemsetAsync(flags.p, 0, 2 * sizeof(unsigned int), nullptr));
    HIPP(hipMemcpy(shapes.p, d->shapes, (size_t)N * sizeof(pt_shape), hipMemcpyHostToDevice));
    HIPP(hipMemcpy(tab_dev.p, tab.data(), tab.size() * sizeof(MeshTab), hipMemcpyHostToDevice));
    if (d->num_meshes <= 64) {
        for (int m = 0; m < d->num_meshes; m++) {
            const pt_mesh& me = d->meshes[m];
            HIPP(hipMemcpy(pos.p + 3 * (size_t)tab[m].vertex_base, me.positions, (size_t)me.num_vertices * 3 * sizeof(float), hipMemcpyHostToDevice));
            HIPP(hipMemcpy(nor.p + 3 * (size_t)tab[m].vertex_base, me.normals, (size_t)me.num_vertices * 3 * sizeof(float), hipMemcpyHostToDevice));
            HIPP(hipMemcpy(idx.p + 3 * (size_t)tab[m].face_base, me.indices, (size_t)me.num_faces * 3 * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    } else {                                               // many small meshes: one staging copy each instead of three transfers per mesh
        std::vector<float> hp(nv * 3), hn(nv * 3);
        std::vector<int32_t> hi(nf * 3);
        for (int m = 0; m < d->num_meshes; m++) {
            const pt_mesh& me = d->meshes[m];
            std::memcpy(hp.data() + 3 * (size_t)tab[m].vertex_base, me.positions, (size_t)me.num_vertices * 3 * sizeof(float));
            std::memcpy(hn.data() + 3 * (size_t)tab[m].vertex_base, me.normals, (size_t)me.num_vertices * 3 * sizeof(float));
            std::memcpy(hi.data() + 3 * (size_t)tab[m].face_base, me.indices, (size_t)me.num_faces * 3 * sizeof(int32_t));
        }
        HIPP(hipMemcpy(pos.p, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPP(hipMemcpy(nor.p, hn.data(), hn.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPP(hipMemcpy(idx.p, hi.data(), hi.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    const int T = 256, G = (N + T - 1) / T;
    hipLaunchKernelGGL(prims_kernel, dim3(G), dim3(T), 0, nullptr, shapes.p, N, tab_dev.p, d->num_meshes, d->num_materials, pos.p, nor.p, idx.p,
                       reinterpret_cast<DPrim*>(prims_dev), reinterpret_cast<DNormals*>(normals_dev), flags.p);
    HIPP(hipGetLastError());
    unsigned int f[2] = {0, 0};
    HIPP(hipMemcpy(f, flags.p, sizeof f, hipMemcpyDeviceToHost));
    if (f[0]) {
        const char* msg = (f[0] & kPErrSphereMat) ? "sphere material id out of range"
                        : (f[0] & kPErrMeshIndex) ? "triangle mesh index out of range"
                        : (f[0] & kPErrFaceIndex) ? "triangle face index out of range"
                        : (f[0] & kPErrVertexIndex) ? "vertex index out of range"
                        : "unknown shape type";
        return pt_fail(PT_ERR_BAD_SCENE, msg);
    }
    *has_sphere = f[1] ? 1 : 0;
    return PT_OK;
}
