// pt_sweep_build.hip — the library's internal tree (pt_tree_sweep.h: pts::build_sweep_tree) built ON THE DEVICE, byte for byte
// the tree the host builder makes.
//
// The host builder (the CPU-side checker of this file; 76 ms for bunny's 288 k primitives and 306 ms for the 1.15 M of the buddha
// stand-in on 8 host threads) is a top-down full-sweep surface-area build: at every node the primitives are swept along x, y
// and z in centroid order and the cut with the smallest  SA(left) n_left + SA(right) n_right  wins, ties by distance from the
// middle, then axis, then position.  The reference's own construct_bvh (bvh.cu:16-54) is host code too (README.md:123,132:
// 10-57 s of start-up); nothing here follows it — this is the producer of the tree the hot path (scene.h:246-301 as restated
// in pt_trace.h) traverses by default.
//
// Level-synchronous formulation, every node of a level at once, everything a streaming pass over n elements:
//   * three orders of the primitives by (centroid, id) — 64-bit radix sorts (rocPRIM), the boxes travel with their order;
//   * every node of the tree is the same range [b, e) of all three orders; a position knows its node by b (its scan key);
//   * per axis: the boxes of all prefixes and of all suffixes of every range = two segmented scans (min / max do not round, the
//     scan's operator keeps the EARLIER operand on ties exactly as the host's sequential merge does, so the very bits — the
//     sign of a zero included — are the host's); cost of the cut after position i in fp64 with the host's operand order;
//   * best cut of every node = a segmented min-scan over (cost, |2k - m|, axis, k), a total order: any association gives the
//     host's choice (read at the node's last position);
//   * the two other orders follow by a stable partition per node = an exclusive segmented scan of the "goes left" flags;
//   * a node's slot in the pre-order output follows from its parent's: left = slot + 1, right = slot + 2 k.
// One level costs ~35 small launches whatever n is; the levels of a million-primitive tree take a few milliseconds together.
// Adversarial input (a branch deeper than 2 log2 n + 16, where the host builder switches to median cuts) is handed back to
// the host builder: PT_ERR_UNSUPPORTED from sweep_build_device, never a different tree.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan_by_key.hpp>
#include <rocprim/iterator/reverse_iterator.hpp>

#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_internal.h"
#include "pt_sweep_build.h"

namespace {

#define HIPS(expr)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return pt_fail(e_ == hipErrorNoDevice ? PT_ERR_NO_DEVICE : PT_ERR_DEVICE,                \
                           std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

struct SB {                     // pts::SBox
    float lo[3], hi[3];
};

// pts::sbox_merge(a, b) with a = the operand that came EARLIER in the sequential sweep: b replaces a only when strictly
// smaller (larger), so equal values — +0 and -0 — keep the earlier one's bits
struct MergeKeepFirst {
    __host__ __device__ SB operator()(const SB& a, const SB& b) const {
        SB r;
        for (int k = 0; k < 3; k++) {
            r.lo[k] = b.lo[k] < a.lo[k] ? b.lo[k] : a.lo[k];
            r.hi[k] = b.hi[k] > a.hi[k] ? b.hi[k] : a.hi[k];
        }
        return r;
    }
};

// the host's sweeps start from sbox_empty() = (+3e38, -3e38): what a sweep holds is min(3e38, ...) / max(-3e38, ...)
__device__ __forceinline__ SB clamp_like_host(const SB& b) {
    SB r;
    for (int k = 0; k < 3; k++) {
        r.lo[k] = b.lo[k] < 3.0e38f ? b.lo[k] : 3.0e38f;
        r.hi[k] = b.hi[k] > -3.0e38f ? b.hi[k] : -3.0e38f;
    }
    return r;
}
__device__ __forceinline__ double area_like_host(const SB& b) {       // pts::sbox_area
    const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
    return 2.0 * (x * y + y * z + z * x);
}

struct Cand {                   // a cut of a node: total order (cost, off, axis, k)
    double cost;
    int32_t off;
    int32_t ak;                 // axis << 28 | k
};
struct CandMin {
    __host__ __device__ Cand operator()(const Cand& a, const Cand& b) const {
        if (b.cost < a.cost) return b;
        if (a.cost < b.cost) return a;
        if (b.off != a.off) return b.off < a.off ? b : a;
        return b.ak < a.ak ? b : a;
    }
};
__device__ __forceinline__ Cand cand_none() { return Cand{__builtin_huge_val(), 0x7fffffff, 0x7fffffff}; }

// ---- the three orders ------------------------------------------------------------------------------------------------------
__global__ void sort_keys_kernel(const SB* __restrict__ boxes, int n, int axis, unsigned long long* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c = (boxes[i].lo[axis] + boxes[i].hi[axis]) * 0.5f;            // pt_tree_sweep.h: cen
    if (c == 0.0f) c = 0.0f;                                             // -0 and +0 compare equal on the host: one key
    uint32_t u = __builtin_bit_cast(uint32_t, c);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                      // unsigned order = float order
    keys[i] = ((unsigned long long)u << 32) | (unsigned long long)(uint32_t)i;      // (centroid, id): a total order
}
// ---- one level ---------------------------------------------------------------------------------------------------------------
// The three orders side by side: position i of the x-, y- and z-order.  All three share their node ranges, so the cut search,
// the flags, their ranks and the partition each take one pass for all three (the box scans stay per order).
struct SB3 { SB* a[3]; };        // the boxes of the three orders: one array per order (24-B elements scan faster than 72-B ones)
struct I3 { int32_t a[3]; };
struct U3 { uint32_t a[3]; };
struct Plus3 {
    __host__ __device__ U3 operator()(const U3& x, const U3& y) const { return U3{{x.a[0] + y.a[0], x.a[1] + y.a[1], x.a[2] + y.a[2]}}; }
};

__global__ void gather_order3_kernel(const unsigned long long* __restrict__ keys, const SB* __restrict__ boxes, int n, int axis,
                                     I3* __restrict__ idx, SB* __restrict__ bx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t p = (int32_t)(uint32_t)keys[i];
    idx[i].a[axis] = p;
    bx[i] = boxes[p];
}
// best cut after position i over the three axes (cand_kernel for all of them); the node's box at its first position
__global__ void cand3_kernel(const SB3 pre, const SB3 suf, const uint32_t* __restrict__ seg_b,
                             const uint32_t* __restrict__ seg_e, int n, Cand* __restrict__ best, SB* __restrict__ whole) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = seg_b[i], e = seg_e[i];
    if ((uint32_t)i == b) whole[b] = clamp_like_host(suf.a[0][i]);
    Cand c = cand_none();
    if ((uint32_t)i + 1 < e) {
        const int k = i - (int)b + 1, m = (int)(e - b);
        const int off = 2 * k > m ? 2 * k - m : m - 2 * k;
        for (int a = 0; a < 3; a++) {
            Cand t;
            t.cost = area_like_host(clamp_like_host(pre.a[a][i])) * k + area_like_host(clamp_like_host(suf.a[a][i + 1])) * (m - k);
            t.off = off;
            t.ak = (a << 28) | k;
            c = a == 0 ? t : CandMin()(c, t);
        }
    }
    best[i] = c;
}
__global__ void mark_left3_kernel(const uint32_t* __restrict__ seg_b, const int32_t* __restrict__ split, const I3* __restrict__ idx, int n,
                                  unsigned char* __restrict__ left) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = seg_b[i];
    const int32_t sp = split[b];
    const int axis = sp >> 28, k = sp & 0x0fffffff;
    if (i - (int)b < k) left[idx[i].a[axis]] = 1;
}
__global__ void flags3_kernel(const I3* __restrict__ idx, const unsigned char* __restrict__ left, int n, U3* __restrict__ f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const I3 v = idx[i];
    f[i] = U3{{left[v.a[0]], left[v.a[1]], left[v.a[2]]}};
}
__global__ void scatter3_kernel(const I3* __restrict__ idx, const SB3 bx, const U3* __restrict__ f, const U3* __restrict__ rank,
                                const uint32_t* __restrict__ seg_b, const uint32_t* __restrict__ seg_e, const int32_t* __restrict__ split, int n,
                                I3* __restrict__ idx_out, const SB3 bx_out, uint32_t* __restrict__ seg_b_out,
                                uint32_t* __restrict__ seg_e_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = seg_b[i], e = seg_e[i];
    const uint32_t k = (uint32_t)(split[b] & 0x0fffffff);
    const U3 fl = f[i], r = rank[i];
    for (int a = 0; a < 3; a++) {
        const uint32_t dest = fl.a[a] ? b + r.a[a] : b + k + ((uint32_t)i - b - r.a[a]);
        idx_out[dest].a[a] = idx[i].a[a];
        bx_out.a[a][dest] = bx.a[a][i];
        if (a == 0) {
            seg_b_out[dest] = fl.a[0] ? b : b + k;
            seg_e_out[dest] = fl.a[0] ? b + k : e;
        }
    }
}
// apply_kernel's view of the x-order
struct BuildCtl {
    unsigned int active;        // nodes of the next level with more than one primitive
    unsigned int too_deep;      // a node beyond the host builder's depth guard: hand the build back
    unsigned int max_child;     // primitives of the largest node of the next level
    int max_depth;              // (kept across levels: the three words before it are cleared per level)
};

// One thread per position; the one that starts a node emits it and splits it.  best_scan = inclusive segmented min-scan of the
// cut candidates: a node's best cut stands at its last position.  The level's control words are summed per block first.
__global__ void apply_kernel(const Cand* __restrict__ best_scan, const uint32_t* __restrict__ seg_b, const uint32_t* __restrict__ seg_e,
                             const SB* __restrict__ whole, const I3* __restrict__ idx3, const SB* __restrict__ bx0, int n,
                             int32_t* __restrict__ slot, int32_t* __restrict__ split, pt_bvh_node* __restrict__ out, int depth,
                             int guard_depth, BuildCtl* __restrict__ ctl) {
    __shared__ unsigned int sh_active, sh_deep, sh_wrote, sh_child;
    if (threadIdx.x == 0) { sh_active = 0; sh_deep = 0; sh_wrote = 0; sh_child = 0; }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && seg_b[i] == (uint32_t)i) {
        const uint32_t b = (uint32_t)i, e = seg_e[i];
        const int m = (int)(e - b);
        const int32_t s = slot[b];
        if (m == 1) {
            split[b] = 0;                               // axis 0, k = 0: the partition below leaves the position where it is
            if (s >= 0) {                               // a leaf not written yet
                pt_bvh_node nd;
                for (int k = 0; k < 3; k++) { nd.bmin[k] = bx0[b].lo[k]; nd.bmax[k] = bx0[b].hi[k]; }
                nd.left = -1; nd.right = -1; nd.prim = idx3[b].a[0];
                out[s] = nd;
                slot[b] = -1;
                sh_wrote = 1u;
            }
        } else if (depth > guard_depth) {
            sh_deep = 1u;
            split[b] = 0;
        } else {
            const Cand c = best_scan[e - 1];
            const int axis = c.ak >> 28, k = c.ak & 0x0fffffff;
            pt_bvh_node nd;
            for (int q = 0; q < 3; q++) { nd.bmin[q] = whole[b].lo[q]; nd.bmax[q] = whole[b].hi[q]; }
            nd.prim = -1;
            nd.left = s + 1;
            nd.right = s + 2 * k;                       // the left subtree holds 2 k - 1 nodes
            out[s] = nd;
            split[b] = (axis << 28) | k;
            slot[b] = s + 1;                            // the children's slots, by their first positions
            slot[b + (uint32_t)k] = s + 2 * k;
            sh_wrote = 1u;
            sh_active = 1u;                             // children exist: at least one more level (fresh leaves are written there)
            atomicMax(&sh_child, (unsigned int)(k > m - k ? k : m - k));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sh_active) atomicAdd(&ctl->active, 1u);
        if (sh_deep) atomicAdd(&ctl->too_deep, 1u);
        if (sh_wrote) atomicMax(&ctl->max_depth, depth);
        if (sh_child) atomicMax(&ctl->max_child, sh_child);
    }
}

// Once every node of a level holds at most kSmall primitives, one thread per node finishes its whole subtree by itself: the
// host builder's own sequential steps (pt_tree_sweep.h: process / run_subtree — the same sweeps, the same comparisons in the same
// order, hence the same bytes) on a private copy of the node's three orders.  The last six to eight levels of a build — as
// many launches-times-whole-array passes as all the levels above them — become one launch.
constexpr int kSmall = 32;      // 16 costs a level more, 64 makes the one-thread subtrees the long pole (buddha stand-in: 30.8 ms at 32, 32.1 at 64)
__device__ __forceinline__ void merge_host(SB& a, const SB& b) {         // pts::sbox_merge
    for (int k = 0; k < 3; k++) {
        a.lo[k] = b.lo[k] < a.lo[k] ? b.lo[k] : a.lo[k];
        a.hi[k] = b.hi[k] > a.hi[k] ? b.hi[k] : a.hi[k];
    }
}
__device__ __forceinline__ SB empty_host() { return SB{{3.0e38f, 3.0e38f, 3.0e38f}, {-3.0e38f, -3.0e38f, -3.0e38f}}; }

__global__ __launch_bounds__(64) void finish_small_kernel(const I3* __restrict__ idx, const SB3 bx, const uint32_t* __restrict__ seg_b,
                                                          const uint32_t* __restrict__ seg_e, int n, const int32_t* __restrict__ slot,
                                                          unsigned char* __restrict__ left, pt_bvh_node* __restrict__ out, int depth0,
                                                          int guard_depth, BuildCtl* __restrict__ ctl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || seg_b[i] != (uint32_t)i) return;
    const int b0 = i, m0 = (int)(seg_e[i] - seg_b[i]);
    const int32_t s0 = slot[b0];
    if (s0 < 0) return;                                 // a leaf written on an earlier level
    int32_t lid[3][kSmall];                             // the node's three orders (primitive ids) and their boxes, private
    SB lbx[3][kSmall];
    int32_t tid[kSmall];
    SB tbx[kSmall];
    double suf[kSmall + 1];
    for (int a = 0; a < 3; a++)
        for (int j = 0; j < m0; j++) { lid[a][j] = idx[b0 + j].a[a]; lbx[a][j] = bx.a[a][b0 + j]; }
    struct Task { int16_t b, e; int32_t slot; int16_t depth; };
    Task todo[kSmall + 2];
    int top = 0, deepest = depth0;
    todo[top++] = Task{0, (int16_t)m0, s0, (int16_t)depth0};
    while (top > 0) {
        const Task t = todo[--top];
        if (t.depth > deepest) deepest = t.depth;
        const int m = t.e - t.b;
        pt_bvh_node nd;
        if (m == 1) {
            for (int q = 0; q < 3; q++) { nd.bmin[q] = lbx[0][t.b].lo[q]; nd.bmax[q] = lbx[0][t.b].hi[q]; }
            nd.left = -1; nd.right = -1; nd.prim = lid[0][t.b];
            out[t.slot] = nd;
            continue;
        }
        if (t.depth > guard_depth) { atomicAdd(&ctl->too_deep, 1u); return; }
        int best_axis = 0, best_k = m / 2, best_off = 0;
        double best_cost = 0.0;
        SB whole = empty_host();
        for (int a = 0; a < 3; a++) {
            SB acc = empty_host();
            for (int j = t.e - 1; j > t.b; j--) {                 // suf[j] = area of the box of [j, e)
                merge_host(acc, lbx[a][j]);
                suf[j] = area_like_host(acc);
            }
            SB w = acc;
            merge_host(w, lbx[a][t.b]);
            if (a == 0) whole = w;
            acc = empty_host();
            bool have = false;
            double a_cost = 0.0;
            int a_off = 0, a_k = 0;
            for (int k = 1; k < m; k++) {                          // cut after the first k of this order
                merge_host(acc, lbx[a][t.b + k - 1]);
                const double cost = area_like_host(acc) * k + suf[t.b + k] * (m - k);
                const int off = 2 * k > m ? 2 * k - m : m - 2 * k;
                if (!have || cost < a_cost || (cost == a_cost && off < a_off)) { have = true; a_cost = cost; a_off = off; a_k = k; }
            }
            if (a == 0 || a_cost < best_cost || (a_cost == best_cost && a_off < best_off)) {
                best_axis = a; best_cost = a_cost; best_off = a_off; best_k = a_k;
            }
        }
        // the other two orders follow: stable partition by membership in the left set
        for (int j = t.b; j < t.b + best_k; j++) left[lid[best_axis][j]] = 1;
        for (int o = 1; o <= 2; o++) {
            const int a = (best_axis + o) % 3;
            int l = t.b, r = 0;
            for (int j = t.b; j < t.e; j++) {
                const int32_t p = lid[a][j];
                if (left[p]) { lid[a][l] = p; lbx[a][l] = lbx[a][j]; l++; } else { tid[r] = p; tbx[r] = lbx[a][j]; r++; }
            }
            for (int j = 0; j < r; j++) { lid[a][l + j] = tid[j]; lbx[a][l + j] = tbx[j]; }
        }
        for (int j = t.b; j < t.b + best_k; j++) left[lid[best_axis][j]] = 0;
        for (int q = 0; q < 3; q++) { nd.bmin[q] = whole.lo[q]; nd.bmax[q] = whole.hi[q]; }
        nd.prim = -1;
        nd.left = t.slot + 1;
        nd.right = t.slot + 2 * best_k;
        out[t.slot] = nd;
        todo[top++] = Task{(int16_t)(t.b + best_k), t.e, nd.right, (int16_t)(t.depth + 1)};
        todo[top++] = Task{t.b, (int16_t)(t.b + best_k), nd.left, (int16_t)(t.depth + 1)};
    }
    atomicMax(&ctl->max_depth, deepest);
}

__global__ void init_segments_kernel(int n, uint32_t* __restrict__ seg_b, uint32_t* __restrict__ seg_e, int32_t* __restrict__ slot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    seg_b[i] = 0u;
    seg_e[i] = (uint32_t)n;
    slot[i] = i == 0 ? 0 : -1;
}

template <class T>
struct Buf {
    T* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T)); }
};

}  // namespace

int pts::sweep_build_device(const float* leaf_boxes, int n, std::vector<pt_bvh_node>& out, int32_t* out_root, int32_t* out_depth,
                            double* out_device_ms) {
    if (!leaf_boxes || n <= 0 || n >= (1 << 28)) return pt_fail(PT_ERR_INVALID_ARG, "sweep_build_device: bad argument");
    Buf<SB> boxes;
    Buf<pt_bvh_node> nodes;
    HIPS(boxes.alloc(n));
    HIPS(nodes.alloc((size_t)2 * n - 1));
    HIPS(hipMemcpy(boxes.p, leaf_boxes, (size_t)n * sizeof(SB), hipMemcpyHostToDevice));
    int rc = sweep_build_on_device(reinterpret_cast<const float*>(boxes.p), n, nodes.p, out_depth, out_device_ms);
    if (rc) return rc;
    out.assign((size_t)2 * n - 1, pt_bvh_node{});
    HIPS(hipMemcpy(out.data(), nodes.p, out.size() * sizeof(pt_bvh_node), hipMemcpyDeviceToHost));
    *out_root = 0;
    return PT_OK;
}

int pts::sweep_build_on_device(const float* leaf_boxes_dev, int n, pt_bvh_node* nodes_dev, int32_t* out_depth, double* out_device_ms) {
    if (!leaf_boxes_dev || !nodes_dev || n <= 0 || n >= (1 << 28)) return pt_fail(PT_ERR_INVALID_ARG, "sweep_build_on_device: bad argument");
    struct { const SB* p; } boxes{reinterpret_cast<const SB*>(leaf_boxes_dev)};
    struct { pt_bvh_node* p; } nodes{nodes_dev};
    const int T = 256, G = (n + T - 1) / T;
    int lg = 0;
    while ((1 << lg) < n) lg++;
    const int guard_depth = 2 * lg + 16;                     // pt_tree_sweep.h: beyond this the host builder cuts at the median

    // One allocation for all working arrays (some thirty of them: a hipMalloc each would cost more than several levels of the build)
    struct Arena {
        unsigned char* base = nullptr;
        size_t used = 0;
        ~Arena() { if (base) (void)hipFree(base); }
        size_t reserve(size_t bytes) { const size_t at = used; used += (bytes + 255) & ~(size_t)255; return at; }
    } arena;
    SB3 bx{}, bx_alt{}, pre{}, suf{};
    struct { SB* p; } whole;
    struct { I3* p; } idx, idx_alt;
    struct { int32_t* p; } slot, split;
    struct { unsigned long long* p; } keys, keys_alt;
    struct { uint32_t* p; } seg_b, seg_e, seg_b_alt, seg_e_alt;
    struct { U3* p; } flags, rank;
    struct { Cand* p; } best, best_seg;
    struct { unsigned char* p; } left, temp;
    struct { BuildCtl* p; } ctl;
    // temporary storage of the rocPRIM calls: the largest of the five kinds, sized once (size queries read no memory)
    size_t t_sort = 0, t_scan_box = 0, t_scan_box_r = 0, t_scan_u = 0, t_reduce = 0;
    {
        unsigned long long* k64 = nullptr; uint32_t* k32 = nullptr; SB* sb = nullptr; Cand* cd = nullptr; U3* u3 = nullptr;
        HIPS(rocprim::radix_sort_keys(nullptr, t_sort, k64, k64, (size_t)n, 0, 64, nullptr));
        HIPS(rocprim::inclusive_scan_by_key(nullptr, t_scan_box, k32, sb, sb, (size_t)n, MergeKeepFirst(), rocprim::equal_to<uint32_t>(), nullptr));
        auto kr = rocprim::make_reverse_iterator(k32 + n);
        auto vr = rocprim::make_reverse_iterator(sb + n);
        HIPS(rocprim::inclusive_scan_by_key(nullptr, t_scan_box_r, kr, vr, vr, (size_t)n, MergeKeepFirst(), rocprim::equal_to<uint32_t>(), nullptr));
        HIPS(rocprim::exclusive_scan_by_key(nullptr, t_scan_u, k32, u3, u3, U3{{0u, 0u, 0u}}, (size_t)n, Plus3(), rocprim::equal_to<uint32_t>(), nullptr));
        HIPS(rocprim::inclusive_scan_by_key(nullptr, t_reduce, k32, cd, cd, (size_t)n, CandMin(), rocprim::equal_to<uint32_t>(), nullptr));
    }
    const size_t t_bytes = std::max(std::max(std::max(t_sort, t_scan_box), std::max(t_scan_box_r, t_scan_u)), t_reduce);
    const size_t nn = (size_t)n;
    size_t o_bx[3], o_bxa[3], o_pre[3], o_suf[3];
    for (int a = 0; a < 3; a++) { o_bx[a] = arena.reserve(nn * sizeof(SB)); o_bxa[a] = arena.reserve(nn * sizeof(SB)); o_pre[a] = arena.reserve(nn * sizeof(SB)); o_suf[a] = arena.reserve(nn * sizeof(SB)); }
    const size_t o_idx = arena.reserve(nn * sizeof(I3)), o_idxa = arena.reserve(nn * sizeof(I3)), o_whole = arena.reserve(nn * sizeof(SB));
    const size_t o_slot = arena.reserve(nn * 4), o_split = arena.reserve(nn * 4), o_keys = arena.reserve(nn * 8), o_keysa = arena.reserve(nn * 8);
    const size_t o_sb = arena.reserve(nn * 4), o_se = arena.reserve(nn * 4), o_sba = arena.reserve(nn * 4), o_sea = arena.reserve(nn * 4);
    const size_t o_flags = arena.reserve(nn * sizeof(U3)), o_rank = arena.reserve(nn * sizeof(U3));
    const size_t o_best = arena.reserve(nn * sizeof(Cand)), o_bests = arena.reserve(nn * sizeof(Cand)), o_left = arena.reserve(nn);
    const size_t o_ctl = arena.reserve(sizeof(BuildCtl)), o_temp = arena.reserve(t_bytes);
    HIPS(hipMalloc(reinterpret_cast<void**>(&arena.base), arena.used));
    for (int a = 0; a < 3; a++) {
        bx.a[a] = reinterpret_cast<SB*>(arena.base + o_bx[a]); bx_alt.a[a] = reinterpret_cast<SB*>(arena.base + o_bxa[a]);
        pre.a[a] = reinterpret_cast<SB*>(arena.base + o_pre[a]); suf.a[a] = reinterpret_cast<SB*>(arena.base + o_suf[a]);
    }
    idx.p = reinterpret_cast<I3*>(arena.base + o_idx); idx_alt.p = reinterpret_cast<I3*>(arena.base + o_idxa);
    whole.p = reinterpret_cast<SB*>(arena.base + o_whole);
    slot.p = reinterpret_cast<int32_t*>(arena.base + o_slot); split.p = reinterpret_cast<int32_t*>(arena.base + o_split);
    keys.p = reinterpret_cast<unsigned long long*>(arena.base + o_keys); keys_alt.p = reinterpret_cast<unsigned long long*>(arena.base + o_keysa);
    seg_b.p = reinterpret_cast<uint32_t*>(arena.base + o_sb); seg_e.p = reinterpret_cast<uint32_t*>(arena.base + o_se);
    seg_b_alt.p = reinterpret_cast<uint32_t*>(arena.base + o_sba); seg_e_alt.p = reinterpret_cast<uint32_t*>(arena.base + o_sea);
    flags.p = reinterpret_cast<U3*>(arena.base + o_flags); rank.p = reinterpret_cast<U3*>(arena.base + o_rank);
    best.p = reinterpret_cast<Cand*>(arena.base + o_best); best_seg.p = reinterpret_cast<Cand*>(arena.base + o_bests);
    left.p = arena.base + o_left; temp.p = arena.base + o_temp;
    ctl.p = reinterpret_cast<BuildCtl*>(arena.base + o_ctl);

    hipEvent_t ev0, ev1;
    HIPS(hipEventCreate(&ev0));
    if (hipEventCreate(&ev1) != hipSuccess) { (void)hipEventDestroy(ev0); return pt_fail(PT_ERR_DEVICE, "hipEventCreate"); }
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{ev0, ev1};

    HIPS(hipEventRecord(ev0, nullptr));
    for (int a = 0; a < 3; a++) {
        hipLaunchKernelGGL(sort_keys_kernel, dim3(G), dim3(T), 0, nullptr, boxes.p, n, a, keys.p);
        size_t tb = t_bytes;
        HIPS(rocprim::radix_sort_keys(temp.p, tb, keys.p, keys_alt.p, (size_t)n, 0, 64, nullptr));
        hipLaunchKernelGGL(gather_order3_kernel, dim3(G), dim3(T), 0, nullptr, keys_alt.p, boxes.p, n, a, idx.p, bx.a[a]);
    }
    hipLaunchKernelGGL(init_segments_kernel, dim3(G), dim3(T), 0, nullptr, n, seg_b.p, seg_e.p, slot.p);
    HIPS(hipMemsetAsync(left.p, 0, (size_t)n, nullptr));
    HIPS(hipMemsetAsync(ctl.p, 0, sizeof(BuildCtl), nullptr));
    HIPS(hipGetLastError());

    SB3 cur_bx = bx, alt_bx = bx_alt;
    I3 *cur_idx = idx.p, *alt_idx = idx_alt.p;
    uint32_t *cur_b = seg_b.p, *cur_e = seg_e.p, *alt_b = seg_b_alt.p, *alt_e = seg_e_alt.p;
    BuildCtl h{};
    int depth = 1;
    for (;; depth++) {
        if (depth > guard_depth + 2) return pt_fail(PT_ERR_UNSUPPORTED, "sweep_build_device: deeper than the host builder's guard");
        size_t tb = t_bytes;
        for (int a = 0; a < 3; a++) {
            tb = t_bytes;
            HIPS(rocprim::inclusive_scan_by_key(temp.p, tb, cur_b, cur_bx.a[a], pre.a[a], (size_t)n, MergeKeepFirst(), rocprim::equal_to<uint32_t>(), nullptr));
            auto kr = rocprim::make_reverse_iterator(cur_b + n);
            auto vr = rocprim::make_reverse_iterator(cur_bx.a[a] + n);
            auto orr = rocprim::make_reverse_iterator(suf.a[a] + n);
            tb = t_bytes;
            HIPS(rocprim::inclusive_scan_by_key(temp.p, tb, kr, vr, orr, (size_t)n, MergeKeepFirst(), rocprim::equal_to<uint32_t>(), nullptr));
        }
        hipLaunchKernelGGL(cand3_kernel, dim3(G), dim3(T), 0, nullptr, pre, suf, cur_b, cur_e, n, best.p, whole.p);
        tb = t_bytes;
        HIPS(rocprim::inclusive_scan_by_key(temp.p, tb, cur_b, best.p, best_seg.p, (size_t)n, CandMin(), rocprim::equal_to<uint32_t>(), nullptr));
        HIPS(hipMemsetAsync(ctl.p, 0, sizeof(unsigned int) * 3, nullptr));           // active, too_deep, max_child (max_depth stays)
        hipLaunchKernelGGL(apply_kernel, dim3(G), dim3(T), 0, nullptr, best_seg.p, cur_b, cur_e, whole.p, cur_idx, cur_bx.a[0], n,
                           slot.p, split.p, nodes.p, depth, guard_depth, ctl.p);
        HIPS(hipGetLastError());
        HIPS(hipMemcpy(&h, ctl.p, sizeof h, hipMemcpyDeviceToHost));
        if (h.too_deep) return pt_fail(PT_ERR_UNSUPPORTED, "sweep_build_device: a branch beyond the depth guard (host builder takes over)");
        if (h.active == 0) break;
        hipLaunchKernelGGL(mark_left3_kernel, dim3(G), dim3(T), 0, nullptr, cur_b, split.p, cur_idx, n, left.p);
        hipLaunchKernelGGL(flags3_kernel, dim3(G), dim3(T), 0, nullptr, cur_idx, left.p, n, flags.p);
        tb = t_bytes;
        HIPS(rocprim::exclusive_scan_by_key(temp.p, tb, cur_b, flags.p, rank.p, U3{{0u, 0u, 0u}}, (size_t)n, Plus3(), rocprim::equal_to<uint32_t>(), nullptr));
        hipLaunchKernelGGL(scatter3_kernel, dim3(G), dim3(T), 0, nullptr, cur_idx, cur_bx, flags.p, rank.p, cur_b, cur_e, split.p, n,
                           alt_idx, alt_bx, alt_b, alt_e);
        HIPS(hipMemsetAsync(left.p, 0, (size_t)n, nullptr));
        HIPS(hipGetLastError());
        std::swap(cur_bx, alt_bx);
        std::swap(cur_idx, alt_idx);
        std::swap(cur_b, alt_b);
        std::swap(cur_e, alt_e);
        if (h.max_child <= (unsigned int)kSmall) {
            // every node of the next level is small: one thread each finishes its subtree (finish_small_kernel)
            hipLaunchKernelGGL(finish_small_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, cur_idx, cur_bx, cur_b, cur_e, n, slot.p, left.p,
                               nodes.p, depth + 1, guard_depth, ctl.p);
            HIPS(hipGetLastError());
            HIPS(hipMemcpy(&h, ctl.p, sizeof h, hipMemcpyDeviceToHost));
            if (h.too_deep) return pt_fail(PT_ERR_UNSUPPORTED, "sweep_build_device: a branch beyond the depth guard (host builder takes over)");
            break;
        }
    }
    HIPS(hipEventRecord(ev1, nullptr));
    HIPS(hipEventSynchronize(ev1));
    float ms = 0;
    HIPS(hipEventElapsedTime(&ms, ev0, ev1));
    if (out_device_ms) *out_device_ms = ms;
    if (out_depth) *out_depth = h.max_depth;
    return PT_OK;
}
