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
#include <hipcub/block/block_scan.hpp>

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
    // field-wise selects on one predicate: returning one of the two operands whole makes the compiler select between their
    // ADDRESSES, which puts arrays of candidates into scratch memory (tile_cand_kernel: 224 B per lane)
    __host__ __device__ Cand operator()(const Cand& a, const Cand& b) const {
        const bool take_b = b.cost < a.cost ? true : a.cost < b.cost ? false : b.off != a.off ? b.off < a.off : b.ak < a.ak;
        Cand r;
        r.cost = take_b ? b.cost : a.cost;
        r.off = take_b ? b.off : a.off;
        r.ak = take_b ? b.ak : a.ak;
        return r;
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
// ---- the box scans and the cut search of a level in three launches ------------------------------------------------------------
// The six segmented scans above (rocPRIM, one launch or two each, 52 B of traffic per element and scan) and cand3_kernel read and
// write 480 B per element and level; the cut search needs none of the scanned boxes in memory — only, per position, the best of
// three costs.  Tiles of 1,024 positions: (A) every tile's aggregate per order and direction, (B) the carries into the tiles (a
// segmented scan over the aggregates, six small blocks), (C) per tile and order both scans again with their carries, the suffix
// boxes through LDS, the prefix boxes in registers, and the best cut after every position straight out: 168 B per element.
// The operator is the segmented form of MergeKeepFirst — (earlier, later) in scan order, so the bits are the sequential sweep's.
#ifndef PT_SWEEP_ITEMS
#define PT_SWEEP_ITEMS 4
#endif
constexpr int kScanT = 256, kScanI = PT_SWEEP_ITEMS, kTile = kScanT * kScanI;
struct FSB {
    SB v;
    uint32_t head;              // 1: a segment starts here (in scan direction); 2: nothing (identity)
};
struct SegMerge {             // (field-wise selects, no whole-operand returns: see CandMin)
    __device__ FSB operator()(const FSB& a, const FSB& b) const {
        const bool keep_a = b.head == 2u;
        const bool take_b = !keep_a && (b.head == 1u || a.head == 2u);
        const SB m = MergeKeepFirst()(a.v, b.v);
        FSB r;
        for (int k = 0; k < 3; k++) {
            r.v.lo[k] = keep_a ? a.v.lo[k] : take_b ? b.v.lo[k] : m.lo[k];
            r.v.hi[k] = keep_a ? a.v.hi[k] : take_b ? b.v.hi[k] : m.hi[k];
        }
        r.head = take_b ? b.head : a.head;
        return r;
    }
};
__device__ __forceinline__ FSB fsb_none() {
    FSB r;
    for (int k = 0; k < 3; k++) { r.v.lo[k] = 0.0f; r.v.hi[k] = 0.0f; }
    r.head = 2u;
    return r;
}
// the same for the cut candidates: a node's best cut = the segmented minimum over its positions.  Tiles write the running minimum
// at every node's LAST position and their aggregate; a node that began in an earlier tile adds the carry into its last tile
// (apply_kernel).  This replaces a scan-by-key over all positions (rocPRIM: 107 us a level for 1.15 M) and the 16-B-per-position
// array it read.
struct FC {
    Cand c;
    uint32_t head;
};
struct SegMin {
    __device__ FC operator()(const FC& a, const FC& b) const {
        const bool keep_a = b.head == 2u;
        const bool take_b = !keep_a && (b.head == 1u || a.head == 2u);
        const Cand m = CandMin()(a.c, b.c);
        FC r;
        r.c.cost = keep_a ? a.c.cost : take_b ? b.c.cost : m.cost;
        r.c.off = keep_a ? a.c.off : take_b ? b.c.off : m.off;
        r.c.ak = keep_a ? a.c.ak : take_b ? b.c.ak : m.ak;
        r.head = take_b ? b.head : a.head;
        return r;
    }
};
__device__ __forceinline__ FC fc_none() { return FC{cand_none(), 2u}; }
__device__ __forceinline__ void set_none(FSB& x) { x = fsb_none(); }
__device__ __forceinline__ void set_none(FC& x) { x = fc_none(); }
// Exclusive SegMin scan of one FC per thread through LDS (sh: 2 x NT entries); `total` = the fold of all of them.
// (hipcub::BlockScan<FC, ...> sends this compiler's MachineCopyPropagation pass into a segmentation fault: ROCm 7.2.0, clang 22.)
template <int NT>
__device__ __forceinline__ FC block_exclusive_min(FC local, FC* sh, FC& total) {
    const int t = threadIdx.x;
    int cur = 0;
    sh[t] = local;
    __syncthreads();
    for (int d = 1; d < NT; d <<= 1) {
        FC v = sh[cur * NT + t];
        if (t >= d) v = SegMin()(sh[cur * NT + t - d], v);
        sh[(cur ^ 1) * NT + t] = v;
        __syncthreads();
        cur ^= 1;
    }
    total = sh[cur * NT + NT - 1];
    return t ? sh[cur * NT + t - 1] : fc_none();
}
// position of item j of thread t: forward tiles run left to right, reverse tiles right to left
__device__ __forceinline__ int tile_pos(int base, int t, int j, bool rev) {
    const int r = t * kScanI + j;
    return rev ? base + kTile - 1 - r : base + r;
}
__device__ __forceinline__ FSB tile_item(const SB* __restrict__ bx, const uint32_t* __restrict__ seg_b, const uint32_t* __restrict__ seg_e,
                                         int n, int i, bool rev) {
    if (i >= n) return fsb_none();
    FSB r;
    r.v = bx[i];
    r.head = rev ? (seg_e[i] == (uint32_t)i + 1u ? 1u : 0u) : (seg_b[i] == (uint32_t)i ? 1u : 0u);
    return r;
}
// (A) agg[(order * 2 + direction) * tiles + tile]
__global__ __launch_bounds__(kScanT) void tile_agg_kernel(const SB3 bx, const uint32_t* __restrict__ seg_b, const uint32_t* __restrict__ seg_e,
                                                          int n, int tiles, FSB* __restrict__ agg) {
    using Scan = hipcub::BlockScan<FSB, kScanT>;
    __shared__ typename Scan::TempStorage tmp;
    const int tile = blockIdx.x, a = blockIdx.y, base = tile * kTile;
    for (int dir = 0; dir < 2; dir++) {
        FSB local = fsb_none();
        for (int j = 0; j < kScanI; j++)
            local = SegMerge()(local, tile_item(bx.a[a], seg_b, seg_e, n, tile_pos(base, threadIdx.x, j, dir == 1), dir == 1));
        FSB scanned, total;
        Scan(tmp).InclusiveScan(local, scanned, SegMerge(), total);
        if (threadIdx.x == 0) agg[(size_t)(a * 2 + dir) * tiles + tile] = total;
        __syncthreads();
    }
}
// (B) carry into tile t = everything before it in scan direction, folded: one block per order and direction
template <class E, class Op>
__global__ __launch_bounds__(1024) void tile_carry_kernel(const E* __restrict__ agg, int tiles, E* __restrict__ carry) {
    using Scan = hipcub::BlockScan<E, 1024>;
    __shared__ typename Scan::TempStorage tmp;
    const bool rev = (blockIdx.x & 1) != 0;
    const E* in = agg + (size_t)blockIdx.x * tiles;
    E* out = carry + (size_t)blockIdx.x * tiles;
    const int per = (tiles + 1023) / 1024;
    const int lo = min((int)threadIdx.x * per, tiles), hi = min(lo + per, tiles);
    E none;
    set_none(none);
    E local = none;
    for (int p = lo; p < hi; p++) local = Op()(local, in[rev ? tiles - 1 - p : p]);
    E run;
    Scan(tmp).ExclusiveScan(local, run, none, Op());
    for (int p = lo; p < hi; p++) {
        const int t = rev ? tiles - 1 - p : p;
        out[t] = run;
        run = Op()(run, in[t]);
    }
}
// the carries of the cut candidates: one block, left to right
constexpr int kCarryT = 256;
__global__ __launch_bounds__(kCarryT) void cand_carry_kernel(const FC* __restrict__ in, int tiles, FC* __restrict__ out) {
    __shared__ FC sh[2 * kCarryT];
    const int per = (tiles + kCarryT - 1) / kCarryT;
    const int lo = min((int)threadIdx.x * per, tiles), hi = min(lo + per, tiles);
    FC local = fc_none();
    for (int p = lo; p < hi; p++) local = SegMin()(local, in[p]);
    FC total;
    FC run = block_exclusive_min<kCarryT>(local, sh, total);
    for (int p = lo; p < hi; p++) {
        out[p] = run;
        run = SegMin()(run, in[p]);
    }
}
// inclusive SegMerge scan of a tile, kScanI consecutive items per thread: the thread's own items in registers, one block scan over
// the threads' totals (hipcub's array form keeps its items in scratch memory here: 224 B per lane)
template <class ScanT>
__device__ __forceinline__ void tile_scan(FSB (&it)[kScanI], typename ScanT::TempStorage& tmp) {
#pragma unroll
    for (int j = 1; j < kScanI; j++) it[j] = SegMerge()(it[j - 1], it[j]);
    FSB before;
    ScanT(tmp).ExclusiveScan(it[kScanI - 1], before, fsb_none(), SegMerge());
#pragma unroll
    for (int j = 0; j < kScanI; j++) it[j] = SegMerge()(before, it[j]);
}
// (C) best cut after every position (cand3_kernel's arithmetic, operand for operand) and the node's box at its first position
__global__ __launch_bounds__(kScanT) void tile_cand_kernel(const SB3 bx, const uint32_t* __restrict__ seg_b, const uint32_t* __restrict__ seg_e,
                                                           int n, int tiles, const FSB* __restrict__ carry, Cand* __restrict__ best_at_end,
                                                           FC* __restrict__ best_agg, SB* __restrict__ whole) {
    using Scan = hipcub::BlockScan<FSB, kScanT>;
    __shared__ typename Scan::TempStorage tmp;
    __shared__ SB suf_sh[kTile + 1];
    static_assert(sizeof(SB) * (kTile + 1) >= sizeof(FC) * 2 * kScanT, "the candidate scan reuses the suffix boxes' LDS");
    const int tile = blockIdx.x, base = tile * kTile, t = threadIdx.x;
    uint32_t sb[kScanI], se[kScanI];
    Cand c[kScanI];
#pragma unroll
    for (int j = 0; j < kScanI; j++) {
        const int i = tile_pos(base, t, j, false);
        sb[j] = i < n ? seg_b[i] : 0u;
        se[j] = i < n ? seg_e[i] : 0u;
        c[j] = cand_none();
    }
    FSB it[kScanI];
    for (int a = 0; a < 3; a++) {
        // suffix boxes: right to left, into LDS by position; suf_sh[kTile] = the run that starts right of the tile
#pragma unroll
        for (int j = 0; j < kScanI; j++) it[j] = tile_item(bx.a[a], seg_b, seg_e, n, tile_pos(base, t, j, true), true);
        tile_scan<Scan>(it, tmp);
        const FSB cr = carry[(size_t)(a * 2 + 1) * tiles + tile];
#pragma unroll
        for (int j = 0; j < kScanI; j++) {
            const int i = tile_pos(base, t, j, true);
            if (i < n) suf_sh[i - base] = it[j].head ? it[j].v : MergeKeepFirst()(cr.v, it[j].v);   // no segment end in [i, tile end): it runs on
        }
        if (t == 0) suf_sh[kTile] = cr.v;
        __syncthreads();
        // prefix boxes: left to right, in registers
#pragma unroll
        for (int j = 0; j < kScanI; j++) it[j] = tile_item(bx.a[a], seg_b, seg_e, n, tile_pos(base, t, j, false), false);
        tile_scan<Scan>(it, tmp);
        const FSB cf = carry[(size_t)(a * 2 + 0) * tiles + tile];
#pragma unroll
        for (int j = 0; j < kScanI; j++) {
            const int i = tile_pos(base, t, j, false);
            if (i >= n) continue;
            const uint32_t b = sb[j], e = se[j];
            if (a == 0 && (uint32_t)i == b) whole[b] = clamp_like_host(suf_sh[i - base]);
            if ((uint32_t)i + 1 < e) {
                const SB pre = it[j].head ? it[j].v : MergeKeepFirst()(cf.v, it[j].v);
                const int k = i - (int)b + 1, m = (int)(e - b);
                const int off = 2 * k > m ? 2 * k - m : m - 2 * k;
                Cand x;
                x.cost = area_like_host(clamp_like_host(pre)) * k + area_like_host(clamp_like_host(suf_sh[i + 1 - base])) * (m - k);
                x.off = off;
                x.ak = (a << 28) | k;
                c[j] = a == 0 ? x : CandMin()(c[j], x);
            }
        }
        __syncthreads();
    }
    // the running minimum of every node up to each position; kept where a node ends, and the tile's aggregate for the carries
    FC fc[kScanI];
#pragma unroll
    for (int j = 0; j < kScanI; j++) {
        const int i = tile_pos(base, t, j, false);
        fc[j] = i < n ? FC{c[j], (uint32_t)i == sb[j] ? 1u : 0u} : fc_none();
    }
    FC local = fc_none();
#pragma unroll
    for (int j = 0; j < kScanI; j++) local = SegMin()(local, fc[j]);
    FC total;
    FC run = block_exclusive_min<kScanT>(local, reinterpret_cast<FC*>(suf_sh), total);     // (the axis loop ended with a barrier)
#pragma unroll
    for (int j = 0; j < kScanI; j++) {
        const int i = tile_pos(base, t, j, false);
        run = SegMin()(run, fc[j]);
        if (i < n && (uint32_t)i + 1 == se[j]) best_at_end[i] = run.c;
    }
    if (t == 0) best_agg[tile] = total;
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
__global__ void apply_kernel(const Cand* __restrict__ best_scan, const FC* __restrict__ best_carry, const uint32_t* __restrict__ seg_b, const uint32_t* __restrict__ seg_e,
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
            Cand c = best_scan[e - 1];
            if (best_carry) {                           // tiled search: best_scan[e - 1] covers the node's part of its last tile
                const uint32_t last_tile = (e - 1) / (uint32_t)kTile;
                if (b < last_tile * (uint32_t)kTile) c = CandMin()(best_carry[last_tile].c, c);
            }
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
#ifndef PT_SWEEP_SMALL
#define PT_SWEEP_SMALL 16
#endif
constexpr int kSmall = PT_SWEEP_SMALL;      // 64 makes the one-thread subtrees the long pole; 32 was the optimum while a level cost ~0.9 ms, with the tiled scans (~0.55 ms a level) it is 16: builder 12.8 -> 12.0 ms for 1.15 M triangles
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
    // PT_SWEEP_SCANS=rocprim: the box scans as six rocPRIM scans-by-key + cand3_kernel (A/B timing; same tree)
    const char* scans_env = std::getenv("PT_SWEEP_SCANS");
    const bool tiled = !(scans_env && std::string(scans_env) == "rocprim");
    const int tiles = (n + kTile - 1) / kTile;
    for (int a = 0; a < 3; a++) {
        o_bx[a] = arena.reserve(nn * sizeof(SB)); o_bxa[a] = arena.reserve(nn * sizeof(SB));
        o_pre[a] = tiled ? 0 : arena.reserve(nn * sizeof(SB)); o_suf[a] = tiled ? 0 : arena.reserve(nn * sizeof(SB));
    }
    const size_t o_agg = arena.reserve((size_t)tiles * 6 * sizeof(FSB)), o_carry = arena.reserve((size_t)tiles * 6 * sizeof(FSB));
    const size_t o_aggc = arena.reserve((size_t)tiles * sizeof(FC)), o_carryc = arena.reserve((size_t)tiles * sizeof(FC));
    const size_t o_idx = arena.reserve(nn * sizeof(I3)), o_idxa = arena.reserve(nn * sizeof(I3)), o_whole = arena.reserve(nn * sizeof(SB));
    const size_t o_slot = arena.reserve(nn * 4), o_split = arena.reserve(nn * 4), o_keys = arena.reserve(nn * 8), o_keysa = arena.reserve(nn * 8);
    const size_t o_sb = arena.reserve(nn * 4), o_se = arena.reserve(nn * 4), o_sba = arena.reserve(nn * 4), o_sea = arena.reserve(nn * 4);
    const size_t o_flags = arena.reserve(nn * sizeof(U3)), o_rank = arena.reserve(nn * sizeof(U3));
    const size_t o_best = tiled ? 0 : arena.reserve(nn * sizeof(Cand)), o_bests = arena.reserve(nn * sizeof(Cand)), o_left = arena.reserve(nn);
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
    FSB* const agg = reinterpret_cast<FSB*>(arena.base + o_agg);
    FSB* const carry = reinterpret_cast<FSB*>(arena.base + o_carry);
    FC* const agg_c = reinterpret_cast<FC*>(arena.base + o_aggc);
    FC* const carry_c = reinterpret_cast<FC*>(arena.base + o_carryc);

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
        if (tiled) {
            hipLaunchKernelGGL(tile_agg_kernel, dim3(tiles, 3), dim3(kScanT), 0, nullptr, cur_bx, cur_b, cur_e, n, tiles, agg);
            hipLaunchKernelGGL((tile_carry_kernel<FSB, SegMerge>), dim3(6), dim3(1024), 0, nullptr, agg, tiles, carry);
            hipLaunchKernelGGL(tile_cand_kernel, dim3(tiles), dim3(kScanT), 0, nullptr, cur_bx, cur_b, cur_e, n, tiles, carry, best_seg.p, agg_c, whole.p);
            hipLaunchKernelGGL(cand_carry_kernel, dim3(1), dim3(kCarryT), 0, nullptr, agg_c, tiles, carry_c);
        } else {
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
        }
        HIPS(hipMemsetAsync(ctl.p, 0, sizeof(unsigned int) * 3, nullptr));           // active, too_deep, max_child (max_depth stays)
        hipLaunchKernelGGL(apply_kernel, dim3(G), dim3(T), 0, nullptr, best_seg.p, tiled ? carry_c : nullptr, cur_b, cur_e, whole.p, cur_idx, cur_bx.a[0], n,
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
