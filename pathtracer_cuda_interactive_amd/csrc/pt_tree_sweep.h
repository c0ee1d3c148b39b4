// Host side, scene creation only: the library's internal tree over the caller's LEAF boxes (pt_api.hip: validate_and_build
// explains why any tree over the same leaf boxes gives the reference's result).
//
// Exact traversal never prunes (scene.h:258-297), so a ray pays for every inner box it touches and the tree is the whole
// cost of a segment.  This is the classic top-down build that looks at ALL cuts: at every node the primitives are swept
// along x, y and z in centroid order and the cut with the smallest  SA(left) * n_left + SA(right) * n_right  wins (ties:
// the cut nearest the middle, then the lower axis, then the lower position — deterministic).  The three orders are sorted
// once and kept sorted by stable partitions, O(n log n) box merges for a balanced tree; a depth guard switches a
// pathological branch to median cuts so that the build stays near that bound.
// Measured on bunny (288,094 primitives, inner visits per segment, exact traversal): caller's median-split tree 52.0, cuts
// of the Morton order (pt_bvh_build.hip, PT_BVH_DEVICE_SAH) 23.2, this 18.0.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "pt_api.h"

namespace pts {

struct SBox {
    float lo[3], hi[3];
};

inline SBox sbox_empty() { return SBox{{3.0e38f, 3.0e38f, 3.0e38f}, {-3.0e38f, -3.0e38f, -3.0e38f}}; }
inline void sbox_merge(SBox& a, const float* b) {
    for (int k = 0; k < 3; k++) {
        a.lo[k] = b[k] < a.lo[k] ? b[k] : a.lo[k];
        a.hi[k] = b[3 + k] > a.hi[k] ? b[3 + k] : a.hi[k];
    }
}
inline double sbox_area(const SBox& b) {
    const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
    return 2.0 * (x * y + y * z + z * x);
}

// leaf_boxes: n x {lo.xyz, hi.xyz}, all finite.  out: the reference's node pool layout (bvh.cuh:7-15), 2n-1 nodes in
// pre-order (node, left subtree, right subtree), root = 0.  Inner boxes are exact unions (min / max do not round).
inline void build_sweep_tree(const float* leaf_boxes, int n, std::vector<pt_bvh_node>& out, int32_t* out_root, int32_t* out_depth) {
    out.assign((size_t)2 * n - 1, pt_bvh_node{});
    std::vector<float> cen((size_t)n * 3);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) cen[(size_t)i * 3 + k] = (leaf_boxes[(size_t)i * 6 + k] + leaf_boxes[(size_t)i * 6 + 3 + k]) * 0.5f;
    std::vector<int32_t> idx[3];
    for (int a = 0; a < 3; a++) {
        idx[a].resize(n);
        for (int i = 0; i < n; i++) idx[a][i] = i;
        std::sort(idx[a].begin(), idx[a].end(), [&](int32_t p, int32_t q) {
            const float cp = cen[(size_t)p * 3 + a], cq = cen[(size_t)q * 3 + a];
            return cp < cq || (cp == cq && p < q);
        });
    }
    std::vector<int32_t> tmp(n);
    std::vector<unsigned char> left_side(n, 0);
    std::vector<double> suffix((size_t)n + 1);
    int lg = 0;
    while ((1 << lg) < n) lg++;
    const int guard_depth = 2 * lg + 16;

    struct Task { int32_t b, e, slot, depth; };
    std::vector<Task> todo;
    todo.push_back({0, n, 0, 1});
    int depth = 1;
    while (!todo.empty()) {
        const Task t = todo.back();
        todo.pop_back();
        const int m = t.e - t.b;
        if (t.depth > depth) depth = t.depth;
        pt_bvh_node& nd = out[t.slot];
        if (m == 1) {
            const int32_t p = idx[0][t.b];
            std::memcpy(nd.bmin, leaf_boxes + (size_t)p * 6, 12);
            std::memcpy(nd.bmax, leaf_boxes + (size_t)p * 6 + 3, 12);
            nd.left = -1; nd.right = -1; nd.prim = p;
            continue;
        }
        int best_axis = 0, best_k = m / 2;
        SBox whole = sbox_empty();
        if (t.depth > guard_depth) {
            // median cut along the axis with the widest centroid spread
            float spread = -1.0f;
            for (int a = 0; a < 3; a++) {
                const float s = cen[(size_t)idx[a][t.e - 1] * 3 + a] - cen[(size_t)idx[a][t.b] * 3 + a];
                if (s > spread) { spread = s; best_axis = a; }
            }
            for (int i = t.b; i < t.e; i++) sbox_merge(whole, leaf_boxes + (size_t)idx[0][i] * 6);
        } else {
            double best_cost = 0.0;
            int best_off = 0;
            bool have = false;
            for (int a = 0; a < 3; a++) {
                const int32_t* ix = idx[a].data();
                SBox acc = sbox_empty();
                for (int i = t.e - 1; i > t.b; i--) {                 // suffix[i - b] = area of the box of [i, e)
                    sbox_merge(acc, leaf_boxes + (size_t)ix[i] * 6);
                    suffix[i - t.b] = sbox_area(acc);
                }
                if (a == 0) { whole = acc; sbox_merge(whole, leaf_boxes + (size_t)ix[t.b] * 6); }
                acc = sbox_empty();
                for (int k = 1; k < m; k++) {                          // cut after the first k of this order
                    sbox_merge(acc, leaf_boxes + (size_t)ix[t.b + k - 1] * 6);
                    const double cost = sbox_area(acc) * k + suffix[k] * (m - k);
                    const int off = 2 * k > m ? 2 * k - m : m - 2 * k;
                    if (!have || cost < best_cost || (cost == best_cost && off < best_off)) {
                        have = true; best_cost = cost; best_off = off; best_axis = a; best_k = k;
                    }
                }
            }
        }
        // the other two orders follow: stable partition by membership in the left set
        const int32_t* chosen = idx[best_axis].data();
        for (int i = t.b; i < t.b + best_k; i++) left_side[chosen[i]] = 1;
        for (int a = 0; a < 3; a++) {
            if (a == best_axis) continue;
            int32_t* ix = idx[a].data();
            int l = t.b, r = 0;
            for (int i = t.b; i < t.e; i++) {
                const int32_t p = ix[i];
                if (left_side[p]) ix[l++] = p; else tmp[r++] = p;
            }
            std::memcpy(ix + l, tmp.data(), (size_t)r * sizeof(int32_t));
        }
        for (int i = t.b; i < t.b + best_k; i++) left_side[chosen[i]] = 0;
        std::memcpy(nd.bmin, whole.lo, 12);
        std::memcpy(nd.bmax, whole.hi, 12);
        nd.prim = -1;
        nd.left = t.slot + 1;
        nd.right = t.slot + 2 * best_k;                               // the left subtree holds 2 k - 1 nodes
        todo.push_back({t.b + best_k, t.e, nd.right, t.depth + 1});
        todo.push_back({t.b, t.b + best_k, nd.left, t.depth + 1});
    }
    *out_root = 0;
    *out_depth = depth;
}

}  // namespace pts
