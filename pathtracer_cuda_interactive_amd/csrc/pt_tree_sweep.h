// Host side, scene creation only: the library's internal tree over the caller's LEAF boxes (pt_api.hip: validate_and_build
// explains why any tree over the same leaf boxes gives the reference's result).
//
// Exact traversal never prunes (scene.h:258-297), so a ray pays for every inner box it touches and the tree is the whole
// cost of a segment.  This is the classic top-down build that looks at ALL cuts: at every node the primitives are swept
// along x, y and z in centroid order and the cut with the smallest  SA(left) * n_left + SA(right) * n_right  wins (ties:
// the cut nearest the middle, then the lower axis, then the lower position — deterministic).  The three orders are sorted
// once and kept sorted by stable partitions, O(n log n) box merges for a balanced tree; a depth guard switches a
// pathological branch to median cuts so that the build stays near that bound.  Big inputs: the top of the tree node by node,
// then whole subtrees on up to 8 threads (a subtree is a function of its range alone: the result does not depend on threads).
// Measured on bunny (288,094 primitives, inner visits per segment, exact traversal): caller's median-split tree 52.0, cuts
// of the Morton order (pt_bvh_build.hip, PT_BVH_DEVICE_SAH) 23.2, this 18.0.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <exception>
#include <new>
#include <system_error>
#include <thread>
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
inline void sbox_merge(SBox& a, const SBox& b) {
    for (int k = 0; k < 3; k++) {
        a.lo[k] = b.lo[k] < a.lo[k] ? b.lo[k] : a.lo[k];
        a.hi[k] = b.hi[k] > a.hi[k] ? b.hi[k] : a.hi[k];
    }
}
inline double sbox_area(const SBox& b) {
    const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
    return 2.0 * (x * y + y * z + z * x);
}

// Runs the jobs 1 .. count-1 on threads of their own and job 0 on the caller's; a thread that cannot be started (process or
// thread limits of the host) simply means its job runs on the caller's thread as well.  Jobs must be independent.
// An exception thrown by any job (std::bad_alloc in a sort key array, say) is caught on its thread, every thread is joined,
// and the first one caught is rethrown on the caller's thread — never std::terminate.
template <class F>
inline void run_side_by_side(int count, F&& job) {
    std::vector<std::thread> side;
    std::vector<int> inline_jobs;
    std::vector<std::exception_ptr> failed((size_t)(count > 0 ? count : 1));
    auto guarded = [&job, &failed](int k) {
        try {
            job(k);
        } catch (...) {
            failed[(size_t)k] = std::current_exception();
        }
    };
    for (int k = 1; k < count; k++) {
        try {
            side.emplace_back(guarded, k);
        } catch (const std::system_error&) {
            inline_jobs.push_back(k);
        } catch (const std::bad_alloc&) {
            inline_jobs.push_back(k);
        }
    }
    guarded(0);
    for (int k : inline_jobs) guarded(k);
    for (std::thread& t : side) t.join();
    for (const std::exception_ptr& e : failed)
        if (e) std::rethrow_exception(e);
}

// leaf_boxes: n x {lo.xyz, hi.xyz}, all finite.  out: the reference's node pool layout (bvh.cuh:7-15), 2n-1 nodes in
// pre-order (node, left subtree, right subtree), root = 0.  Inner boxes are exact unions (min / max do not round).
inline void build_sweep_tree(const float* leaf_boxes, int n, std::vector<pt_bvh_node>& out, int32_t* out_root, int32_t* out_depth) {
    out.assign((size_t)2 * n - 1, pt_bvh_node{});
    std::vector<float> cen((size_t)n * 3);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) cen[(size_t)i * 3 + k] = (leaf_boxes[(size_t)i * 6 + k] + leaf_boxes[(size_t)i * 6 + 3 + k]) * 0.5f;
    // Three orders of the primitives, by centroid along x, y, z (ties: id).  Every order carries its own copy of the boxes
    // (bx[a][i] = box of primitive idx[a][i]): the sweeps and partitions below then stream through memory instead of chasing
    // ids — at a million primitives that is the difference between cache misses and bandwidth.
    std::vector<int32_t> idx[3];
    std::vector<SBox> bx[3];
    auto sort_axis = [&](int a) {
        std::vector<std::pair<float, int32_t>> key(n);
        for (int i = 0; i < n; i++) key[i] = {cen[(size_t)i * 3 + a], i};
        std::sort(key.begin(), key.end());                        // (centroid, id): a total order
        idx[a].resize(n);
        bx[a].resize(n);
        for (int i = 0; i < n; i++) {
            idx[a][i] = key[i].second;
            std::memcpy(&bx[a][i], leaf_boxes + (size_t)key[i].second * 6, sizeof(SBox));
        }
    };
    if (n >= 65536) {
        run_side_by_side(3, sort_axis);
    } else {
        for (int a = 0; a < 3; a++) sort_axis(a);
    }
    // scratch per order, indexed like idx[]: tasks own disjoint ranges of all of these
    std::vector<int32_t> tmp[3];
    std::vector<SBox> tmp_box[3];
    std::vector<double> suffix[3];
    for (int a = 0; a < 3; a++) { tmp[a].resize(n); tmp_box[a].resize(n); suffix[a].resize((size_t)n + 1); }
    std::vector<unsigned char> left_side(n, 0);      // by primitive id: a task only touches its own primitives
    int lg = 0;
    while ((1 << lg) < n) lg++;
    const int guard_depth = 2 * lg + 16;

    struct Task { int32_t b, e, slot, depth; };
    // One node: leaf, or choose the cut, partition the three orders, emit the node, hand back the two children.
    // Returns the number of children pushed (0 or 2).  Everything it writes lies in [t.b, t.e) of the shared arrays, in the
    // task's own node slots, or belongs to the task's own primitives — tasks on disjoint ranges can run concurrently.
    auto process = [&](const Task& t, Task* kids, const bool wide) -> int {
        const int m = t.e - t.b;
        pt_bvh_node& nd = out[t.slot];
        if (m == 1) {
            const int32_t p = idx[0][t.b];
            std::memcpy(nd.bmin, bx[0][t.b].lo, 12);
            std::memcpy(nd.bmax, bx[0][t.b].hi, 12);
            nd.left = -1; nd.right = -1; nd.prim = p;
            return 0;
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
            for (int i = t.b; i < t.e; i++) sbox_merge(whole, bx[0][i]);
        } else {
            // every axis on its own (on its own thread for the big nodes at the top of the tree), then the best of the three
            struct AxisBest { double cost; int off, k; SBox whole; };
            AxisBest ab[3];
            auto sweep = [&](int a) {
                const SBox* bb = bx[a].data();
                double* suf = suffix[a].data();
                SBox acc = sbox_empty();
                for (int i = t.e - 1; i > t.b; i--) {                 // suf[i] = area of the box of [i, e)
                    sbox_merge(acc, bb[i]);
                    suf[i] = sbox_area(acc);
                }
                ab[a].whole = acc;
                sbox_merge(ab[a].whole, bb[t.b]);
                acc = sbox_empty();
                bool have = false;
                for (int k = 1; k < m; k++) {                          // cut after the first k of this order
                    sbox_merge(acc, bb[t.b + k - 1]);
                    const double cost = sbox_area(acc) * k + suf[t.b + k] * (m - k);
                    const int off = 2 * k > m ? 2 * k - m : m - 2 * k;
                    if (!have || cost < ab[a].cost || (cost == ab[a].cost && off < ab[a].off)) {
                        have = true; ab[a].cost = cost; ab[a].off = off; ab[a].k = k;
                    }
                }
            };
            if (wide) {
                run_side_by_side(3, sweep);
            } else {
                for (int a = 0; a < 3; a++) sweep(a);
            }
            whole = ab[0].whole;
            best_axis = 0;
            for (int a = 1; a < 3; a++)
                if (ab[a].cost < ab[best_axis].cost || (ab[a].cost == ab[best_axis].cost && ab[a].off < ab[best_axis].off)) best_axis = a;
            best_k = ab[best_axis].k;
        }
        // the other two orders follow: stable partition by membership in the left set
        const int32_t* chosen = idx[best_axis].data();
        for (int i = t.b; i < t.b + best_k; i++) left_side[chosen[i]] = 1;
        auto partition = [&](int a) {
            int32_t* ix = idx[a].data();
            SBox* bb = bx[a].data();
            int32_t* ti = tmp[a].data();
            SBox* tb = tmp_box[a].data();
            int l = t.b, r = t.b;
            for (int i = t.b; i < t.e; i++) {
                const int32_t p = ix[i];
                if (left_side[p]) { ix[l] = p; bb[l] = bb[i]; l++; } else { ti[r] = p; tb[r] = bb[i]; r++; }
            }
            std::memcpy(ix + l, ti + t.b, (size_t)(r - t.b) * sizeof(int32_t));
            std::memcpy(bb + l, tb + t.b, (size_t)(r - t.b) * sizeof(SBox));
        };
        const int oa = (best_axis + 1) % 3, ob = (best_axis + 2) % 3;
        if (wide) {
            run_side_by_side(2, [&](int k) { partition(k == 0 ? oa : ob); });
        } else {
            partition(oa);
            partition(ob);
        }
        for (int i = t.b; i < t.b + best_k; i++) left_side[chosen[i]] = 0;
        std::memcpy(nd.bmin, whole.lo, 12);
        std::memcpy(nd.bmax, whole.hi, 12);
        nd.prim = -1;
        nd.left = t.slot + 1;
        nd.right = t.slot + 2 * best_k;                               // the left subtree holds 2 k - 1 nodes
        kids[0] = {t.b, t.b + best_k, nd.left, t.depth + 1};
        kids[1] = {t.b + best_k, t.e, nd.right, t.depth + 1};
        return 2;
    };
    auto run_subtree = [&](const Task& root) -> int {                 // depth-first to completion; returns the deepest level met
        std::vector<Task> todo;
        todo.push_back(root);
        int deepest = root.depth;
        Task kids[2];
        while (!todo.empty()) {
            const Task t = todo.back();
            todo.pop_back();
            if (t.depth > deepest) deepest = t.depth;
            if (process(t, kids, false) == 2) { todo.push_back(kids[1]); todo.push_back(kids[0]); }
        }
        return deepest;
    };

    int depth = 1;
    unsigned workers = n >= 65536 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (workers <= 1) {
        depth = run_subtree(Task{0, n, 0, 1});
    } else {
        // The top of the tree one node at a time, largest first, until there are enough subtrees to share out; then every
        // worker takes whole subtrees.  Which thread builds which subtree changes nothing: every task is a function of its
        // range alone and writes to its own slots.
        std::vector<Task> pending;
        pending.push_back(Task{0, n, 0, 1});
        Task kids[2];
        for (;;) {
            size_t big = 0;
            for (size_t k = 1; k < pending.size(); k++)
                if (pending[k].e - pending[k].b > pending[big].e - pending[big].b) big = k;
            const int size = pending[big].e - pending[big].b;
            if (pending.size() >= 16u * workers || size <= std::max(4096, n / (int)(64u * workers))) break;
            const Task t = pending[big];
            pending[big] = pending.back();
            pending.pop_back();
            if (t.depth > depth) depth = t.depth;
            if (process(t, kids, true) == 2) { pending.push_back(kids[0]); pending.push_back(kids[1]); }
        }
        std::sort(pending.begin(), pending.end(), [](const Task& x, const Task& y) { return (x.e - x.b) > (y.e - y.b) || ((x.e - x.b) == (y.e - y.b) && x.b < y.b); });
        std::atomic<size_t> next{0};
        std::vector<int> deepest(workers, 1);
        run_side_by_side((int)workers, [&](int w) {
            for (size_t k = next.fetch_add(1); k < pending.size(); k = next.fetch_add(1))
                deepest[w] = std::max(deepest[w], run_subtree(pending[k]));
        });
        for (int dd : deepest) depth = std::max(depth, dd);
    }
    *out_root = 0;
    *out_depth = depth;
}

}  // namespace pts
