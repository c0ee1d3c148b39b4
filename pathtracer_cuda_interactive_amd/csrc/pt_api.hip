// pt_api.hip — kernels and C ABI of libpt_hip.so (gfx950 only).  See include/pt_api.h.
//
// Host half of the library: scene validation and re-layout, scratch management, kernel selection and launch,
// the extern "C" entry points.  The kernels live in pt_kernels.h, the device functions in pt_trace.h / pt_math.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_internal.h"
#include "pt_tree_sweep.h"
#include "pt_sweep_build.h"
#include "pt_scene_prep.h"
#include "pt_kernels.h"
#include "pt_kernel_q.h"

using namespace ptl;
using namespace ptk;

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

}  // namespace

int pt_fail(int code, const std::string& msg) { return fail(code, msg); }

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorNoDevice ? PT_ERR_NO_DEVICE : PT_ERR_DEVICE,                 \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }          // the owner makes the buffer's device current first (pt_scene_destroy)
    int ensure(size_t count) {
        if (count <= n && p) return PT_OK;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)));
        n = count;
        return PT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

// Makes the scene's device current for the duration of a call and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int enter(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) {
            HIP_TRY(hipSetDevice(device));
            switched = true;
        }
        return PT_OK;
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};

constexpr uint32_t kLdsSceneLimit = 36 * 1024;   // scenes up to this size (64-B nodes) are staged whole into LDS
constexpr uint32_t kOctNodeLimit = 24 * 1024;    // 8 octant copies of the node table must fit in this many bytes of LDS
constexpr int kMaxStack = 64;                    // the reference's own cap (scene.h:251)
constexpr int kMinInternalTree = 16;              // primitives below which no internal tree is built (a handful of nodes either way: scene1, 4 shapes, is 6 % slower with one)
constexpr int kDeviceSweepMin = 4096;             // primitives from which the internal tree is built on the device (below: a few ms on the host either way)
constexpr int kProbeRays = 32768;                // validate_and_build: rays that choose between the caller's tree and the internal one
constexpr uint64_t kDefaultScratchBytes = 8ull << 30;   // per-sample scratch cap (3 % of the 288 GB of HBM): every sample pass
                                                        // pays the launch floor once (buddha stand-in 135.6 ms in 5 passes, 130.6 in 1)
#ifndef PT_TOP_NODES
#define PT_TOP_NODES 512
#endif
constexpr uint32_t kTopNodes = PT_TOP_NODES;              // scenes read from global memory: this many nodes are numbered breadth-first
                                                 // from the root, so that [0, k) is the top of the tree for every k (LDS cache)
#ifndef PT_TOP_LDS_KB
#define PT_TOP_LDS_KB 26
#endif
constexpr uint32_t kTopLdsBudget = PT_TOP_LDS_KB * 1024;    // LDS per block that keeps 6 blocks per CU resident (160 KB / 6)
constexpr uint32_t kMaxLdsBudget = 160 * 1024;               // the breadth-first numbered prefix is sized for the largest budget an option may ask for
constexpr int kMaxFramesInFlight = 4;            // frame slots a handle may hold (option frames_in_flight)
#ifndef PT_FRAMES_IN_FLIGHT
#define PT_FRAMES_IN_FLIGHT 2
#endif
constexpr int kDefaultFramesInFlight = PT_FRAMES_IN_FLIGHT;
constexpr size_t kWorkBytes = 8 * kCounterStride * sizeof(uint32_t);   // 8 band counters, one 128-B line each
constexpr size_t kWorkWords = kWorkBytes / sizeof(unsigned long long);

uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }

// every float finite?  (leaf boxes that stayed on the device)
__global__ void finite_kernel(const float* __restrict__ v, size_t n, unsigned int* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !__builtin_isfinite(v[i])) atomicAdd(bad, 1u);
}
bool boxes_finite_device(const float* v_dev, size_t n) {
    unsigned int* bad = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&bad), sizeof(unsigned int)) != hipSuccess) return false;
    unsigned int h = 1;
    if (hipMemset(bad, 0, sizeof(unsigned int)) == hipSuccess) {
        hipLaunchKernelGGL(finite_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, v_dev, n, bad);
        if (hipMemcpy(&h, bad, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) h = 1;
    }
    (void)hipFree(bad);
    return h == 0;
}

// PT_SWEEP_BUILD=host in the environment keeps the internal tree's build on the host (A/B timing, debugging)
bool host_sweep_forced() {
    const char* v = std::getenv("PT_SWEEP_BUILD");
    return v && std::string(v) == "host";
}

}  // namespace

struct pt_scene {
    int device = 0;
    int num_cus = 0;
    size_t lds_per_block_max = 0;
    // scene arrays.  Two layouts of the hierarchy: tree[0] = the caller's tree (the reference's, bvh.cu:16-54), tree[1] = the
    // library's internal tree over the SAME leaf boxes, made at create time (validate_and_build: the full-sweep surface-area
    // tree of pt_tree_sweep.h / pt_sweep_build.hip, or the caller's own topology in internal form).  Exact traversal runs on
    // tree[1] and falls back to tree[0] for the rays whose result depends on the visit order (see launch_render).
    struct Tree {
        DevBuf<DNode> nodes;
        DevBuf<DNode> nodes_oct;     // [8][num_nodes] octant-specialised copies (small scenes only)
        bool have_oct = false;
        int32_t num_nodes = 0, root_ref = 0, stack_cap = 0;
        uint32_t top_avail = 0;      // nodes [0, top_avail) are the top of the tree in breadth-first order (0 = plain pre-order)
        int depth = 0;
    } tree[2];
    bool have_fast = false;
    DevBuf<unsigned long long> ref_path;   // per primitive: its root-to-leaf turns in the caller's tree (ties on t, pt_trace.h: ref_visits_first)
    DevBuf<int32_t> ref_anc;         // per primitive and level: the inner node of the caller's tree there
    int32_t ref_levels = 0;
    bool fast_is_callers_topology = false;   // tree[1] = the caller's own tree in internal form (the sweep tree did not win)
    DevBuf<DPrim> prims;
    DevBuf<DNormals> normals;
    DevBuf<DMaterial> materials;
    DevBuf<DEmission> emission;
    DevBuf<DLight> lights;
    SceneDev dev{};
    uint32_t scene_bytes = 0;
    bool tri_only = false;           // the scene holds no sphere
    bool diffuse_only = false;       // every material is DIFFUSE
    // scratch
    DevBuf<float> accum;
    DevBuf<float> fb_tmp;
    // What a frame's trace kernel writes lives in a frame slot; render call k uses slot k % frames_in_flight.  With more than one
    // slot a single-pass frame runs on the slot's own stream: the trace kernel at once (it reads the immutable scene and writes
    // the slot only), the resolve — the one step that touches the caller's buffer — once the caller's stream has reached the
    // point of the call (in_ev), and the caller's stream continues behind the resolve (free_ev).  The next frame's trace kernel
    // then fills the CUs while this one's last paths drain (the tail of a persistent launch is a partly empty chip), and
    // everything the caller can observe stays in the order of the caller's stream.  Both waits across streams sit beside the
    // chain trace k -> resolve k -> trace k + slots, which is in order on ONE stream (an event wait across streams costs tens of
    // microseconds on this runtime; with the resolve on the caller's stream, two of them per frame sat on that chain and
    // two slots rendered slower than one).  A slot is reused once the resolve that read its samples has finished (free_ev).
    struct FrameSlot {
        DevBuf<float4> samples;
        // control block: [0, kWorkBytes) the band work counters, then the statistics (kNumCounters uint64, pt_kernels.h)
        DevBuf<unsigned long long> ctl;
        DevBuf<int32_t> redo_stack;  // global-memory traversal stacks of the reference-order reruns (one column per lane of the grid)
        hipStream_t stream = nullptr;
        hipEvent_t free_ev = nullptr, in_ev = nullptr;
        hipStream_t free_stream = nullptr;   // the stream free_ev was last recorded on
        bool used = false;
    } slot[kMaxFramesInFlight];
    int cur_slot = 0;                // the slot of the last render call
    int64_t opt_frames_in_flight = kDefaultFramesInFlight;
    FrameSlot& cur() { return slot[cur_slot]; }
    const FrameSlot& cur() const { return slot[cur_slot]; }
    uint32_t* work_counter() const { return reinterpret_cast<uint32_t*>(cur().ctl.p); }
    unsigned long long* counters() const { return cur().ctl.p ? cur().ctl.p + kWorkWords : nullptr; }
    void drop_slots() {
        for (auto& f : slot) {
            f.samples.release(); f.ctl.release(); f.redo_stack.release();
            if (f.stream) (void)hipStreamDestroy(f.stream);
            if (f.free_ev) (void)hipEventDestroy(f.free_ev);
            if (f.in_ev) (void)hipEventDestroy(f.in_ev);
            f.stream = nullptr; f.free_ev = nullptr; f.in_ev = nullptr; f.used = false;
        }
    }
    hipStream_t last_stream = nullptr;
    bool have_timing = false;
    // options
    int64_t opt_blocks_per_cu = 0;
    int64_t opt_scratch_bytes = 0;
    int64_t opt_force_global = 0;
    int64_t opt_stats = 0;
    int64_t opt_specialize = 1;      // compile-time specialisation on scene content (no spheres -> sphere code removed)
    int64_t opt_octants = 1;         // use the 8 ray-octant node tables when the scene is small enough
    int64_t opt_top_cache = 1;       // scenes in global memory: keep the top of the tree in LDS
    int64_t opt_fast_tree = 1;       // exact traversal on the library's internal tree where one was kept (0 = on the caller's tree)
    int64_t opt_chunk = 0;           // work items a wave reserves per atomic (0 = automatic)
    int64_t opt_lds_budget_kb = 0;   // scenes in global memory: LDS per block for traversal stacks + top-of-tree cache (0 = 26 KB: 6 blocks per CU)
    int64_t opt_item_order = 1;      // work item order inside a band: 1 = row-major (all samples of a row, then the next row), 0 = sample-major
    int64_t opt_xcd_regions = 0;     // 0 = 8 row bands (one per XCD); 1 = a single work queue
    int64_t opt_kernel = 2;          // 2 = decoupled traversal/shading (default), 1 = segment-synchronous wavefront kernel,
                                     // 3 = paths regrouped across the waves of a workgroup (pt_kernel_q.h; LDS-resident scenes, else as 2)
    int64_t opt_q_target = 0, opt_q_swap = 0, opt_q_low = 0;   // schedule knobs of kernel 3; 0 = automatic
    int64_t opt_v2_thresh = 0, opt_v2_inner = 0, opt_v2_minw = 0;   // 0 = auto (see pick_kernel)
    // info of last launch
    // wall time of pt_scene_create's stages, microseconds: [0] total [1] primitive records [2] caller's tree checked and re-laid
    // [3] internal tree built [4] ... re-laid [5] uploads + probe [6] tie tables; built_on_device: the sweep ran on the GPU
    int64_t create_us[7] = {0, 0, 0, 0, 0, 0, 0}, sweep_on_device = 0;
    int64_t info_grid = 0, info_lds_bytes = 0, info_lds_scene = 0, info_passes = 0, info_occupancy = 0, info_blocks_per_cu = 0, info_debug_reruns = 0, fast_cost_permille = 0, info_kernel = 0;
    struct PassEvents { hipEvent_t t0, t1, r0, r1; };            // trace begin, trace end (on the trace kernel's stream); resolve begin, end (caller's)
    // HIP events of the last `opt_timing_frames` render calls (a ring; default 1): a caller that enqueues frame after frame
    // without a host sync in between — bench.py's timed loop — reads every frame's kernel time afterwards (pt_get_frame_times)
    struct FrameRec { std::vector<PassEvents> ev; size_t passes = 0; };
    std::vector<FrameRec> frames;
    uint64_t frame_seq = 0;                                      // render calls so far; the last one sits in frames[(frame_seq - 1) % size]
    int64_t opt_timing_frames = 1;
    const FrameRec* last_frame() const { return frame_seq && !frames.empty() ? &frames[(frame_seq - 1) % frames.size()] : nullptr; }
    void drop_events() {
        for (auto& f : frames)
            for (auto& pe : f.ev) { (void)hipEventDestroy(pe.t0); (void)hipEventDestroy(pe.t1); (void)hipEventDestroy(pe.r0); (void)hipEventDestroy(pe.r1); }
        frames.clear();
        frame_seq = 0;
    }
    // launch configuration of the last kernel variant used (occupancy query and attribute call are not free per frame)
    const void* cfg_fn = nullptr;
    uint32_t cfg_lds = 0;
    int cfg_occ = 0;
};

namespace {

// Points the kernel argument block at one of the two trees; fallback: exact traversal on the internal tree — ties settled in the
// caller's visit order, rays with a zero direction component traced on tree[0].
void select_tree(pt_scene* S, int which, bool fallback = false) {
    const pt_scene::Tree& T = S->tree[which];
    SceneDev& dv = S->dev;
    dv.nodes = T.nodes.p;
    dv.nodes_oct = T.have_oct ? T.nodes_oct.p : nullptr;
    dv.num_nodes = T.num_nodes;
    dv.root_ref = T.root_ref;
    dv.stack_cap = T.stack_cap;
    dv.fallback = fallback ? 1 : 0;
    dv.ref_nodes = S->tree[0].nodes.p;
    dv.ref_root_ref = S->tree[0].root_ref;
    dv.redo_cap = S->tree[0].stack_cap;
    dv.redo_stack = S->cur().redo_stack.p;
    dv.fixed_order = which == 1 ? 1 : 0;
    dv.ref_path = S->ref_path.p;
    dv.ref_anc = S->ref_anc.p;
    dv.ref_levels = S->ref_levels;
}

struct TreeHost {
    std::vector<DNode> nodes;
    int32_t root_ref = 0;
    int depth = 1;
    int stack_need = 1;            // stack entries a traversal can hold at once (without the sentinel)
    std::vector<int32_t> inner_of_pool;   // caller's node pool index -> index in `nodes` (-1: a leaf)
    uint32_t top_avail = 0;
};

// pt_bvh_node pool (one node per leaf and per inner node, bvh.cuh:7-15) -> inner-only DNodes carrying both child boxes.
// Validates the topology.  leaf_boxes (optional, [N][6]): receives every primitive's leaf box.  *nested: every node's box
// below the root contains its children's boxes.
int convert_tree(const pt_bvh_node* in, int num_nodes, int root, int N, TreeHost& out, float* leaf_boxes, bool* nested, bool internal = false) {
    std::vector<int32_t> inner_id(num_nodes, -1);
    std::vector<DNode>& nodes = out.nodes;
    nodes.clear();
    nodes.reserve(N > 1 ? N - 1 : 1);
    int depth = 1;
    std::vector<char> prim_seen(N, 0);
    auto leaf_ok = [&](const pt_bvh_node& nd) { return nd.prim >= 0 && nd.prim < N; };
    auto contains = [](const pt_bvh_node& outer, const pt_bvh_node& inner) {
        for (int k = 0; k < 3; k++)
            if (!(inner.bmin[k] >= outer.bmin[k] && inner.bmax[k] <= outer.bmax[k])) return false;
        return true;
    };
    const pt_bvh_node& rootn = in[root];
    if (rootn.prim != -1) {
        if (!leaf_ok(rootn)) return fail(PT_ERR_BAD_SCENE, "leaf primitive id out of range");
        out.root_ref = ~rootn.prim;
        prim_seen[rootn.prim] = 1;
        if (leaf_boxes) { std::memcpy(leaf_boxes + 6 * (size_t)rootn.prim, rootn.bmin, 12); std::memcpy(leaf_boxes + 6 * (size_t)rootn.prim + 3, rootn.bmax, 12); }
    } else {
        // The internal tree is traversed left child first (pt_trace.h: visit_node).  While the left subtree is being worked
        // on, the right child waits on the stack: entries needed = max(1 + need(left), need(right)), so the child that needs
        // less goes LEFT and the need of a node is the larger of its children's, plus one only when they are equal — the
        // Strahler number of the tree, at most log2(leaves) + 1 whatever its depth.
        std::vector<int32_t> need;
        if (internal) {                                               // (the library's own tree: no validation needed here)
            need.assign(num_nodes, 0);
            std::vector<int32_t> pre, open;
            open.push_back(root);
            while (!open.empty()) {
                const int32_t k = open.back();
                open.pop_back();
                pre.push_back(k);
                for (int32_t ch : {in[k].left, in[k].right})
                    if (in[ch].prim == -1) open.push_back(ch);
            }
            for (size_t k = pre.size(); k-- > 0;) {                  // parents come before their children in `pre`
                const pt_bvh_node& nd = in[pre[k]];
                const int32_t a = need[nd.left], b = need[nd.right];
                need[pre[k]] = a == b ? a + 1 : std::max(a, b);
            }
            out.stack_need = need[root];
        }
        auto swapped = [&](const pt_bvh_node& nd) { return internal && need[nd.left] > need[nd.right]; };
        // DFS pre-order: a parent and the subtree of the child it lists first are contiguous in memory
        struct Item { int32_t ref_node; int depth; };
        std::vector<Item> todo;
        todo.push_back({root, 1});
        size_t visited = 0;
        std::vector<int32_t> order;
        while (!todo.empty()) {
            Item it = todo.back();
            todo.pop_back();
            const pt_bvh_node& nd = in[it.ref_node];
            if (++visited > (size_t)num_nodes) return fail(PT_ERR_BAD_SCENE, "BVH is not a tree (cycle)");
            if (inner_id[it.ref_node] != -1) return fail(PT_ERR_BAD_SCENE, "BVH node referenced twice");
            if (nd.left < 0 || nd.left >= num_nodes || nd.right < 0 || nd.right >= num_nodes)
                return fail(PT_ERR_BAD_SCENE, "BVH child index out of range");
            inner_id[it.ref_node] = (int32_t)order.size();
            order.push_back(it.ref_node);
            if (it.depth + 1 > depth) depth = it.depth + 1;
            const pt_bvh_node& ln = in[nd.left];
            const pt_bvh_node& rn = in[nd.right];
            if (it.ref_node != root && nested && !(contains(nd, ln) && contains(nd, rn))) *nested = false;
            const bool sw = swapped(nd);
            const int32_t first = sw ? nd.right : nd.left, second = sw ? nd.left : nd.right;
            if (in[second].prim == -1) todo.push_back({second, it.depth + 1});
            if (in[first].prim == -1) todo.push_back({first, it.depth + 1});
        }
        // Scenes too big for LDS: renumber so that the first kTopNodes ids are the top of the tree in breadth-first order
        // (every ray starts there; the kernel keeps a prefix of them in LDS), the rest stays in pre-order.
        if ((size_t)N * (sizeof(DPrim) + sizeof(DNormals)) + order.size() * sizeof(DNode) > kLdsSceneLimit && order.size() > 1) {
            // only as many as fit next to the traversal stacks (make_plan)
            const uint32_t stack_bytes = (uint32_t)(kBlock / 64) * (uint32_t)((internal ? out.stack_need : std::max(depth - 1, 1)) + 1) * 64u * 4u;
            const size_t want = std::min<size_t>(kTopNodes, kMaxLdsBudget > stack_bytes ? (kMaxLdsBudget - stack_bytes) / sizeof(DNode) : 0);
            // which ones: without pruning a node is visited iff the ray hits its box, so the nodes most rays visit are the ones
            // with the largest boxes — take them in order of surface area (a child's box is never larger than its parent's,
            // so every prefix of this list is closed under "parent of"; the kernel stages a prefix)
            std::vector<int32_t> top;
            std::vector<char> in_top(num_nodes, 0);
            auto area = [&](int32_t k) {
                const pt_bvh_node& b = in[k];
                const double x = (double)b.bmax[0] - b.bmin[0], y = (double)b.bmax[1] - b.bmin[1], z = (double)b.bmax[2] - b.bmin[2];
                return 2.0 * (x * y + y * z + z * x);
            };
            typedef std::pair<double, int32_t> Cand;                  // (area, -position in pre-order): ties go to the earlier node
            std::priority_queue<Cand> open;
            if (want) open.push(Cand(std::numeric_limits<double>::infinity(), 0));
            while (!open.empty() && top.size() < want) {
                const int32_t k = order[(size_t)-open.top().second];
                open.pop();
                top.push_back(k);
                const pt_bvh_node& nd = in[k];
                for (int32_t ch : {nd.left, nd.right})
                    if (in[ch].prim == -1) open.push(Cand(area(ch), -inner_id[ch]));
            }
            for (int32_t r : top) in_top[r] = 1;
            std::vector<int32_t> renum(top);
            for (int32_t r : order)
                if (!in_top[r]) renum.push_back(r);
            order.swap(renum);
            for (size_t k = 0; k < order.size(); k++) inner_id[order[k]] = (int32_t)k;
            out.top_avail = (uint32_t)top.size();
        }
        nodes.resize(order.size());
        for (size_t k = 0; k < order.size(); k++) {
            const pt_bvh_node& nd = in[order[k]];
            const bool sw = swapped(nd);
            const int32_t li = sw ? nd.right : nd.left, ri = sw ? nd.left : nd.right;
            const pt_bvh_node& ln = in[li];
            const pt_bvh_node& rn = in[ri];
            DNode& o = nodes[k];
            std::memcpy(o.lmin, ln.bmin, 12); std::memcpy(o.lmax, ln.bmax, 12);
            std::memcpy(o.rmin, rn.bmin, 12); std::memcpy(o.rmax, rn.bmax, 12);
            for (const pt_bvh_node* ch : {&ln, &rn})
                if (ch->prim != -1) {
                    if (!leaf_ok(*ch)) return fail(PT_ERR_BAD_SCENE, "leaf primitive id out of range");
                    if (prim_seen[ch->prim]) return fail(PT_ERR_BAD_SCENE, "primitive referenced by two leaves");
                    prim_seen[ch->prim] = 1;
                    if (leaf_boxes) { std::memcpy(leaf_boxes + 6 * (size_t)ch->prim, ch->bmin, 12); std::memcpy(leaf_boxes + 6 * (size_t)ch->prim + 3, ch->bmax, 12); }
                }
            o.left = ln.prim != -1 ? ~ln.prim : inner_id[li];
            o.right = rn.prim != -1 ? ~rn.prim : inner_id[ri];
            o.pad0 = o.pad1 = 0;
        }
        out.root_ref = 0;
    }
    for (int i = 0; i < N; i++)
        if (!prim_seen[i]) return fail(PT_ERR_BAD_SCENE, "primitive not covered by any leaf");
    if (!internal && depth - 1 > kMaxStack - 1)
        return fail(PT_ERR_BAD_SCENE, "BVH deeper than the traversal stack (reference cap 64, scene.h:251)");
    out.depth = depth;
    out.inner_of_pool.swap(inner_id);
    if (!internal) out.stack_need = std::max(depth - 1, 1);           // nearer child first: a pending sibling per level
    if (nodes.empty()) nodes.resize(1);   // single-primitive scene: no inner nodes; keep a dummy so pointers are valid
    return PT_OK;
}

int validate_and_build(const pt_scene_desc* d, pt_scene* S) {
    if (!d) return fail(PT_ERR_INVALID_ARG, "null scene description");
    if (d->num_shapes <= 0 || !d->shapes) return fail(PT_ERR_BAD_SCENE, "scene has no shapes");
    if (d->num_nodes != 2 * d->num_shapes - 1 || !d->nodes)
        return fail(PT_ERR_BAD_SCENE, "BVH must have 2*num_shapes-1 nodes (bvh.cu:16-54)");
    if (d->root < 0 || d->root >= d->num_nodes) return fail(PT_ERR_BAD_SCENE, "BVH root out of range");
    if (d->num_materials <= 0 || !d->materials) return fail(PT_ERR_BAD_SCENE, "scene has no materials");
    if (d->num_meshes < 0 || (d->num_meshes > 0 && !d->meshes)) return fail(PT_ERR_BAD_SCENE, "bad mesh array");
    if (d->num_lights < 0 || (d->num_lights > 0 && !d->lights)) return fail(PT_ERR_BAD_SCENE, "bad light array");
    for (int m = 0; m < d->num_materials; m++)
        if (d->materials[m].type < PT_MAT_DIFFUSE || d->materials[m].type > PT_MAT_PHONG)
            return fail(PT_ERR_BAD_SCENE, "unknown material type (scene.h:410 asserts)");
    for (int m = 0; m < d->num_meshes; m++) {
        const pt_mesh& me = d->meshes[m];
        if (me.num_vertices <= 0 || me.num_faces <= 0 || !me.positions || !me.indices)
            return fail(PT_ERR_BAD_SCENE, "empty mesh");
        if (!me.normals) return fail(PT_ERR_BAD_SCENE, "mesh without vertex normals (required, SURVEY H5a)");
        if (me.material_id < 0 || me.material_id >= d->num_materials) return fail(PT_ERR_BAD_SCENE, "mesh material id out of range");
    }

    // ---- primitives
    using clk = std::chrono::steady_clock;
    auto us_since = [](clk::time_point t0) { return (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count(); };
    const clk::time_point t_all = clk::now();
    clk::time_point t_stage = t_all;
    const int N = d->num_shapes;
    const bool on_device = N >= kDeviceSweepMin && !host_sweep_forced();     // big scenes are prepared on the device (see below)
    std::vector<DPrim> prims(on_device ? 0 : N);
    std::vector<DNormals> normals(on_device ? 0 : N);
    S->tri_only = true;
    if (on_device) {
        int prc, has_sphere = 0;
        if ((prc = S->prims.ensure((size_t)N)) || (prc = S->normals.ensure((size_t)N))) return prc;
        if ((prc = ptp::prims_device(d, S->prims.p, S->normals.p, &has_sphere))) return prc;
        S->tri_only = !has_sphere;
    } else {
    std::memset(prims.data(), 0, sizeof(DPrim) * N);
    std::memset(normals.data(), 0, sizeof(DNormals) * N);
    for (int i = 0; i < N; i++) {
        const pt_shape& s = d->shapes[i];
        DPrim& p = prims[i];
        if (s.type == PT_SHAPE_SPHERE) {
            if (s.material_id < 0 || s.material_id >= d->num_materials) return fail(PT_ERR_BAD_SCENE, "sphere material id out of range");
            p.v[0] = s.center[0]; p.v[1] = s.center[1]; p.v[2] = s.center[2]; p.v[3] = s.radius;
            p.info = (int32_t)(0x80000000u | (uint32_t)s.material_id);
            p.light = s.area_light_id;
            S->tri_only = false;
        } else if (s.type == PT_SHAPE_TRIANGLE) {
            if (s.mesh_index < 0 || s.mesh_index >= d->num_meshes) return fail(PT_ERR_BAD_SCENE, "triangle mesh index out of range");
            const pt_mesh& me = d->meshes[s.mesh_index];
            if (s.face_index < 0 || s.face_index >= me.num_faces) return fail(PT_ERR_BAD_SCENE, "triangle face index out of range");
            const int32_t* idx = me.indices + 3 * (size_t)s.face_index;
            for (int k = 0; k < 3; k++) {
                if (idx[k] < 0 || idx[k] >= me.num_vertices) return fail(PT_ERR_BAD_SCENE, "vertex index out of range");
                for (int c = 0; c < 3; c++) {
                    p.v[3 * k + c] = me.positions[3 * (size_t)idx[k] + c];
                    normals[i].n[3 * k + c] = me.normals[3 * (size_t)idx[k] + c];
                }
            }
            p.info = me.material_id;
            p.light = me.area_light_id;
        } else {
            return fail(PT_ERR_BAD_SCENE, "unknown shape type");
        }
    }
    }

    S->diffuse_only = true;
    for (int m = 0; m < d->num_materials; m++)
        if (d->materials[m].type != PT_MAT_DIFFUSE) { S->diffuse_only = false; break; }

    S->create_us[1] = us_since(t_stage); t_stage = clk::now();
    // ---- BVH: the caller's tree (validated), and the internal tree over the same leaf boxes
    // From kDeviceSweepMin primitives up all of it happens on the device (pt_scene_prep.hip, pt_sweep_build.hip): the caller's
    // pool goes up once, is checked and re-laid there, its leaf boxes feed the sweep builder, the builder's pool is re-laid in
    // turn — nothing of either tree comes back to the host.  The host code below serves small scenes, PT_SWEEP_BUILD=host, and
    // whatever the device path hands back (input that runs into the sweep builder's depth guard).
    TreeHost ref;
    TreeHost fast;
    bool have_fast = false;
    bool trees_on_device = false;              // tree[0] (and tree[1] when have_fast) already sit in S->tree[]
    int ref_depth = 0;
    size_t ref_inner = 0;
    DevBuf<pt_bvh_node> pool_dev;              // the caller's pool and the map pool index -> DNode index (tie tables)
    DevBuf<int32_t> iop_dev;
    S->sweep_on_device = 0;
    if (on_device) {
        DevBuf<float> boxes_dev;
        ptp::RelayResult r0;
        int prc;
        if ((prc = pool_dev.ensure((size_t)d->num_nodes)) || (prc = iop_dev.ensure((size_t)d->num_nodes)) || (prc = boxes_dev.ensure((size_t)N * 6)) ||
            (prc = S->tree[0].nodes.ensure((size_t)N - 1)))
            return prc;
        HIP_TRY(hipMemcpy(pool_dev.p, d->nodes, (size_t)d->num_nodes * sizeof(pt_bvh_node), hipMemcpyHostToDevice));
        prc = ptp::relay_tree_device(pool_dev.p, d->num_nodes, d->root, N, false, S->tree[0].nodes.p, iop_dev.p, boxes_dev.p, kBlock, kTopNodes,
                                     kMaxLdsBudget, &r0);
        if (prc) return prc;                   // PT_ERR_BAD_SCENE with the host path's messages
        {
            pt_scene::Tree& T0 = S->tree[0];
            T0.have_oct = false; T0.num_nodes = N - 1; T0.root_ref = 0; T0.stack_cap = r0.stack_need + 1; T0.top_avail = r0.top_avail; T0.depth = r0.depth;
        }
        ref_depth = r0.depth;
        ref_inner = (size_t)N - 1;
        trees_on_device = true;
        S->create_us[2] = us_since(t_stage); t_stage = clk::now();
        if (r0.nested) {
            // (why a tree of the library's own may stand in for a nested caller's tree: see the host path below)
            DevBuf<pt_bvh_node> fpool_dev;
            DevBuf<int32_t> fiop_dev;
            int32_t fdepth = 0;
            double dev_ms = 0;
            bool ok = !fpool_dev.ensure((size_t)d->num_nodes) && !fiop_dev.ensure((size_t)d->num_nodes) && !S->tree[1].nodes.ensure((size_t)N - 1) &&
                      boxes_finite_device(boxes_dev.p, (size_t)N * 6) &&
                      pts::sweep_build_on_device(boxes_dev.p, N, fpool_dev.p, &fdepth, &dev_ms) == PT_OK;
            S->create_us[3] = us_since(t_stage); t_stage = clk::now();
            ptp::RelayResult r1;
            if (ok && ptp::relay_tree_device(fpool_dev.p, d->num_nodes, 0, N, true, S->tree[1].nodes.p, fiop_dev.p, nullptr, kBlock, kTopNodes,
                                             kMaxLdsBudget, &r1) == PT_OK && r1.nested) {
                pt_scene::Tree& T1 = S->tree[1];
                T1.have_oct = false; T1.num_nodes = N - 1; T1.root_ref = 0; T1.stack_cap = r1.stack_need + 1; T1.top_avail = r1.top_avail; T1.depth = r1.depth;
                have_fast = true;
                S->sweep_on_device = 1;
            } else {
                (void)hipGetLastError();
                S->tree[1].nodes.release();
                {
                    // the device builder handed the input back (depth guard) or failed: the host builder, from the leaf boxes
                    std::vector<float> leaf_boxes((size_t)N * 6);
                    std::vector<pt_bvh_node> fnodes;
                    int32_t froot = 0, hdepth = 0;
                    bool built = hipMemcpy(leaf_boxes.data(), boxes_dev.p, leaf_boxes.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
                    for (size_t k = 0; k < leaf_boxes.size() && built; k++) built = std::isfinite(leaf_boxes[k]);
                    if (built) {
                        try {
                            pts::build_sweep_tree(leaf_boxes.data(), N, fnodes, &froot, &hdepth);
                        } catch (const std::exception&) {
                            built = false;
                        }
                    }
                    bool fnested = true;
                    have_fast = built && convert_tree(fnodes.data(), (int)fnodes.size(), froot, N, fast, nullptr, &fnested, true) == PT_OK && fnested;
                }
            }
        }
    } else {
        std::vector<float> leaf_boxes((size_t)N * 6);
        bool nested = true;
        int trc = convert_tree(d->nodes, d->num_nodes, d->root, N, ref, leaf_boxes.data(), &nested);
        if (trc) return trc;
        ref_depth = ref.depth;
        ref_inner = ref.nodes.size();
        S->create_us[2] = us_since(t_stage); t_stage = clk::now();
        if (N >= kMinInternalTree && nested) {
            // A leaf is tested by the reference iff the ray hits every box on its way down (scene.h:278-297, no pruning).  The slab
            // test is monotone in the box (rounding is), so where every box contains its children's that is: iff it hits the LEAF's
            // own box — whatever the tree above it.  `nested` says the caller's tree is of that kind; then any tree over the same
            // leaf boxes tests the same leaves, and only the ORDER of the tests (ties on t, scene.h:270) still depends on the tree.
            // The tree: all cuts along x, y and z at every node (pt_tree_sweep.h), as deep as it likes — traversed left child
            // first, its stack need is its Strahler number, not its depth (convert_tree).
            bool finite = true;
            for (size_t k = 0; k < leaf_boxes.size() && finite; k++) finite = std::isfinite(leaf_boxes[k]);
            if (finite) {
                std::vector<pt_bvh_node> fnodes;
                int32_t froot = 0, fdepth = 0;
                bool built = true;
                try {
                    pts::build_sweep_tree(leaf_boxes.data(), N, fnodes, &froot, &fdepth);
                } catch (const std::exception&) {        // out of host memory: the caller's tree serves alone
                    built = false;
                }
                S->create_us[3] = us_since(t_stage); t_stage = clk::now();
                bool fnested = true;
                have_fast = built && convert_tree(fnodes.data(), (int)fnodes.size(), froot, N, fast, nullptr, &fnested, true) == PT_OK && fnested;
            }
        }
    }
    S->create_us[4] = us_since(t_stage); t_stage = clk::now();
    // trees the device path made are in S->tree[] already; what the host made is uploaded below
    const TreeHost* hosts[2] = {trees_on_device ? nullptr : &ref, (have_fast && !S->sweep_on_device) ? &fast : nullptr};
    std::vector<DMaterial> mats(d->num_materials);
    for (int m = 0; m < d->num_materials; m++) {
        const pt_material& src = d->materials[m];
        mats[m] = DMaterial{src.type, src.reflectance[0], src.reflectance[1], src.reflectance[2], src.eta, src.exponent, 0.0f, 0.0f};
    }
    std::vector<DEmission> emis(std::max(d->num_lights, 1));
    std::memset(emis.data(), 0, sizeof(DEmission) * emis.size());
    for (int l = 0; l < d->num_lights; l++) {
        const pt_light& src = d->lights[l];
        emis[l] = DEmission{src.radiance[0], src.radiance[1], src.radiance[2], src.type == PT_LIGHT_DIFFUSE_AREA ? 1 : 0};
    }
    std::vector<DLight> dlights(std::max(d->num_lights, 1));
    std::memset(dlights.data(), 0, sizeof(DLight) * dlights.size());
    for (int l = 0; l < d->num_lights; l++) {
        const pt_light& src = d->lights[l];
        dlights[l] = DLight{src.radiance[0], src.radiance[1], src.radiance[2], src.type == PT_LIGHT_DIFFUSE_AREA ? 1 : 0,
                            src.position[0], src.position[1], src.position[2], src.shape_id};
        if (src.type == PT_LIGHT_DIFFUSE_AREA && (src.shape_id < 0 || src.shape_id >= N))
            return fail(PT_ERR_BAD_SCENE, "area light refers to a shape that does not exist");
    }
    int rc;
    auto upload_tree = [&](int t, const TreeHost& H) -> int {
        const std::vector<DNode>& nodes = H.nodes;
        pt_scene::Tree& T = S->tree[t];
        // 8 ray-octant copies of the node table for LDS-resident scenes: octant bit k set <=> 1/d[k] < 0, in which case
        // Hit() swaps the two slab distances of axis k (bbox.cuh:40-55); here the two planes are swapped instead.
        T.have_oct = false;
        if (nodes.size() * 8 * sizeof(DNode) <= kOctNodeLimit) {
            std::vector<DNode> nodes_oct(nodes.size() * 8);
            for (int o = 0; o < 8; o++)
                for (size_t k = 0; k < nodes.size(); k++) {
                    DNode n = nodes[k];
                    for (int ax = 0; ax < 3; ax++)
                        if (o & (1 << ax)) { std::swap(n.lmin[ax], n.lmax[ax]); std::swap(n.rmin[ax], n.rmax[ax]); }
                    nodes_oct[(size_t)o * nodes.size() + k] = n;
                }
            int r;
            if ((r = T.nodes_oct.ensure(nodes_oct.size()))) return r;
            HIP_TRY(hipMemcpy(T.nodes_oct.p, nodes_oct.data(), nodes_oct.size() * sizeof(DNode), hipMemcpyHostToDevice));
            T.have_oct = true;
        }
        int r;
        if ((r = T.nodes.ensure(nodes.size()))) return r;
        HIP_TRY(hipMemcpy(T.nodes.p, nodes.data(), nodes.size() * sizeof(DNode), hipMemcpyHostToDevice));
        T.num_nodes = (int32_t)nodes.size();
        T.root_ref = H.root_ref;
        T.stack_cap = H.stack_need + 1;                         // + the kDone sentinel at the bottom
        T.top_avail = H.top_avail;
        T.depth = H.depth;
        return PT_OK;
    };
    for (int t = 0; t < 2; t++)
        if (hosts[t] && (rc = upload_tree(t, *hosts[t]))) return rc;
    S->have_fast = have_fast;
    if (!on_device && ((rc = S->prims.ensure(prims.size())) || (rc = S->normals.ensure(normals.size())))) return rc;
    if ((rc = S->materials.ensure(mats.size()))) return rc;
    if ((rc = S->emission.ensure(emis.size()))) return rc;
    if ((rc = S->lights.ensure(dlights.size()))) return rc;
    HIP_TRY(hipMemcpy(S->lights.p, dlights.data(), dlights.size() * sizeof(DLight), hipMemcpyHostToDevice));
    if (!on_device) {
        HIP_TRY(hipMemcpy(S->prims.p, prims.data(), prims.size() * sizeof(DPrim), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(S->normals.p, normals.data(), normals.size() * sizeof(DNormals), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(S->materials.p, mats.data(), mats.size() * sizeof(DMaterial), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(S->emission.p, emis.data(), emis.size() * sizeof(DEmission), hipMemcpyHostToDevice));

    SceneDev& dv = S->dev;
    dv.prims = S->prims.p; dv.normals = S->normals.p;
    dv.materials = S->materials.p; dv.emission = S->emission.p; dv.lights = S->lights.p;
    dv.num_prims = N;
    dv.num_materials = d->num_materials;
    dv.num_emission = d->num_lights;
    dv.bg[0] = d->background[0]; dv.bg[1] = d->background[1]; dv.bg[2] = d->background[2];
    // Keep the internal tree only where it is the better one FOR THIS KIND OF RAY: without pruning a node is visited iff the
    // ray hits its box, and what the boxes of a tree add up to depends on the scene (a soup of large overlapping triangles is
    // better off with the caller's median splits than with cuts of the Morton order; a far-away ground sphere skews any
    // surface-area estimate).  So measure: the same probe rays — from random points on the primitives into uniform
    // directions, which is where and how the segments after the first bounce start — through both trees, inner visits counted.
    S->fast_cost_permille = 0;
    if (have_fast) {
        DevBuf<unsigned long long> visits;
        if ((rc = visits.ensure(2))) return rc;
        HIP_TRY(hipMemset(visits.p, 0, 2 * sizeof(unsigned long long)));
        {
            select_tree(S, 0);
            const SceneDev d0 = S->dev;
            select_tree(S, 1);
            const SceneDev d1 = S->dev;
            const uint32_t lds = (uint32_t)(kBlock / 64) * (uint32_t)std::max(d0.stack_cap, d1.stack_cap) * 64u * 4u;
            hipLaunchKernelGGL(probe_kernel, dim3(kProbeRays / kBlock, 2), dim3(kBlock), lds, nullptr, d0, d1, visits.p);
            HIP_TRY(hipGetLastError());
        }
        unsigned long long v[2] = {0, 0};
        HIP_TRY(hipMemcpy(v, visits.p, sizeof(v), hipMemcpyDeviceToHost));
        S->fast_cost_permille = v[0] ? (int64_t)((1000ull * v[1] + v[0] / 2) / v[0]) : 1000;
        if (!(v[1] * 100ull < v[0] * 90ull)) {              // not at least 10 % fewer boxes: the sweep tree is not worth having
            S->have_fast = false;
            S->tree[1].nodes.release();
            S->tree[1].nodes_oct.release();
            S->tree[1].have_oct = false;
            // ... but the way the internal tree is TRAVERSED still is, for scenes in global memory: left child first with the
            // children ordered for a short stack, leaves set aside, ties settled in place.  So the caller's own topology becomes the
            // internal tree (a caller who hands in a tree as good as the sweep tree: 5.43 -> 4.9 ms on bunny,
            // profiles/r02_device_bvh_build.log).
            const bool global_scene = (size_t)N * (sizeof(DPrim) + sizeof(DNormals)) + ref_inner * sizeof(DNode) > kLdsSceneLimit;
            if (global_scene) {
                TreeHost own;
                bool own_nested = true;
                if (convert_tree(d->nodes, d->num_nodes, d->root, N, own, nullptr, &own_nested, true) == PT_OK && own_nested) {
                    if ((rc = upload_tree(1, own))) return rc;
                    S->have_fast = true;
                    S->fast_is_callers_topology = true;
                }
            }
        }
    }
    S->create_us[5] = us_since(t_stage); t_stage = clk::now();
    if (S->have_fast) {
        // what settles ties on t in the caller's visit order: every primitive's path from the caller's root (turn bits) and
        // the inner nodes along it.  (Depth <= 64 levels of leaves and inner nodes: at most 63 turns.)
        const int levels = std::max(ref_depth - 1, 1);
        const size_t anc_words = (size_t)N * (size_t)levels;
        if (S->ref_path.ensure((size_t)N) || S->ref_anc.ensure(anc_words)) {
            // no room for the tables (num_shapes x depth words): the scene simply keeps to the caller's tree
            (void)hipGetLastError();
            S->have_fast = false;
            S->fast_is_callers_topology = false;
            S->ref_path.release(); S->ref_anc.release();
            S->tree[1].nodes.release(); S->tree[1].nodes_oct.release(); S->tree[1].have_oct = false;
        } else {
            bool made = false;
            if (trees_on_device) {
                // on the device: every leaf of the caller's pool (uploaded above) walks to the root (pt_scene_prep.hip) — a
                // depth-first walk over a million leaves writing a 100 MB table is 150 ms of host time, and the table would
                // have to be uploaded
                DevBuf<int32_t> parent_dev;
                if (!parent_dev.ensure((size_t)d->num_nodes) && hipMemset(S->ref_anc.p, 0, anc_words * sizeof(int32_t)) == hipSuccess)
                    made = ptp::tie_tables_device(pool_dev.p, d->num_nodes, d->root, iop_dev.p, N, levels, parent_dev.p, S->ref_path.p, S->ref_anc.p) == PT_OK;
                if (!made) return fail(PT_ERR_DEVICE, "tie tables on the device failed");
            }
            if (!made) {
                std::vector<unsigned long long> path(N, 0ull);
                std::vector<int32_t> anc(anc_words, 0);
                struct Walk { int32_t node; int32_t level; unsigned long long turns; };
                std::vector<Walk> todo;
                std::vector<int32_t> trail(levels, 0);
                todo.push_back({d->root, 0, 0ull});
                while (!todo.empty()) {
                    const Walk w = todo.back();
                    todo.pop_back();
                    const pt_bvh_node& nd = d->nodes[w.node];
                    if (nd.prim != -1) {
                        path[nd.prim] = w.turns;
                        std::memcpy(&anc[(size_t)nd.prim * levels], trail.data(), (size_t)w.level * sizeof(int32_t));
                        continue;
                    }
                    trail[w.level] = ref.inner_of_pool[w.node];
                    // depth-first, left subtree first: `trail` below w.level is still this node's path when its right child is taken up
                    todo.push_back({nd.right, w.level + 1, w.turns | (1ull << w.level)});
                    todo.push_back({nd.left, w.level + 1, w.turns});
                }
                HIP_TRY(hipMemcpy(S->ref_path.p, path.data(), path.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(S->ref_anc.p, anc.data(), anc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            }
            S->ref_levels = levels;
        }
    }
    // everything the scene holds is complete from here on: trace kernels run on streams of the handle's own (FrameSlot) that do
    // not synchronise with the stream the preparation kernels ran on
    HIP_TRY(hipDeviceSynchronize());
    S->create_us[6] = us_since(t_stage);
    S->create_us[0] = us_since(t_all);
    select_tree(S, 0);
    S->scene_bytes = (uint32_t)std::min<size_t>(
        ref_inner * sizeof(DNode) + (size_t)N * (sizeof(DPrim) + sizeof(DNormals)) + mats.size() * sizeof(DMaterial) +
            emis.size() * sizeof(DEmission) + 64, 0xffffffffu);
    return PT_OK;
}

// res: 0 scene in global memory, 1 scene staged in LDS, 2 staged in LDS with the 8 octant node tables,
//      3 scene in global memory with the top of the tree cached in LDS
LdsPlan make_plan(const pt_scene* S, int res, bool stack16, int which = 0) {
    LdsPlan lp{};
    const pt_scene::Tree& T = S->tree[which];
    uint32_t off = 0;
    if (res == 3) {
        // as many top nodes as fit next to the stacks within the block's LDS budget (default: without costing a resident block)
        const uint32_t stack_bytes = (uint32_t)(kBlock / 64) * (uint32_t)T.stack_cap * 64u * 4u;
        const uint32_t budget = S->opt_lds_budget_kb > 0 ? (uint32_t)S->opt_lds_budget_kb * 1024u : kTopLdsBudget;
        const uint32_t room = budget > stack_bytes ? budget - stack_bytes : 0u;
        lp.top_count = std::min<uint32_t>(T.top_avail, room / (uint32_t)sizeof(DNode));
        lp.nodes_off = 0;
        off = lp.top_count * (uint32_t)sizeof(DNode);
    } else if (res != 0) {
        lp.nodes_off = off;
        off = align16(off + (res == 2 ? 8u * oct_table_pitch((uint32_t)T.num_nodes, kLdsNodeStride)
                                      : (uint32_t)T.num_nodes * kLdsNodeStride));
        lp.prims_off = off; off = align16(off + (uint32_t)S->dev.num_prims * sizeof(DPrim));
        lp.normals_off = off; off = align16(off + (uint32_t)S->dev.num_prims * sizeof(DNormals));
        lp.mats_off = off; off = align16(off + (uint32_t)S->dev.num_materials * sizeof(DMaterial));
        lp.emis_off = off; off = align16(off + (uint32_t)std::max(S->dev.num_emission, 1) * sizeof(DEmission));
    }
    lp.stack_off = off;
    off += (uint32_t)(kBlock / 64) * (uint32_t)T.stack_cap * 64u * (stack16 ? 2u : 4u);
    lp.total = align16(off);
    return lp;
}

using TraceFn = void (*)(SceneDev, RenderDev, LdsPlan, float4*, uint32_t*, unsigned long long*);

TraceFn pick_kernel_v1(bool lds, bool prune, bool stats) {
    if (lds) {
        if (prune) return stats ? trace_kernel<true, true, true> : trace_kernel<true, true, false>;
        return stats ? trace_kernel<true, false, true> : trace_kernel<true, false, false>;
    }
    if (prune) return stats ? trace_kernel<false, true, true> : trace_kernel<false, true, false>;
    return stats ? trace_kernel<false, false, true> : trace_kernel<false, false, false>;
}

template <int RES, int THRESH, int INNER, int MINW, int SPEC>
TraceFn pick_v2_rt(bool prune, bool stats) {
    if (prune) return stats ? trace_kernel_v2<RES, true, true, THRESH, INNER, MINW, SPEC> : trace_kernel_v2<RES, true, false, THRESH, INNER, MINW, SPEC>;
    return stats ? trace_kernel_v2<RES, false, true, THRESH, INNER, MINW, SPEC> : trace_kernel_v2<RES, false, false, THRESH, INNER, MINW, SPEC>;
}

template <int RES, int THRESH, int INNER, int MINW>
TraceFn pick_v2_r(bool prune, bool stats, int spec) {
    if (spec == 2) return pick_v2_rt<RES, THRESH, INNER, MINW, 2>(prune, stats);
    if (spec == 1) return pick_v2_rt<RES, THRESH, INNER, MINW, 1>(prune, stats);
    return pick_v2_rt<RES, THRESH, INNER, MINW, 0>(prune, stats);
}

template <int THRESH, int INNER, int MINW>
TraceFn pick_v2_ti(int res, bool prune, bool stats, int spec) {
    if (res == 3) return pick_v2_r<3, THRESH, INNER, MINW>(prune, stats, spec);
    if (res == 2) return pick_v2_r<2, THRESH, INNER, MINW>(prune, stats, spec);
    if (res == 1) return pick_v2_r<1, THRESH, INNER, MINW>(prune, stats, spec);
    return pick_v2_r<0, THRESH, INNER, MINW>(prune, stats, spec);
}

// (thresh, inner, min-waves-per-SIMD) variants compiled in; inner < 0 selects the "vote" burst of -inner steps, 1..9 a
// burst of `inner` inner steps + 1 leaf step, >= 100 the encoded burst REPS*100 + N_INNER*10 + N_LEAF (pt_kernels.h).
// Measured on MI355X (tests/tools/gpu_tune.py, tests/tools/gpu_ab.py, profiles/r01_tune_round*.log): LDS-resident scenes are
// fastest with T40 / 6 inner + 2 leaf steps / W6 (cbox 3.77 ms against 3.99 with 3+1, 4.10 with 4+1, 3.92 with two rounds of
// 3+1; sphere and mixed scenes 1-6 % ahead of the vote burst they used before; r01_tune_round21/22), scenes in global memory with
// T32 / I4 / W6 (bunny 10.7 ms; every other burst shape within 1 %).  I8 and unbounded descent are slower, T56 starves the
// scheduler phase, W6 (<= 80 VGPRs -> 6 waves/SIMD) beats the unconstrained 82-VGPR build by 3-5 %, W8 (64 VGPRs, spills)
// is 5-8 % slower.  Re-checked on the internal tree (fewer inner visits per leaf test): 6 + 2 still wins on LDS-resident scenes
// (cbox: 5+2 +2.4 %, 4+2 +4.6 %, 3+2 +5.4 %, two rounds of 2+1 +13 %; thresholds 24 / 32 / 48 / 56: +1.7 / 0.0 / +1.2 / +7.5 %; profiles/r02_tune_round44_lds_bursts_rejected.log,
// r02_tune_round46_thresh_rejected.log).
TraceFn pick_kernel_v2(int res, bool prune, bool stats, int spec, int thresh, int inner, int minw) {
#define PT_V2(T, I, W) if (thresh == T && inner == I && minw == W) return pick_v2_ti<T, I, W>(res, prune, stats, spec);
    PT_V2(40, -6, 6) PT_V2(32, 4, 6) PT_V2(40, 4, 6) PT_V2(40, 3, 6) PT_V2(40, 162, 6)
    PT_V2(32, 1004, 6) PT_V2(32, 1231, 6)
    PT_V2(32, 1231, 5) PT_V2(40, 162, 5)        // 96 VGPRs, 5 waves per SIMD: for the instantiations that spill under the 80-VGPR cap
#undef PT_V2
    return nullptr;
}

// Kernels with next-event estimation (PT_RENDER_NEE): exact traversal, the default schedule of the residency, generic
// scene content or triangles-with-diffuse-materials only.
template <int RES, int THRESH, int INNER>
TraceFn pick_nee_r(bool stats, int spec) {
    if (spec == 2) return stats ? trace_kernel_v2<RES, false, true, THRESH, INNER, 6, 2, true> : trace_kernel_v2<RES, false, false, THRESH, INNER, 6, 2, true>;
    return stats ? trace_kernel_v2<RES, false, true, THRESH, INNER, 6, 0, true> : trace_kernel_v2<RES, false, false, THRESH, INNER, 6, 0, true>;
}
TraceFn pick_kernel_nee(int res, bool stats, int spec) {
    if (res == 3) return pick_nee_r<3, 32, 4>(stats, spec);
    if (res == 2) return pick_nee_r<2, 40, 162>(stats, spec);
    if (res == 1) return pick_nee_r<1, 40, 162>(stats, spec);
    return pick_nee_r<0, 32, 4>(stats, spec);
}

// trace_kernel_q (option "kernel" = 3): LDS-resident scenes, exact traversal, no next-event estimation.
using TraceFnQ = void (*)(SceneDev, RenderDev, LdsPlan, QParams, float4*, uint32_t*, unsigned long long*);
template <int RES, bool POSTPONE>
TraceFnQ pick_q_r(bool stats, int spec) {
    if (spec == 2) return stats ? trace_kernel_q<RES, true, 2, POSTPONE> : trace_kernel_q<RES, false, 2, POSTPONE>;
    if (spec == 1) return stats ? trace_kernel_q<RES, true, 1, POSTPONE> : trace_kernel_q<RES, false, 1, POSTPONE>;
    return stats ? trace_kernel_q<RES, true, 0, POSTPONE> : trace_kernel_q<RES, false, 0, POSTPONE>;
}
// internal_tree: scenes in global memory set leaves aside (order-free leaf tests need the internal tree's tie handling)
TraceFnQ pick_kernel_q(const pt_scene* S, int res, bool internal_tree) {
    const int spec = !(S->tri_only && S->opt_specialize) ? 0 : (S->diffuse_only ? 2 : 1);
    const bool stats = S->opt_stats != 0;
    if (res == 2) return pick_q_r<2, false>(stats, spec);
    if (res == 1) return pick_q_r<1, false>(stats, spec);
    if (res == 3) return internal_tree ? pick_q_r<3, true>(stats, spec) : pick_q_r<3, false>(stats, spec);
    return internal_tree ? pick_q_r<0, true>(stats, spec) : pick_q_r<0, false>(stats, spec);
}

// LDS plan of trace_kernel_q: the staged scene as for v2, 16-bit traversal stacks for the kQT traversal waves only, then the
// control words and the two rings (contiguous: the kernel clears them in one sweep).  Also settles the schedule knobs.
QParams make_plan_q(const pt_scene* S, LdsPlan& lp, int which, int res) {
    QParams q{};
    const bool lds_scene = res == 1 || res == 2;
    const uint32_t stack_bytes = (uint32_t)kQT * (uint32_t)S->tree[which].stack_cap * 64u * (lds_scene ? 2u : 4u);
    // scenes in global memory carry 32-bit stacks as deep as the tree's Strahler number: they run with rings of half the size, and
    // the top of the tree takes what two workgroups per CU leave (160 KB / 2, minus stacks and rings)
    uint32_t ring = kQRing;
    if (!lds_scene) ring = std::max<uint32_t>(64u, kQRing / 2);
    q.ring_log2 = 0;
    while ((1u << q.ring_log2) < ring) q.ring_log2++;
    const uint32_t ring_bytes = kQCtlBytes + 2u * ring * kQEntryBytes;
    if (res == 3) {
        const uint32_t budget = S->opt_lds_budget_kb > 0 ? (uint32_t)S->opt_lds_budget_kb * 1024u : 80u * 1024u;
        const uint32_t used = stack_bytes + ring_bytes + 64u;
        lp.top_count = std::min<uint32_t>(S->tree[which].top_avail, budget > used ? (budget - used) / (uint32_t)sizeof(DNode) : 0u);
        lp.nodes_off = 0;
        lp.stack_off = lp.top_count * (uint32_t)sizeof(DNode);
    }
    uint32_t off = align16(lp.stack_off + stack_bytes);
    q.ctl_off = off; off += kQCtlBytes;
    q.shade_off = off; off += ring * kQEntryBytes;
    q.ready_off = off; off += ring * kQEntryBytes;
    lp.total = align16(off);
    // Paths in flight per workgroup.  The bound that keeps the rings from filling up for good (pt_kernel_q.h): with every
    // T-lane holding a finished path, a ring that cannot take a wave's worth more (> ring - 64 entries each) — that many
    // paths must not exist:  target <= T-lanes + 2 x ring - 128.
    const int32_t t_lanes = kQT * 64, cap = t_lanes + 2 * (int32_t)ring - 128;
    int32_t target = S->opt_q_target > 0 ? (int32_t)S->opt_q_target : t_lanes + (int32_t)ring;
    q.target = std::max(64, std::min(target, cap));
    q.swap = S->opt_q_swap > 0 ? (int32_t)std::min<int64_t>(S->opt_q_swap, 64) : 16;
    q.low = S->opt_q_low > 0 ? (int32_t)S->opt_q_low : 8;
    return q;
}

// Residency the next launch will use (see make_plan).
// The tree a render traverses: the internal tree on the default kernel where scene creation kept one (see launch_render), the
// caller's tree otherwise.  Pruned traversal (opt-in, tolerance semantics: DESIGN.md §6) takes the internal tree as well — left
// child first like the exact one, so that its short stacks hold; bunny 7.55 -> 3.89 ms, no pixel moved.
int which_tree(const pt_scene* S) {
    return (S->have_fast && S->opt_fast_tree && S->opt_kernel >= 2) ? 1 : 0;
}

int scene_residency(const pt_scene* S, int which = 0) {
    const pt_scene::Tree& T = S->tree[which];
    if (S->opt_force_global || S->scene_bytes > kLdsSceneLimit)
        return (T.top_avail > 0 && S->opt_top_cache && S->opt_kernel >= 2 && !S->opt_force_global) ? 3 : 0;
    if (S->opt_kernel >= 2 && S->opt_octants && T.have_oct) return 2;
    return 1;
}

TraceFn pick_kernel(const pt_scene* S, int res, bool prune, bool stats, bool internal_tree) {
    if (S->opt_kernel >= 2) {
        int t = (int)S->opt_v2_thresh, i = (int)S->opt_v2_inner, w = (int)S->opt_v2_minw;
        const bool lds = res == 1 || res == 2;
        if (t == 0) t = lds ? 40 : 32;
        const bool tri = S->tri_only && S->opt_specialize;
        // LDS-resident: 6 inner + 2 leaf steps; global memory: 4 + 1 on the caller's tree; on the internal tree two rounds of
        // 3 + 1 with leaves set aside (v2_inner 1000 + burst).  Setting leaves aside: bunny -2.2 %, dragon stand-in -3.4 %
        // against the plain 4 + 1 (LDS-resident scenes lose 9 % with it: profiles/r02_tune_round43_postponed_leaves.log); two
        // rounds of 3 + 1 instead of one of 4 + 1: teapot -8.5 %, bunny -1.2 %, buddha stand-in -1.5 %, dragon stand-in -1.8 %
        // (r02_tune_round48_global_thresh.log, r02_tune_round49_global_burst.log; thresholds 24 / 40 / 48 lose, so do three rounds of
        // 3 + 1 and two of 2 + 1, and every set-aside shape on LDS-resident scenes: r02_tune_round50_more_bursts_rejected.log; 7 / 8 waves per
        // SIMD by register cap, with the blocks to match: +11..16 % / +28..41 % from the spills, r02_tune_round51_more_waves_rejected.log).
        if (i == 0) i = lds ? 162 : (internal_tree ? 1231 : 4);
        if (i >= 1000 && !internal_tree) return nullptr;        // order-free leaf tests need the internal tree's tie handling
        if (w == 0) w = 6;
        const int spec = !tri ? 0 : (S->diffuse_only ? 2 : 1);
        return pick_kernel_v2(res, prune, stats, spec, t, i, w);
    }
    return pick_kernel_v1(res == 1 || res == 2, prune, stats);
}

// Sums the kCounterSlots per-workgroup counter slots of the last frame (the caller has synchronised the stream).
int read_slot_sums(const pt_scene* S, unsigned long long* out) {
    std::vector<unsigned long long> raw(kTimelineBase);
    HIP_TRY(hipMemcpy(raw.data(), S->counters(), raw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::fill(out, out + kSlotStride, 0ull);
    for (int s = 0; s < kCounterSlots; s++)
        for (int k = 0; k < kSlotStride; k++) out[k] += raw[(size_t)s * kSlotStride + k];
    return PT_OK;
}

// Blocks of a trace launch: every resident slot of the device, or fewer when there is not enough work to fill them.
int launch_grid(int num_cus, int blocks_per_cu, uint64_t work_items, int block_threads = kBlock) {
    const uint64_t blocks_needed = (work_items + (uint64_t)block_threads - 1) / (uint64_t)block_threads;
    return (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)num_cus * (uint64_t)blocks_per_cu, blocks_needed));
}
// Lanes (= global-memory stack columns of the reference-order reruns) of that launch.
size_t redo_stack_lanes(int num_cus, int blocks_per_cu, uint64_t work_items) {
    return (size_t)launch_grid(num_cus, blocks_per_cu, work_items) * kBlock;
}

struct RowSel {
    int begin, step, count;
};

int select_rows(const pt_render_params* p, RowSel* out) {
    int rb = p->row_begin, re = p->row_end;
    if (rb == 0 && re == 0) re = p->height;
    int step = p->row_stride > 1 ? p->row_stride : 1;
    if (rb < 0 || re > p->height || rb > re) return fail(PT_ERR_INVALID_ARG, "row range outside the image");
    int n = 0;
    for (int j = rb; j < re; j += step) n++;
    *out = RowSel{rb, step, n};
    return PT_OK;
}

// mode: 0 = pt_render (fb = mean), 2 = pt_render_accumulate (accum (+)= sum)
int launch_render(pt_scene* S, const pt_render_params* p, float* out_dev, int mode, hipStream_t stream) {
    if (!S || !p || !out_dev) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (p->width <= 0 || p->height <= 0 || p->spp <= 0) return fail(PT_ERR_INVALID_ARG, "width, height and spp must be positive");
    if (p->sample_offset < 0 || p->stream_stride < 0) return fail(PT_ERR_INVALID_ARG, "negative sample_offset / stream_stride");
    // PCG stream = pixel_index*stride + sample_offset + s: samples [sample_offset, sample_offset + spp) must stay below the
    // stride, or pixel p's sample `stride + k` would replay pixel p+1's sample k (identical random sequences in neighbouring
    // pixels).  With the default stride (0 -> spp) that allows sample_offset 0 only: progressive callers pass an explicit
    // stride = any upper bound on the total sample count.
    if ((int64_t)p->sample_offset + p->spp > (int64_t)(p->stream_stride > 0 ? p->stream_stride : p->spp))
        return fail(PT_ERR_INVALID_ARG, p->stream_stride > 0
                        ? "sample_offset + spp exceeds stream_stride: PCG streams of neighbouring pixels would overlap"
                        : "sample_offset > 0 needs an explicit stream_stride >= total sample count (default stride = spp)");
    RowSel rows;
    int rc = select_rows(p, &rows);
    if (rc) return rc;
    DeviceGuard guard;
    { int grc = guard.enter(S->device); if (grc) return grc; }
    // this call's frame slot (see pt_scene::FrameSlot)
    const int in_flight = (int)std::min<int64_t>(kMaxFramesInFlight, std::max<int64_t>(1, S->opt_frames_in_flight));
    const bool timing = S->opt_timing_frames > 0;        // 0: no HIP events around the kernels (four records less per frame)
    // Is the previous call's frame still on the GPU?  If not there is nothing to overlap with, and this frame runs on the
    // caller's stream as with one slot: a caller that synchronises after every frame (a display loop) does not pay for two
    // event waits across streams per frame.
    bool prev_busy = false;
    if (in_flight > 1 && S->frame_seq > 0 && S->cur().used) {
        prev_busy = hipEventQuery(S->cur().free_ev) == hipErrorNotReady;
        (void)hipGetLastError();
    }
    S->cur_slot = (int)(S->frame_seq % (uint64_t)in_flight);
    pt_scene::FrameSlot& slot = S->cur();
    if ((rc = slot.ctl.ensure(kWorkWords + kNumCounters))) return rc;
    if (!slot.free_ev) HIP_TRY(hipEventCreateWithFlags(&slot.free_ev, hipEventDisableTiming));
    S->last_stream = stream;
    S->have_timing = false;
    S->info_passes = 0;
    {   // this call's slot of the timing ring
        const size_t ring = (size_t)std::max<int64_t>(1, S->opt_timing_frames);
        if (S->frames.size() != ring) { S->drop_events(); S->frames.resize(ring); }
    }
    pt_scene::FrameRec& frec = S->frames[S->frame_seq % S->frames.size()];
    frec.passes = 0;
    S->frame_seq++;
    if (rows.count == 0) {           // nothing to trace: the frame's counters read zero
        if (slot.used && slot.free_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, slot.free_ev, 0));
        HIP_TRY(hipMemsetAsync(slot.ctl.p, 0, (kWorkWords + kTimelineBase) * sizeof(unsigned long long), stream));
        HIP_TRY(hipEventRecord(slot.free_ev, stream));
        slot.used = true; slot.free_stream = stream;
        return PT_OK;
    }

    const uint64_t npix = (uint64_t)rows.count * (uint64_t)p->width;
    if (npix > (1ull << 30)) return fail(PT_ERR_INVALID_ARG, "more than 2^30 pixels per call");
    // samples per pass: bounded by the scratch budget and by 2^30 work items per launch
    uint64_t scratch = S->opt_scratch_bytes > 0 ? (uint64_t)S->opt_scratch_bytes : kDefaultScratchBytes;
    if (S->opt_scratch_bytes <= 0 && std::min<uint64_t>(scratch, npix * (uint64_t)p->spp * sizeof(float4)) > slot.samples.n * sizeof(float4)) {
        // the buffer must grow under the default budget: never to more than a quarter of what is free on the device right
        // now, so that several handles / ranks on one GPU, or a smaller GPU, split into more passes instead of failing
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            scratch = std::min<uint64_t>(scratch, std::max<uint64_t>((uint64_t)free_b / 4, slot.samples.n * sizeof(float4)));
    }
    uint64_t spp_pass = std::max<uint64_t>(1, std::min<uint64_t>(scratch / (npix * sizeof(float4)), (1ull << 30) / npix));
    spp_pass = std::min<uint64_t>(spp_pass, (uint64_t)p->spp);
    if (mode == 2 && spp_pass < (uint64_t)p->spp)
        return fail(PT_ERR_UNSUPPORTED, "pt_render_accumulate: spp of one call must fit the scratch budget (single pass)");
    // allocation failure (another handle took the memory meanwhile): halve the samples per pass and retry
    while ((rc = slot.samples.ensure(npix * spp_pass))) {
        if (spp_pass == 1 || mode == 2) return rc;
        (void)hipGetLastError();
        spp_pass = (spp_pass + 1) / 2;
    }
    const int n_pass = (int)(((uint64_t)p->spp + spp_pass - 1) / spp_pass);
    if (mode == 2 && n_pass > 1)
        return fail(PT_ERR_UNSUPPORTED, "pt_render_accumulate: spp of one call must fit the scratch budget (single pass)");
    if (n_pass > 1 && (rc = S->accum.ensure(npix * 3))) return rc;

    const int traversal = p->traversal == PT_TRAVERSAL_DEFAULT ? PT_TRAVERSAL_EXACT : p->traversal;
    if (traversal != PT_TRAVERSAL_EXACT && traversal != PT_TRAVERSAL_PRUNED) return fail(PT_ERR_INVALID_ARG, "unknown traversal mode");
    // Exact traversal runs on the internal tree where scene creation kept one.  What depends on the visit order is handled the
    // reference's way: two valid hits with equal t (the first one VISITED wins, scene.h:270) are ordered by one box test in
    // the caller's tree (pt_trace.h: ref_visits_first); rays with a zero direction component (1/d infinite: the monotonicity
    // argument of validate_and_build does not cover 0 * inf) are traced on the caller's tree in reference order.
    const int which = which_tree(S);
    const int res = scene_residency(S, which);
    const bool lds_scene = res == 1 || res == 2;
    LdsPlan lp = make_plan(S, res, lds_scene && S->opt_kernel >= 2, which);   // trace_kernel_v2 keeps 16-bit stacks for LDS scenes
    if (lp.total > S->lds_per_block_max) return fail(PT_ERR_DEVICE, "LDS plan exceeds the per-block limit");
    if (p->flags & ~PT_RENDER_NEE) return fail(PT_ERR_INVALID_ARG, "unknown bits in pt_render_params.flags");
    const bool nee = (p->flags & PT_RENDER_NEE) != 0;
    if (nee && (S->opt_kernel < 2 || traversal != PT_TRAVERSAL_EXACT))
        return fail(PT_ERR_UNSUPPORTED, "PT_RENDER_NEE runs on the default kernel with exact traversal only");
    // kernel 3 (paths regrouped across the waves of a workgroup) serves exact traversal without next-event estimation; those run
    // on kernel 2
    const bool use_q = S->opt_kernel == 3 && !nee && traversal == PT_TRAVERSAL_EXACT;
    QParams qp{};
    TraceFnQ fnq = nullptr;
    TraceFn fn = nullptr;
    if (use_q) {
        qp = make_plan_q(S, lp, which, res);
        if (lp.total > S->lds_per_block_max) return fail(PT_ERR_DEVICE, "LDS plan exceeds the per-block limit");
        fnq = pick_kernel_q(S, res, which == 1);
    } else {
        fn = nee ? pick_kernel_nee(res, S->opt_stats != 0, (S->tri_only && S->diffuse_only && S->opt_specialize) ? 2 : 0)
                 : pick_kernel(S, res, traversal == PT_TRAVERSAL_PRUNED, S->opt_stats != 0, which == 1);
        if (!fn) return fail(PT_ERR_INVALID_ARG, "no kernel variant compiled for these v2_thresh / v2_inner options");
    }
    const void* fn_any = use_q ? reinterpret_cast<const void*>(fnq) : reinterpret_cast<const void*>(fn);
    const int block_threads = use_q ? kQBlock : kBlock;
    if (S->cfg_fn != fn_any || S->cfg_lds != lp.total) {
        if (lp.total > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(fn_any, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.total));
        int q = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, fn_any, block_threads, lp.total));
        S->cfg_fn = fn_any; S->cfg_lds = lp.total; S->cfg_occ = std::max(q, 1);
    }
    const int occ = S->cfg_occ;
    // scenes read from global memory: 6 blocks per CU (what 80 VGPRs allow) with a 26 KB stack + top-of-tree budget each.
    // With the internal tree's short stacks (its Strahler number, not its depth) the sixth block no longer costs cached nodes
    // worth having: bunny +0.6 %, buddha stand-in -1.1 %, dragon stand-in -7.6 % against 5 blocks of 31 KB
    // (profiles/r02_tune_round41_lds_budget.log, r02_tune_round42_blocks.log).  LDS-resident scenes take every block they can get.
    int bpc = S->opt_blocks_per_cu > 0 ? (int)S->opt_blocks_per_cu : (!lds_scene ? std::min(occ, 6) : occ);
    // Single-pass frames of a handle with more than one slot run on the slot's stream while an earlier frame is on the GPU
    // (pt_scene::FrameSlot).  Such a frame does not take every resident slot of the chip: the waves of a launch run dry
    // together, and while its blocks sit on their last few paths a successor that may only ENTER as they leave starts late.
    // With three slots a frame takes half of each CU — two frames are resident side by side, half a frame apart, the third
    // enters where the first leaves — with two slots all but one block per CU (cbox 2.55 ms per frame with full grids,
    // 2.50 with 5 + 1 of 6, 2.49 with 3 + 3 and three slots; bunny 4.64 / 4.53 / 4.39; profiles/r03_frames_in_flight.log).
    const bool own_stream = in_flight > 1 && n_pass == 1 && prev_busy;
    if (own_stream && S->opt_blocks_per_cu <= 0) bpc = std::max(1, in_flight >= 3 ? bpc / 2 : bpc - 1);
    S->info_kernel = use_q ? 3 : S->opt_kernel == 1 ? 1 : 2;
    S->info_occupancy = occ;
    S->info_blocks_per_cu = bpc;
    S->info_lds_bytes = lp.total;
    S->info_lds_scene = lds_scene;

    if (which == 1) {
        // one global-memory stack column per lane of the LARGEST grid this call launches (pass 0 traces the most samples) for
        // the reruns (rare: latency does not matter).  Sized from the grid, not from an assumed blocks-per-CU: `blocks_per_cu`
        // is a caller's option and the kernel indexes the buffer by blockIdx (pt_kernels.h: redo_stk).
        const size_t lanes = use_q ? (size_t)launch_grid(S->num_cus, bpc, npix * spp_pass, kQBlock) * kQS * 64    // S-waves rerun
                                   : redo_stack_lanes(S->num_cus, bpc, npix * spp_pass);
        if ((rc = slot.redo_stack.ensure(lanes * (size_t)S->tree[0].stack_cap))) return rc;
    }
    select_tree(S, which, which == 1);
    float* accum = mode == 2 ? out_dev : S->accum.p;
    hipStream_t tstream = stream;
    if (own_stream) {
        if (!slot.stream) {
            // A stream of the lowest priority: the runtime maps streams onto a few hardware queues PER PRIORITY (4 by default),
            // and two streams on one hardware queue run in order.  Among the application's own normal-priority streams the two
            // slot streams ended up sharing a queue with each other or with the caller's as often as not (one more stream in
            // the process: 2.70 ms per cbox frame with two slots against 2.69 with one; on queues of their own 2.53 —
            // profiles/r03_frames_in_flight.log).  Low rather than high: a frame waits for the caller's other GPU work (the
            // RCCL gather of the rank's rows, N > 1), not the other way round.  PT_SLOT_STREAM_PRIORITY=normal|high: A/B runs.
            const char* pr = std::getenv("PT_SLOT_STREAM_PRIORITY");
            const std::string want = pr ? pr : "low";
            int least = 0, greatest = 0;
            if (want != "normal" && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
                HIP_TRY(hipStreamCreateWithPriority(&slot.stream, hipStreamNonBlocking, want == "high" ? greatest : least));
            else
                HIP_TRY(hipStreamCreateWithFlags(&slot.stream, hipStreamNonBlocking));
        }
        if (!slot.in_ev) HIP_TRY(hipEventCreateWithFlags(&slot.in_ev, hipEventDisableTiming));
        tstream = slot.stream;
        HIP_TRY(hipEventRecord(slot.in_ev, stream));
    }
    if (slot.used && slot.free_stream != tstream) HIP_TRY(hipStreamWaitEvent(tstream, slot.free_ev, 0));   // the resolve that last read this slot's samples
    // one memset per frame: work counters + the 64 counter slots (+ timeline / histograms when a STATS kernel will run)
    HIP_TRY(hipMemsetAsync(slot.ctl.p, 0, (kWorkWords + (S->opt_stats ? kNumCounters : kTimelineBase)) * sizeof(unsigned long long), tstream));
    for (int pass = 0; pass < n_pass; pass++) {
        const int s0 = pass * (int)spp_pass;
        const int sn = std::min<int>((int)spp_pass, p->spp - s0);
        RenderDev rd{};
        std::memcpy(rd.cam_origin, p->cam_origin, 12);
        std::memcpy(rd.cam_top_left, p->cam_top_left, 12);
        std::memcpy(rd.cam_horizontal, p->cam_horizontal, 12);
        std::memcpy(rd.cam_vertical, p->cam_vertical, 12);
        rd.width = p->width; rd.height = p->height;
        rd.row_begin = rows.begin; rd.row_step = rows.step; rd.num_rows = rows.count;
        rd.spp_pass = sn;
        rd.sample_base = p->sample_offset + s0;
        rd.stream_stride = p->stream_stride > 0 ? p->stream_stride : p->spp;
        rd.seed = p->seed;
        rd.max_depth = p->max_depth > 0 ? p->max_depth : 50;
        rd.rr_depth = p->rr_depth >= 0 ? p->rr_depth : 5;
        rd.npix = (uint32_t)npix;
        rd.total_work = (uint32_t)(npix * (uint64_t)sn);
        rd.num_regions = S->opt_xcd_regions > 0 ? (int)std::min<int64_t>(S->opt_xcd_regions, 8) : 8;
        rd.rows_per_region = (rows.count + rd.num_regions - 1) / rd.num_regions;
        rd.div_width = make_fastdiv((uint32_t)p->width);
        rd.div_spp = make_fastdiv((uint32_t)sn);
        rd.row_major = S->opt_item_order == 1 ? 1 : 0;
        rd.div_npix_full = make_fastdiv((uint32_t)rd.rows_per_region * (uint32_t)p->width);
        {   // the band that holds the remainder rows (all bands after it are empty)
            const int full = rows.count / rd.rows_per_region, rest = rows.count - full * rd.rows_per_region;
            rd.short_region = rest ? full : -1;
            rd.div_npix_last = make_fastdiv((uint32_t)std::max(rest, 1) * (uint32_t)p->width);
        }

        const int grid = launch_grid(S->num_cus, bpc, rd.total_work, block_threads);
        if (which == 1 && (size_t)grid * (use_q ? kQS * 64 : kBlock) * (size_t)S->tree[0].stack_cap > slot.redo_stack.n)
            return fail(PT_ERR_DEVICE, "internal error: rerun stacks smaller than the grid");
        S->info_grid = grid;
        // chunk: work items a wave reserves per atomic.  Big launches (a wave traces >= 2048 items): 128 — the waves of a
        // launch run dry within two paths' time of each other instead of four (cbox -1.9 %, bunny -0.9 % against 256; 64
        // buys nothing more and saturates the counters on cheap scenes: profiles/r02_tune_round36_*.log); mid-size
        // launches 256; small launches (interactive 1-2 spp frames) down to 64 so that the items are spread over all
        // resident waves instead of the first total/256 of them
        const uint64_t per_wave = rd.total_work / ((uint64_t)grid * (use_q ? kQS : kBlock / 64));     // waves that draw from the feed
        rd.chunk = per_wave >= 2048 ? 128u
                 : per_wave >= 4 * kMaxChunk ? kMaxChunk : (uint32_t)std::max<uint64_t>(64, std::min<uint64_t>(kMaxChunk, (per_wave / 64) * 64));
        if (S->opt_chunk > 0) rd.chunk = (uint32_t)std::min<int64_t>(kMaxChunk, std::max<int64_t>(64, (S->opt_chunk / 64) * 64));

        if (timing && frec.ev.size() <= (size_t)pass) {
            pt_scene::PassEvents fresh{};
            hipError_t ee = hipEventCreate(&fresh.t0);
            if (ee == hipSuccess && (ee = hipEventCreate(&fresh.t1)) != hipSuccess) (void)hipEventDestroy(fresh.t0);
            if (ee == hipSuccess && (ee = hipEventCreate(&fresh.r0)) != hipSuccess) { (void)hipEventDestroy(fresh.t0); (void)hipEventDestroy(fresh.t1); }
            if (ee == hipSuccess && (ee = hipEventCreate(&fresh.r1)) != hipSuccess) { (void)hipEventDestroy(fresh.t0); (void)hipEventDestroy(fresh.t1); (void)hipEventDestroy(fresh.r0); }
            if (ee != hipSuccess) return fail(PT_ERR_DEVICE, std::string("hipEventCreate: ") + hipGetErrorString(ee));
            frec.ev.push_back(fresh);
        }
        const pt_scene::PassEvents pe = timing ? frec.ev[pass] : pt_scene::PassEvents{};
        if (pass > 0) HIP_TRY(hipMemsetAsync(slot.ctl.p, 0, kWorkBytes, tstream));
        if (timing) HIP_TRY(hipEventRecord(pe.t0, tstream));
        if (use_q)
            hipLaunchKernelGGL(fnq, dim3(grid), dim3(kQBlock), lp.total, tstream, S->dev, rd, lp, qp, slot.samples.p,
                               S->work_counter(), S->counters());
        else
            hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), lp.total, tstream, S->dev, rd, lp, slot.samples.p,
                               S->work_counter(), S->counters());
        HIP_TRY(hipGetLastError());
        if (timing) HIP_TRY(hipEventRecord(pe.t1, tstream));
        if (own_stream) HIP_TRY(hipStreamWaitEvent(tstream, slot.in_ev, 0));   // the resolve writes the caller's buffer
        if (timing) HIP_TRY(hipEventRecord(pe.r0, tstream));

        const bool first = pass == 0, last = pass == n_pass - 1;
        int rmode;
        float scale = 1.0f;
        int rfirst;
        if (mode == 2) {
            // progressive: every pass adds its own partial sum; first only if sample_offset==0 and pass 0
            rmode = 2;
            rfirst = (first && p->sample_offset == 0) ? 1 : 0;
        } else if (last) {
            rmode = 0; rfirst = first ? 1 : 0; scale = 1.0f / (float)p->spp;
        } else {
            rmode = 1; rfirst = first ? 1 : 0;
        }
        hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, tstream, slot.samples.p, accum,
                           out_dev, (uint32_t)npix, sn, rmode, rfirst, scale);
        HIP_TRY(hipGetLastError());
        if (timing) HIP_TRY(hipEventRecord(pe.r1, tstream));
        HIP_TRY(hipEventRecord(slot.free_ev, tstream));
        slot.used = true; slot.free_stream = tstream;
        if (own_stream) HIP_TRY(hipStreamWaitEvent(stream, slot.free_ev, 0));
        S->info_passes++;
        if (timing) frec.passes = (size_t)pass + 1;
    }
    S->have_timing = timing;
    return PT_OK;
}

// trace_kernel_q bounds every wait on its rings; a wait that ran into its bound leaves a mark in counter slot 15 and an
// invalid frame behind.  (The stream has been synchronised.)
int check_schedule_error(const pt_scene* S) {
    if (S->info_kernel != 3 || !S->cur().ctl.p) return PT_OK;
    unsigned long long c[kSlotStride];
    int rc = read_slot_sums(S, c);
    if (rc) return rc;
    return c[15] ? fail(PT_ERR_DEVICE, "trace_kernel_q: schedule error, frame invalid (bits " + std::to_string(c[15]) + ": 1 bounded wait ran out, 2 ring entry corrupted, 4 impossible path state)") : PT_OK;
}

// HIP-event times of one recorded render call, summed over its sample passes (the events must have completed).
int frame_times(const pt_scene::FrameRec& f, double* kernel_ms, double* resolve_ms) {
    double t = 0, r = 0;
    for (size_t k = 0; k < f.passes; k++) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, f.ev[k].t0, f.ev[k].t1));
        t += ms;
        HIP_TRY(hipEventElapsedTime(&ms, f.ev[k].r0, f.ev[k].r1));
        r += ms;
    }
    *kernel_ms = t;
    *resolve_ms = r;
    return PT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int pt_api_version(void) { return PT_API_VERSION; }
const char* pt_last_error(void) { return g_err.c_str(); }

int pt_scene_create(const pt_scene_desc* desc, pt_scene** out) {
    if (!out) return fail(PT_ERR_INVALID_ARG, "null output pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available: the path tracer has no CPU fallback");
    pt_scene* S = new pt_scene();
    auto bail = [&](int rc) { pt_scene_destroy(S); return rc; };
    if (hipGetDevice(&S->device) != hipSuccess) return bail(fail(PT_ERR_DEVICE, "hipGetDevice failed"));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, S->device) != hipSuccess) return bail(fail(PT_ERR_DEVICE, "hipGetDeviceProperties failed"));
    S->num_cus = prop.multiProcessorCount;
    S->lds_per_block_max = prop.sharedMemPerBlock;
    int rc = validate_and_build(desc, S);
    if (rc) return bail(rc);
    *out = S;
    return PT_OK;
}

int pt_scene_destroy(pt_scene* S) {
    if (!S) return PT_OK;
    DeviceGuard guard;
    (void)guard.enter(S->device);
    if (S->last_stream || S->have_timing) (void)hipDeviceSynchronize();
    for (auto& T : S->tree) { T.nodes.release(); T.nodes_oct.release(); } S->drop_slots(); S->prims.release(); S->normals.release(); S->materials.release(); S->emission.release();
    S->lights.release(); S->accum.release(); S->fb_tmp.release();
    S->drop_events();
    delete S;
    return PT_OK;
}

int pt_render_async(pt_scene* S, const pt_render_params* p, float* fb_dev, void* hip_stream) {
    return launch_render(S, p, fb_dev, 0, reinterpret_cast<hipStream_t>(hip_stream));
}

int pt_render_accumulate(pt_scene* S, const pt_render_params* p, float* accum_dev, void* hip_stream) {
    return launch_render(S, p, accum_dev, 2, reinterpret_cast<hipStream_t>(hip_stream));
}

int pt_render(pt_scene* S, const pt_render_params* p, float* fb, int fb_on_device) {
    if (!S || !p || !fb) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (fb_on_device) {
        int rc = launch_render(S, p, fb, 0, nullptr);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(nullptr));
        return check_schedule_error(S);
    }
    RowSel rows;
    int rc = select_rows(p, &rows);
    if (rc) return rc;
    if (p->width <= 0) return fail(PT_ERR_INVALID_ARG, "width must be positive");
    const size_t n = (size_t)rows.count * (size_t)p->width * 3;
    DeviceGuard guard;
    { int grc = guard.enter(S->device); if (grc) return grc; }
    if ((rc = S->fb_tmp.ensure(n))) return rc;
    rc = launch_render(S, p, S->fb_tmp.p, 0, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(fb, S->fb_tmp.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return check_schedule_error(S);
}

int pt_get_counters(pt_scene* S, pt_counters* out) {
    if (!S || !out) return fail(PT_ERR_INVALID_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    DeviceGuard guard;
    { int grc = guard.enter(S->device); if (grc) return grc; }
    HIP_TRY(hipStreamSynchronize(S->last_stream));
    if (!S->cur().ctl.p) return PT_OK;
    unsigned long long c[kSlotStride];
    int rc = read_slot_sums(S, c);
    if (rc) return rc;
    out->paths = c[0]; out->segments = c[1]; out->node_visits = c[2]; out->leaf_tests = c[3];
    if (S->info_kernel == 3 && c[15]) return fail(PT_ERR_DEVICE, "trace_kernel_q: schedule error, frame invalid (bits " + std::to_string(c[15]) + ": 1 bounded wait ran out, 2 ring entry corrupted, 4 impossible path state)");
    if (S->have_timing && S->last_frame()) {
        int rc2 = frame_times(*S->last_frame(), &out->kernel_ms, &out->resolve_ms);
        if (rc2) return rc2;
    }
    return PT_OK;
}

int pt_get_frame_times(pt_scene* S, int max_frames, double* kernel_ms, double* resolve_ms, int* n_out) {
    if (!S || !kernel_ms || !resolve_ms || !n_out || max_frames < 0) return fail(PT_ERR_INVALID_ARG, "bad argument");
    *n_out = 0;
    DeviceGuard guard;
    { int grc = guard.enter(S->device); if (grc) return grc; }
    HIP_TRY(hipStreamSynchronize(S->last_stream));
    const uint64_t have = std::min<uint64_t>(S->frame_seq, S->frames.size());
    const int n = (int)std::min<uint64_t>(have, (uint64_t)max_frames);
    for (int k = 0; k < n; k++) {                        // oldest first
        const pt_scene::FrameRec& f = S->frames[(S->frame_seq - (uint64_t)n + (uint64_t)k) % S->frames.size()];
        int rc = frame_times(f, &kernel_ms[k], &resolve_ms[k]);
        if (rc) return rc;
    }
    *n_out = n;
    return PT_OK;
}

int pt_scene_set_option(pt_scene* S, const char* key, int64_t value) {
    if (!S || !key) return fail(PT_ERR_INVALID_ARG, "null argument");
    const std::string k(key);
    if (k == "blocks_per_cu") {
        if (value < 0 || value > 32) return fail(PT_ERR_INVALID_ARG, "blocks_per_cu must be 0 (automatic) .. 32");
        S->opt_blocks_per_cu = value;
    }
    else if (k == "scratch_bytes") S->opt_scratch_bytes = value;
    else if (k == "force_global") S->opt_force_global = value;
    else if (k == "stats") S->opt_stats = value;
    else if (k == "xcd_regions") S->opt_xcd_regions = value;
    else if (k == "octants") S->opt_octants = value;
    else if (k == "top_cache") S->opt_top_cache = value;
    else if (k == "fast_tree") S->opt_fast_tree = value;
    else if (k == "lds_budget_kb") S->opt_lds_budget_kb = value;
    else if (k == "chunk") S->opt_chunk = value;
    else if (k == "item_order") S->opt_item_order = value;
    else if (k == "specialize") S->opt_specialize = value;
    else if (k == "timing_frames") {
        if (value < 0 || value > 4096) return fail(PT_ERR_INVALID_ARG, "timing_frames must be 0 .. 4096");
        S->opt_timing_frames = value;
    }
    else if (k == "kernel") { if (value < 1 || value > 3) return fail(PT_ERR_INVALID_ARG, "kernel must be 1, 2 or 3"); S->opt_kernel = value; }
    else if (k == "frames_in_flight") {
        if (value < 1 || value > kMaxFramesInFlight) return fail(PT_ERR_INVALID_ARG, "frames_in_flight must be 1 .. " + std::to_string(kMaxFramesInFlight));
        S->opt_frames_in_flight = value;
    }
    else if (k == "q_target") S->opt_q_target = value;
    else if (k == "q_swap") S->opt_q_swap = value;
    else if (k == "q_low") S->opt_q_low = value;
    else if (k == "v2_thresh") S->opt_v2_thresh = value;
    else if (k == "v2_inner") S->opt_v2_inner = value;
    else if (k == "v2_minw") S->opt_v2_minw = value;
    else return fail(PT_ERR_INVALID_ARG, "unknown option " + k);
    return PT_OK;
}

int pt_scene_get_info(pt_scene* S, const char* key, int64_t* value) {
    if (!S || !key || !value) return fail(PT_ERR_INVALID_ARG, "null argument");
    const std::string k(key);
    if (k == "grid") *value = S->info_grid;
    else if (k == "lds_bytes") *value = S->info_lds_bytes;
    else if (k == "lds_scene") *value = S->info_lds_scene;
    else if (k == "residency") *value = scene_residency(S, which_tree(S));
    else if (k == "passes") *value = S->info_passes;
    else if (k == "occupancy") *value = S->info_occupancy;                // what the occupancy query allows
    else if (k == "blocks_per_cu") *value = S->info_blocks_per_cu;        // what the last launch used
    else if (k == "redo_segments") {                                     // STATS: segments rerun in reference order by the last render
        DeviceGuard guard;
        { int grc = guard.enter(S->device); if (grc) return grc; }
        HIP_TRY(hipStreamSynchronize(S->last_stream));
        unsigned long long c[kSlotStride] = {0};
        if (S->cur().ctl.p) { int rc = read_slot_sums(S, c); if (rc) return rc; }
        *value = (int64_t)c[12];
    }
    else if (k == "num_cus") *value = S->num_cus;
    else if (k == "bvh_depth") *value = S->tree[0].depth;
    else if (k == "stack_entries") *value = S->tree[which_tree(S)].stack_cap;   // per lane, sentinel included
    else if (k == "fast_tree") *value = S->have_fast ? 1 : 0;             // an internal tree exists (the caller's tree is nested)
    else if (k == "debug_reruns") *value = S->info_debug_reruns;          // rays the last pt_debug_intersect reran in reference order
    else if (k == "fast_tree_cost_permille") *value = S->fast_cost_permille;   // summed inner-box area, internal tree / caller's tree x 1000 (0: none built)
    else if (k == "fast_tree_is_callers") *value = (S->have_fast && S->fast_is_callers_topology) ? 1 : 0;
    else if (k == "fast_tree_on") *value = which_tree(S);    // ... and the next exact render traverses it
    else if (k == "fast_tree_depth") *value = S->have_fast ? S->tree[1].depth : 0;
    else if (k == "scene_bytes") *value = S->scene_bytes;
    else if (k == "num_inner_nodes") *value = S->tree[0].num_nodes;
    else if (k == "top_nodes") { const int w = which_tree(S); *value = make_plan(S, scene_residency(S, w), false, w).top_count; }
    else if (k == "device") *value = S->device;
    else if (k == "sweep_on_device") *value = S->sweep_on_device;         // the internal tree was built on the GPU (pt_sweep_build.hip)
    else if (k.rfind("create_us", 0) == 0 && k.size() == 10 && k[9] >= '0' && k[9] <= '6') *value = S->create_us[k[9] - '0'];
    else if (k == "kernel") *value = S->info_kernel;                      // the kernel the last render ran on (1, 2 or 3)
    else if (k == "block_threads") *value = S->info_kernel == 3 ? kQBlock : kBlock;
    else if (k == "frames_in_flight") *value = S->opt_frames_in_flight;
    else if (k.rfind("qdiag", 0) == 0 && k.size() >= 6 && k.size() <= 7 && k.find_first_not_of("0123456789", 5) == std::string::npos &&
             std::stoi(k.substr(5)) < 11) {
        // schedule diagnostics of the last STATS render of trace_kernel_q: counter slots [4..14], see pt_kernel_q.h
        DeviceGuard guard;
        { int grc = guard.enter(S->device); if (grc) return grc; }
        HIP_TRY(hipStreamSynchronize(S->last_stream));
        unsigned long long c[kSlotStride] = {0};
        if (S->cur().ctl.p) { int rc = read_slot_sums(S, c); if (rc) return rc; }
        *value = (int64_t)c[4 + std::stoi(k.substr(5))];
    }
    else if (k.rfind("diag", 0) == 0 && k.size() >= 5 && k.size() <= 7 && k.find_first_not_of("0123456789", 4) == std::string::npos &&
             std::stoi(k.substr(4)) < 8 + kNumCounters - kTimelineBase) {
        // schedule diagnostics / launch timeline of the last STATS render (trace_kernel_v2): see pt_kernels.h
        DeviceGuard guard;
        { int grc = guard.enter(S->device); if (grc) return grc; }
        HIP_TRY(hipStreamSynchronize(S->last_stream));
        unsigned long long v = 0;
        const int idx = std::stoi(k.substr(4));
        if (S->cur().ctl.p && idx < 8) {                  // schedule diagnostics: summed over the counter slots
            unsigned long long c[kSlotStride];
            int rc = read_slot_sums(S, c);
            if (rc) return rc;
            v = c[4 + idx];
        } else if (S->cur().ctl.p) {                      // launch timeline / histograms
            HIP_TRY(hipMemcpy(&v, S->counters() + kTimelineBase + (idx - 8), sizeof v, hipMemcpyDeviceToHost));
        }
        *value = (int64_t)v;
    }
    else if (k == "vgprs" || k == "vgprs_pruned") {
        hipFuncAttributes fa;
        const int w = which_tree(S);
        const int res = scene_residency(S, w);
        const void* f;
        if (S->opt_kernel == 3 && k == "vgprs") {
            f = reinterpret_cast<const void*>(pick_kernel_q(S, res, w == 1));
        } else {
            TraceFn fn = pick_kernel(S, res, k == "vgprs_pruned", S->opt_stats != 0, w == 1);
            if (!fn) return fail(PT_ERR_INVALID_ARG, "no such kernel variant");
            f = reinterpret_cast<const void*>(fn);
        }
        HIP_TRY(hipFuncGetAttributes(&fa, f));
        *value = fa.numRegs;
    } else return fail(PT_ERR_INVALID_ARG, "unknown info key " + k);
    return PT_OK;
}

int pt_debug_math(int op, const float* x, const float* y, float* out0, float* out1, int n) {
    if (!x || !y || !out0 || !out1 || n < 0 || op < 0 || op > 2) return fail(PT_ERR_INVALID_ARG, "bad argument");
    if (n == 0) return PT_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    DevBuf<float> dx, dy, d0, d1;
    int rc;
    if ((rc = dx.ensure(n)) || (rc = dy.ensure(n)) || (rc = d0.ensure(n)) || (rc = d1.ensure(n))) return rc;
    hipError_t e = hipMemcpy(dx.p, x, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dy.p, y, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d1.p, 0, n * sizeof(float));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(math_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, op, dx.p, dy.p, d0.p, d1.p, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out0, d0.p, n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out1, d1.p, n * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(PT_ERR_DEVICE, std::string("pt_debug_math: ") + hipGetErrorString(e));
    return PT_OK;
}

int pt_debug_math_host(int op, const float* x, const float* y, float* out0, float* out1, int n) {
    if (!x || !y || !out0 || !out1 || n < 0 || op < 0 || op > 2) return fail(PT_ERR_INVALID_ARG, "bad argument");
    for (int k = 0; k < n; k++) {
        if (op == 0) {
            ptm::sincos_det(x[k], out0[k], out1[k]);
        } else if (op == 1) {
            out0[k] = ptm::pow_det(x[k], y[k]);
            out1[k] = 0.0f;
        } else {
            ptm::Pcg r = ptm::pcg_init((uint64_t)__builtin_bit_cast(uint32_t, x[k]), (uint64_t)__builtin_bit_cast(uint32_t, y[k]));
            out0[k] = ptm::pcg_float(r);
            out1[k] = ptm::pcg_float(r);
        }
    }
    return PT_OK;
}

int pt_bvh_build_sweep(const pt_scene_desc* d, pt_bvh_node* out_nodes, int32_t* out_root, int32_t* out_depth, double* out_build_ms) {
    if (!d || !out_nodes || !out_root) return fail(PT_ERR_INVALID_ARG, "bad argument");
    const int N = d->num_shapes;
    if (N <= 0 || !d->nodes || d->num_nodes != 2 * N - 1) return fail(PT_ERR_BAD_SCENE, "needs the leaf boxes of a 2*num_shapes-1 node pool");
    std::vector<float> boxes((size_t)N * 6);
    std::vector<char> seen(N, 0);
    for (int k = 0; k < d->num_nodes; k++) {
        const pt_bvh_node& nd = d->nodes[k];
        if (nd.prim == -1) continue;
        if (nd.prim < 0 || nd.prim >= N || seen[nd.prim]) return fail(PT_ERR_BAD_SCENE, "leaves must cover every shape exactly once");
        seen[nd.prim] = 1;
        std::memcpy(&boxes[(size_t)nd.prim * 6], nd.bmin, 12);
        std::memcpy(&boxes[(size_t)nd.prim * 6 + 3], nd.bmax, 12);
    }
    for (int i = 0; i < N; i++)
        if (!seen[i]) return fail(PT_ERR_BAD_SCENE, "leaves must cover every shape exactly once");
    for (float v : boxes)
        if (!std::isfinite(v)) return fail(PT_ERR_BAD_SCENE, "leaf box is not finite");
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<pt_bvh_node> nodes;
    int32_t root = 0, depth = 0;
    try {
        pts::build_sweep_tree(boxes.data(), N, nodes, &root, &depth);
    } catch (const std::exception& e) {
        return fail(PT_ERR_DEVICE, std::string("pt_bvh_build_sweep: ") + e.what());
    }
    std::memcpy(out_nodes, nodes.data(), nodes.size() * sizeof(pt_bvh_node));
    *out_root = root;
    if (out_depth) *out_depth = depth;
    if (out_build_ms) *out_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}

int pt_bvh_build_sweep_device(const pt_scene_desc* d, pt_bvh_node* out_nodes, int32_t* out_root, int32_t* out_depth, double* out_build_ms) {
    if (!d || !out_nodes || !out_root) return fail(PT_ERR_INVALID_ARG, "bad argument");
    const int N = d->num_shapes;
    if (N <= 0 || !d->nodes || d->num_nodes != 2 * N - 1) return fail(PT_ERR_BAD_SCENE, "needs the leaf boxes of a 2*num_shapes-1 node pool");
    std::vector<float> boxes((size_t)N * 6);
    std::vector<char> seen(N, 0);
    for (int k = 0; k < d->num_nodes; k++) {
        const pt_bvh_node& nd = d->nodes[k];
        if (nd.prim == -1) continue;
        if (nd.prim < 0 || nd.prim >= N || seen[nd.prim]) return fail(PT_ERR_BAD_SCENE, "leaves must cover every shape exactly once");
        seen[nd.prim] = 1;
        std::memcpy(&boxes[(size_t)nd.prim * 6], nd.bmin, 12);
        std::memcpy(&boxes[(size_t)nd.prim * 6 + 3], nd.bmax, 12);
    }
    for (int i = 0; i < N; i++)
        if (!seen[i]) return fail(PT_ERR_BAD_SCENE, "leaves must cover every shape exactly once");
    for (float v : boxes)
        if (!std::isfinite(v)) return fail(PT_ERR_BAD_SCENE, "leaf box is not finite");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    std::vector<pt_bvh_node> nodes;
    int32_t root = 0, depth = 0;
    int rc = pts::sweep_build_device(boxes.data(), N, nodes, &root, &depth, out_build_ms);
    if (rc) return rc;
    std::memcpy(out_nodes, nodes.data(), nodes.size() * sizeof(pt_bvh_node));
    *out_root = root;
    if (out_depth) *out_depth = depth;
    return PT_OK;
}

int pt_debug_intersect(pt_scene* S, const float* rays, int n, int traversal, float* out_tuv, int32_t* out_prim) {
    if (!S || !rays || !out_tuv || !out_prim || n < 0) return fail(PT_ERR_INVALID_ARG, "bad argument");
    if (n == 0) return PT_OK;
    DeviceGuard guard;
    { int grc = guard.enter(S->device); if (grc) return grc; }
    DevBuf<float> dr, dt;
    DevBuf<int32_t> dp, dn;
    int rc;
    if ((rc = dr.ensure((size_t)n * 8)) || (rc = dt.ensure((size_t)n * 3)) || (rc = dp.ensure(n)) || (rc = dn.ensure(1))) return rc;
    // the tree renders traverse — exact and pruned alike (which_tree): the internal tree with ties settled in the caller's visit
    // order and reference-order reruns when there is one (whatever the scene's size: this kernel reads nodes from global
    // memory), so that the closest hits — ties included — can be checked ray by ray; else the caller's
    const bool fast = S->have_fast && S->opt_fast_tree;
    select_tree(S, fast ? 1 : 0, fast);
    const int cap = fast && S->dev.redo_cap > S->dev.stack_cap ? S->dev.redo_cap : S->dev.stack_cap;
    const uint32_t lds = (uint32_t)(kBlock / 64) * (uint32_t)cap * 64u * 4u;
    hipError_t e = hipMemcpy(dr.p, rays, (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dn.p, 0, sizeof(int32_t));
    if (e == hipSuccess) {
        if (traversal == PT_TRAVERSAL_PRUNED)
            hipLaunchKernelGGL(intersect_kernel<true>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), lds, nullptr, S->dev, dr.p, n, dt.p, dp.p, dn.p);
        else
            hipLaunchKernelGGL(intersect_kernel<false>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), lds, nullptr, S->dev, dr.p, n, dt.p, dp.p, dn.p);
        e = hipGetLastError();
    }
    int32_t reruns = 0;
    if (e == hipSuccess) e = hipMemcpy(&reruns, dn.p, sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) S->info_debug_reruns = reruns;
    if (e == hipSuccess) e = hipMemcpy(out_tuv, dt.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_prim, dp.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(PT_ERR_DEVICE, std::string("pt_debug_intersect: ") + hipGetErrorString(e));
    return PT_OK;
}

}  // extern "C"
