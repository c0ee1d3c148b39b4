// pt_kernel_q.h — trace_kernel_q: the path tracer with paths REGROUPED ACROSS THE WAVES OF A WORKGROUP by what they need next
// (included only by pt_api.hip; option "kernel" = 3; every residency, exact traversal, no next-event estimation).
//
// trace_kernel_v2 keeps one path per lane for the path's whole life: a lane whose traversal is finished waits until enough
// lanes of ITS wave wait with it before the wave shades and refills (47 % of its VALU lane slots do work on cbox).  Here the
// waves of a workgroup have roles and paths move between them through two rings in LDS:
//
//   T-waves (kQT of them)   traversal only: every lane holds one ray and runs the inner_step / leaf_step bursts of v2.  A lane
//                           whose traversal is finished hands the path — ray, closest hit, radiance so far, throughput, RNG —
//                           to the SHADE ring and takes the next ready ray from the READY ring, whichever path that is.
//   S-waves (kQS of them)   take 64 finished paths from the shade ring at a time and shade them with full waves
//                           (radiance.cuh:32-74); survivors go to the ready ring with their next ray (1/d computed here, on
//                           full waves); ended paths store their sample.  S-waves also START paths, 64 at a time
//                           (main.cu:32-44), whenever the workgroup holds fewer paths than its target.
//
// A path is 22 words; it lives in a T-lane's registers or in a 96-B ring entry, nowhere else.  Which lane or wave executes a
// step never changes what is computed for a path: the frame is bit-identical to trace_kernel_v2's and the oracle's
// (tests/test_gpu_parity.py::test_scheduler_variants_are_bit_identical, test_gpu_fullsize.py).
//
// Rings: 32-bit head / tail cursors claimed with compare-and-swap by one lane per wave.  A position may be CLAIMED by a
// consumer as soon as a producer has RESERVED it, and reserved again (one lap later) as soon as it has been claimed — so
// every entry carries a sequence word that orders the four parties of two consecutive laps:  2 x lap = free for that
// lap's producer, 2 x lap + 1 = written, 2 x lap + 2 = read = free for the next lap's producer.  Every wait is bounded (kQSpinLimit / the wall-clock watchdog): a logic error
// ends the launch with an error flag (PT_ERR_DEVICE on the host) instead of hanging the GPU.
// Bounds that keep the rings from deadlocking: paths in flight per workgroup <= q.target <= T-lanes + 2 x ring - 64 (host checks).
#pragma once

#include "pt_kernels.h"

namespace ptk {

#ifndef PT_Q_NT
#define PT_Q_NT 9
#endif
#ifndef PT_Q_NS
#define PT_Q_NS 3
#endif
#ifndef PT_Q_RING
#define PT_Q_RING 256
#endif
#ifndef PT_Q_MINW
#define PT_Q_MINW 6
#endif
#ifndef PT_Q_SPRIO
#define PT_Q_SPRIO 2
#endif
#ifndef PT_Q_BURST_IN
#define PT_Q_BURST_IN 6         // inner steps and
#endif
#ifndef PT_Q_BURST_LF
#define PT_Q_BURST_LF 2         // leaf steps of a T-wave between two looks at the rings
#endif
constexpr int kQT = PT_Q_NT;                        // traversal waves per workgroup
constexpr int kQS = PT_Q_NS;                        // shading waves per workgroup
constexpr int kQBlock = (kQT + kQS) * 64;
constexpr uint32_t kQRing = PT_Q_RING;              // entries per ring (power of two) for LDS-resident scenes; scenes in global memory,
                                                    // whose traversal stacks are 32-bit and deeper, run with half of it (QParams::ring_log2)
constexpr uint32_t kQEntryBytes = 96;               // 24 words: 22 of path, flag, pad
constexpr uint32_t kQCtlBytes = 64;
constexpr uint32_t kQSpinLimit = 1u << 22;
constexpr unsigned long long kQWatchdogTicks = 200000000ull;    // 2 s of the 100 MHz wall clock WITHOUT anything to do: no legitimate wait is a thousandth of that
static_assert((kQRing & (kQRing - 1)) == 0, "ring size must be a power of two");

// Runtime knobs of the schedule (pt_scene_set_option "q_target" / "q_swap" / "q_low"); none of them changes a result.
struct QParams {
    int32_t target;      // paths in flight per workgroup that S-waves fill up to
    int32_t swap;        // a T-wave exchanges finished / empty lanes with the rings when at least this many lanes want it
    int32_t low;         // an S-wave shades a partial batch (fewer than 64 finished paths) only while the ready ring holds fewer rays than this
    uint32_t ctl_off, shade_off, ready_off;   // LDS byte offsets: control words, the two rings
    uint32_t ring_log2;  // entries per ring = 1 << ring_log2
};

using ptd::f4v;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) f4v lds_f4w;
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f2v lds_f2w;

// control words (uint32 index into the block's control area)
enum { kQsHead = 0, kQsTail = 1, kQrHead = 2, kQrTail = 3, kQLive = 4, kQFeedsOpen = 5, kQError = 6 };

#define PT_Q_WG __HIP_MEMORY_SCOPE_WORKGROUP

__device__ __forceinline__ uint32_t q_load(lds_u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, PT_Q_WG); }

// Wave-uniform.  Reserves n consecutive ring positions for writing; false when the ring has no room for them right now.
__device__ __forceinline__ bool q_reserve(lds_u32* head_tail, uint32_t n, int lane, uint32_t& pos, uint32_t ring) {
    uint32_t ok = 0, p = 0;
    if (lane == 0) {
        for (int tries = 0; tries < 8 && !ok; tries++) {
            const uint32_t head = q_load(head_tail);
            uint32_t tail = q_load(head_tail + 1);
            if (tail + n - head > ring) break;               // a stale head only makes this stricter
            if (__hip_atomic_compare_exchange_strong(head_tail + 1, &tail, tail + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, PT_Q_WG)) {
                ok = 1; p = tail;
            }
        }
    }
    pos = __builtin_amdgcn_readfirstlane(p);
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

// Wave-uniform.  Claims up to `want` (at least `at_least`) written-or-being-written positions for reading; returns how many.
__device__ __forceinline__ uint32_t q_claim(lds_u32* head_tail, uint32_t want, uint32_t at_least, int lane, uint32_t& pos) {
    uint32_t got = 0, p = 0;
    if (lane == 0) {
        for (int tries = 0; tries < 8 && !got; tries++) {
            uint32_t head = q_load(head_tail);
            const uint32_t tail = q_load(head_tail + 1);
            const uint32_t avail = tail - head;              // a stale tail only makes this smaller
            const uint32_t n = avail < want ? avail : want;
            if (n == 0 || n < at_least) break;
            if (__hip_atomic_compare_exchange_strong(head_tail, &head, head + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, PT_Q_WG)) {
                got = n; p = head;
            }
        }
    }
    pos = __builtin_amdgcn_readfirstlane(p);
    return __builtin_amdgcn_readfirstlane(got);
}

// The 22 words of a path.  a0..a3: the closest hit (t, u, v, primitive) on the way to shading; 1/d (and nothing) on the way
// to traversal.  depth_flags: bits 0..15 bounces so far (the ray in flight is a camera ray iff 0), bit 31 = trace this ray on
// the caller's tree in reference order (1/d not finite, pt_api.hip: validate_and_build).
struct QPath {
    ptm::V3 org, dir;
    float a0, a1, a2, a3;
    ptm::V3 L, T;
    ptm::Pcg rng;
    uint32_t depth_flags, my_w;
};

constexpr uint32_t kQRedoBit = 0x80000000u;

__device__ __forceinline__ unsigned char* q_entry(unsigned char* ring, uint32_t pos, uint32_t rl) {
    // An entry is kept as two halves of 48 B in two arrays (words 0..11 in the first, 12..23 — the sequence word among them —
    // ring x 48 B further on): consecutive lanes move consecutive entries, and 128-bit LDS accesses 48 B apart touch every
    // bank once, where a 96-B pitch put two lanes of each group on the same banks (40 % of the LDS cycles were conflicts).
    return ring + (pos & ((1u << rl) - 1u)) * (kQEntryBytes / 2);
}
__device__ __forceinline__ uint32_t q_half(uint32_t rl) { return (kQEntryBytes / 2) << rl; }      // byte distance between the two halves of an entry
// Sequence word of position `pos` when it is free for its producer (see the header of this file).
__device__ __forceinline__ uint32_t q_seq(uint32_t pos, uint32_t rl) { return (pos >> rl) * 2u; }
// ... and the sequence arithmetic wraps where the 32-bit positions do (after 2^32 / ring laps)
__device__ __forceinline__ uint32_t q_seq_mask(uint32_t rl) { return (uint32_t)((2ull << (32u - rl)) - 1ull); }

// error bits: 1 a bounded wait ran out, 2 an entry failed its check word, 4 a path carried an impossible primitive or sample index
__device__ __forceinline__ void q_flag_error(lds_u32* ctl, uint32_t bits = 1u) { __hip_atomic_fetch_or(ctl + kQError, bits, __ATOMIC_RELAXED, PT_Q_WG); }
#ifndef PT_Q_CHECK
#define PT_Q_CHECK 0
#endif
__device__ __forceinline__ uint32_t q_check_word(const QPath& p) {
    uint32_t h = 0x9e3779b9u;
    auto mix = [&h](float f) { h = (h ^ __builtin_bit_cast(uint32_t, f)) * 0x01000193u; };
    mix(p.org.x); mix(p.org.y); mix(p.org.z); mix(p.dir.x); mix(p.dir.y); mix(p.dir.z); mix(p.a0); mix(p.a1); mix(p.a2); mix(p.a3);
    mix(p.L.x); mix(p.L.y); mix(p.L.z); mix(p.T.x); mix(p.T.y); mix(p.T.z);
    h = (h ^ (uint32_t)p.rng.state) * 0x01000193u; h = (h ^ (uint32_t)(p.rng.state >> 32)) * 0x01000193u;
    h = (h ^ (uint32_t)p.rng.inc) * 0x01000193u; h = (h ^ (uint32_t)(p.rng.inc >> 32)) * 0x01000193u;
    h = (h ^ p.depth_flags) * 0x01000193u; h = (h ^ p.my_w) * 0x01000193u;
    return h;
}

// Producer side of one entry (the position has been reserved): wait until its last reader is done, write, publish.
__device__ __forceinline__ void q_write(unsigned char* ring, uint32_t pos, const QPath& p, lds_u32* ctl, uint32_t rl) {
    unsigned char* e = q_entry(ring, pos, rl);
    unsigned char* e2 = e + q_half(rl);
    const uint32_t seq = q_seq(pos, rl);
    lds_u32* flag = (lds_u32*)(e2 + 40);
    uint32_t spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, PT_Q_WG) != seq) {
        if (++spins > kQSpinLimit) { q_flag_error(ctl); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    lds_f4w* w = (lds_f4w*)e;
    f4v v0 = {p.org.x, p.org.y, p.org.z, p.dir.x};
    f4v v1 = {p.dir.y, p.dir.z, p.a0, p.a1};
    f4v v2 = {p.a2, p.a3, p.L.x, p.L.y};
    f4v v3 = {p.L.z, p.T.x, p.T.y, p.T.z};
    f4v v4 = {__builtin_bit_cast(float, (uint32_t)p.rng.state), __builtin_bit_cast(float, (uint32_t)(p.rng.state >> 32)),
              __builtin_bit_cast(float, (uint32_t)p.rng.inc), __builtin_bit_cast(float, (uint32_t)(p.rng.inc >> 32))};
    f2v v5 = {__builtin_bit_cast(float, p.depth_flags), __builtin_bit_cast(float, p.my_w)};
    lds_f4w* w2 = (lds_f4w*)e2;
    w[0] = v0; w[1] = v1; w[2] = v2; w2[0] = v3; w2[1] = v4;
    *(lds_f2w*)(e2 + 32) = v5;
    if (PT_Q_CHECK) *(lds_u32*)(e2 + 44) = q_check_word(p);
    __hip_atomic_store(flag, seq + 1u, __ATOMIC_RELEASE, PT_Q_WG);
}

// Consumer side (the position has been claimed): wait until the producer has published it, read, hand the entry back.
__device__ __forceinline__ void q_read(unsigned char* ring, uint32_t pos, QPath& p, lds_u32* ctl, uint32_t rl) {
    unsigned char* e = q_entry(ring, pos, rl);
    unsigned char* e2 = e + q_half(rl);
    const uint32_t seq = q_seq(pos, rl);
    lds_u32* flag = (lds_u32*)(e2 + 40);
    uint32_t spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, PT_Q_WG) != seq + 1u) {
        if (++spins > kQSpinLimit) { q_flag_error(ctl); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    const lds_f4w* w = (const lds_f4w*)e;
    const lds_f4w* w2 = (const lds_f4w*)e2;
    const f4v v0 = w[0], v1 = w[1], v2 = w[2], v3 = w2[0], v4 = w2[1];
    const f2v v5 = *(const lds_f2w*)(e2 + 32);
    p.org = ptm::mk(v0.x, v0.y, v0.z); p.dir = ptm::mk(v0.w, v1.x, v1.y);
    p.a0 = v1.z; p.a1 = v1.w; p.a2 = v2.x; p.a3 = v2.y;
    p.L = ptm::mk(v2.z, v2.w, v3.x); p.T = ptm::mk(v3.y, v3.z, v3.w);
    // (elements are copied into floats first: __builtin_bit_cast applied to an ext-vector element expression reads element 0)
    const float s0 = v4.x, s1 = v4.y, s2 = v4.z, s3 = v4.w, d0 = v5.x, d1 = v5.y;
    p.rng.state = (uint64_t)__builtin_bit_cast(uint32_t, s0) | ((uint64_t)__builtin_bit_cast(uint32_t, s1) << 32);
    p.rng.inc = (uint64_t)__builtin_bit_cast(uint32_t, s2) | ((uint64_t)__builtin_bit_cast(uint32_t, s3) << 32);
    p.depth_flags = __builtin_bit_cast(uint32_t, d0); p.my_w = __builtin_bit_cast(uint32_t, d1);
    const uint32_t chk = PT_Q_CHECK ? *(lds_u32*)(e2 + 44) : 0u;
    __hip_atomic_store(flag, (seq + 2u) & q_seq_mask(rl), __ATOMIC_RELEASE, PT_Q_WG);
    if (PT_Q_CHECK && chk != q_check_word(p)) q_flag_error(ctl, 2u);
}

// Counter slots of the regrouping kernel beyond [0..3] (paths, segments, node visits, leaf tests); STATS builds only:
// [4] T-wave loop iterations  [5] exchanges  [6] lanes pushed  [7] inner steps  [8] lanes active in them  [9] leaf steps
// [10] lanes active in them  [11] shade batches  [12] reruns on the caller's tree  [13] lanes in shade batches
// [14] start batches  [15] T-wave iterations that found the wave empty
// RES as in trace_kernel_v2 (0 global memory, 1 LDS, 2 LDS + octant tables, 3 global memory + top of the tree in LDS).
// POSTPONE (internal tree, scenes in global memory): a lane that reaches a leaf sets it aside and goes on with its next stack entry;
// the burst is v2's for that case, two rounds of 3 inner steps + 1 leaf step (pt_api.hip: pick_kernel has the measurements).
template <int RES, bool STATS, int SPEC, bool POSTPONE = false>
__global__ __launch_bounds__(kQBlock, PT_Q_MINW) void trace_kernel_q(SceneDev scn, RenderDev rp, LdsPlan lp, QParams q,
                                                                      float4* __restrict__ samples,
                                                                      uint32_t* __restrict__ work_counter,
                                                                      unsigned long long* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    lds_u32* ctl = (lds_u32*)(smem + q.ctl_off);
    unsigned char* shade_ring = smem + q.shade_off;
    unsigned char* ready_ring = smem + q.ready_off;
    // control words and entry flags start at zero; feeds_open = the S-waves
    const uint32_t rl = q.ring_log2, ring_n = 1u << rl;
    for (uint32_t i = threadIdx.x; i < (kQCtlBytes + 2 * ring_n * kQEntryBytes) / 4; i += blockDim.x) {
        // the control area and the two rings are contiguous (pt_api.hip: make_plan_q)
        ((lds_u32*)(smem + q.ctl_off))[i] = (i == (uint32_t)kQFeedsOpen) ? (uint32_t)kQS : 0u;
    }
    const ptd::SceneView sv = make_scene_view<RES>(scn, lp, smem);      // stages the scene; ends with a __syncthreads()

    constexpr bool TRI_ONLY = SPEC >= 1, DIFFUSE_ONLY = SPEC >= 2;
    using STK = typename std::conditional<RES == 1 || RES == 2, int16_t, int32_t>::type;     // as v2: 16-bit references where the scene is small
    constexpr int TOP = RES == 3 ? 1 : 0;
    constexpr int32_t DONE = ptd::done_value<STK>();
    const bool fbk = scn.fallback != 0;
    unsigned long long idle_since = 0;       // wall clock (100 MHz) when this wave last ran out of things to do; 0 = it is busy
    uint32_t n_paths = 0, n_segs = 0, n_redo = 0;
    ptd::TravStats st;
    st.nodes = 0; st.leaves = 0;
    unsigned long long dg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    if (wave < kQT) {
        // ---------------------------------------------------------------------------------- traversal role
        STK* stk = reinterpret_cast<STK*>(smem + lp.stack_off) + (size_t)wave * scn.stack_cap * 64 + lane;
        ptd::stack_init(stk);
        bool has = false;
        ptd::Ray ray;
        ray.org = ptm::mk(0, 0, 0); ray.dir = ptm::mk(0, 0, 1); ray.tnear = 0; ray.tfar = 0;
        ptd::Trav tv;
        tv.inv = ptm::mk(1, 1, 1);
        tv.best.t = 0; tv.best.u = 0; tv.best.v = 0; tv.best.prim = -1;
        tv.cur = DONE; tv.sp = 1; tv.node_off = 0; tv.redo = false;
        int32_t pend = DONE;                    // POSTPONE: the leaf set aside
        // what the lane only carries from ring to ring
        ptm::V3 cL = ptm::mk(0, 0, 0), cT = ptm::mk(1, 1, 1);
        ptm::Pcg crng;
        crng.state = 0; crng.inc = 1;
        uint32_t cdepth = 0, cw = 0;
        for (;;) {
            if (STATS) dg[0]++;
            // ---- traversal burst: 6 inner + 2 leaf steps (the shape v2 uses for LDS-resident scenes), or — POSTPONE — two rounds of
            // 3 inner steps + 1 leaf step with leaves set aside
            if (POSTPONE) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        if (STATS) {
                            const int n_in = __popcll(__ballot(tv.cur >= 0));
                            if (n_in) { dg[3]++; dg[4] += (unsigned)n_in; }
                        }
                        if (tv.cur >= 0) {
                            if (STATS) st.nodes++;
                            ptd::inner_step<false, false, STK, TOP>(sv, ray.org, tv, stk);
                            if (tv.cur < 0 && tv.cur != DONE && pend == DONE) {       // a leaf: set it aside, take the next entry
                                pend = tv.cur;
                                tv.sp--;
                                tv.cur = stk[tv.sp * 64];
                            }
                        }
                    }
                    if (STATS) {
                        const int n_lf = __popcll(__ballot(pend != DONE));
                        if (n_lf) { dg[5]++; dg[6] += (unsigned)n_lf; }
                    }
                    if (pend != DONE) {
                        if (STATS) st.leaves++;
                        ptd::leaf_test<TRI_ONLY>(sv, ray, tv, pend, fbk);
                        pend = DONE;
                        if (tv.cur < 0 && tv.cur != DONE) {                           // the lane was blocked on a second leaf
                            pend = tv.cur;
                            tv.sp--;
                            tv.cur = stk[tv.sp * 64];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < PT_Q_BURST_IN; k++) {
                    if (STATS) {
                        const int n_in = __popcll(__ballot(tv.cur >= 0));
                        if (n_in) { dg[3]++; dg[4] += (unsigned)n_in; }
                    }
                    if (tv.cur >= 0) {
                        if (STATS) st.nodes++;
                        ptd::inner_step<false, RES == 2, STK, TOP>(sv, ray.org, tv, stk);
                    }
                }
#pragma unroll
                for (int k = 0; k < PT_Q_BURST_LF; k++) {
                    if (STATS) {
                        const int n_lf = __popcll(__ballot(tv.cur < 0 && tv.cur != DONE));
                        if (n_lf) { dg[5]++; dg[6] += (unsigned)n_lf; }
                    }
                    if (tv.cur < 0 && tv.cur != DONE) {
                        if (STATS) st.leaves++;
                        ptd::leaf_step<STK, TRI_ONLY, false>(sv, ray, tv, stk, false, fbk);
                    }
                }
            }
            // ---- exchange with the rings
            const bool fin = has && tv.cur == DONE && (!POSTPONE || pend == DONE);
            const unsigned long long fin_m = __ballot(fin);
            const int n_fin = __popcll(fin_m);
            const int n_emp = __popcll(__ballot(!has));
            const int n_trav = 64 - n_fin - n_emp;
            // Push when enough lanes have finished to be worth the ring's atomics (or nothing else is left to do); pull when
            // there are rays to be had for enough empty lanes.  An exchange that could move nothing is not started: a wave
            // whose lanes wait for rays only looks at the ring's cursors.
            const bool do_push = n_fin > 0 && (n_fin >= q.swap || n_trav == 0);
            uint32_t r_avail = 0;
            if (n_emp > 0 || do_push) r_avail = q_load(ctl + kQrTail) - q_load(ctl + kQrHead);
            const int holes = n_emp + (do_push ? n_fin : 0);
            const int fill = (int)r_avail < holes ? (int)r_avail : holes;
            const bool do_pull = fill > 0 && (do_push || fill >= q.swap || fill == holes || n_trav < 32);
            if (do_push || do_pull || n_trav == 0) {
                if (STATS) dg[1]++;
                if (do_push) {
                    uint32_t pos;
                    if (q_reserve(ctl + kQsHead, (uint32_t)n_fin, lane, pos, ring_n)) {
                        if (fin) {
                            QPath p;
                            p.org = ray.org; p.dir = ray.dir;
                            p.a0 = tv.best.t; p.a1 = tv.best.u; p.a2 = tv.best.v; p.a3 = __builtin_bit_cast(float, tv.best.prim);
                            p.L = cL; p.T = cT; p.rng = crng;
                            p.depth_flags = cdepth | (tv.redo ? kQRedoBit : 0u);
                            p.my_w = cw;
                            q_write(shade_ring, pos + lane_rank(fin_m), p, ctl, rl);
                            has = false;
                        }
                        if (STATS) dg[2] += (unsigned)n_fin;
                    }
                }
                const unsigned long long emp_m = __ballot(!has);
                const uint32_t want = (uint32_t)__popcll(emp_m);
                uint32_t got = 0, pos = 0;
                if (want && r_avail) got = q_claim(ctl + kQrHead, want, 1u, lane, pos);
                if (!has && lane_rank(emp_m) < got) {
                    QPath p;
                    q_read(ready_ring, pos + lane_rank(emp_m), p, ctl, rl);
                    ray.org = p.org; ray.dir = p.dir;
                    cL = p.L; cT = p.T; crng = p.rng; cdepth = p.depth_flags & 0xffffu; cw = p.my_w;
                    const bool primary = cdepth == 0u;
                    ray.tnear = primary ? 0.0f : 1e-4f;
                    ray.tfar = primary ? __builtin_inff() : FLT_MAX;
                    // trav_begin with the 1/d the S-wave computed (same division, same bits)
                    tv.inv = ptm::mk(p.a0, p.a1, p.a2);
                    tv.best.t = FLT_MAX; tv.best.u = 0.0f; tv.best.v = 0.0f; tv.best.prim = -1;
                    tv.cur = sv.root_ref;
                    tv.sp = 1;
                    tv.redo = (p.depth_flags & kQRedoBit) != 0u;
                    if (tv.redo) tv.cur = DONE;          // traced on the caller's tree by the S-wave that shades it
                    const uint32_t oct = (tv.inv.x < 0.0f ? 1u : 0u) | (tv.inv.y < 0.0f ? 2u : 0u) | (tv.inv.z < 0.0f ? 4u : 0u);
                    tv.node_off = oct * sv.oct_stride;
                    has = true;
                    n_segs++;
                }
                if (__ballot(has && !(tv.cur == DONE && (!POSTPONE || pend == DONE))) == 0ull) {
                    // nothing to traverse (lanes are empty, or hold finished paths the shade ring has no room for yet):
                    // finished, or wait for the S-waves — never longer than the watchdog allows
                    if (STATS) dg[11]++;
                    if (q_load(ctl + kQFeedsOpen) == 0u && (int32_t)q_load(ctl + kQLive) <= 0) break;
                    if (q_load(ctl + kQError) != 0u) break;
                    const unsigned long long now = wall_clock64();
                    if (idle_since == 0ull) idle_since = now;
                    if (now - idle_since > kQWatchdogTicks) { q_flag_error(ctl); break; }
                    __builtin_amdgcn_s_sleep(8);
                } else {
                    idle_since = 0ull;
                }
            }
        }
    } else {
        // ---------------------------------------------------------------------------------- shading role
        const int swave = wave - kQT;
        // a finished path waits in the shade ring, and its T-lane successor waits for a ray, until an S-wave gets to it: the few
        // S-waves go first when they have something to issue
        __builtin_amdgcn_s_setprio(PT_Q_SPRIO);
        WorkFeed feed;
        feed_init(feed, rp);
        bool feed_open = true;
        ptd::SceneView sv_ref = sv;
        sv_ref.nodes = scn.ref_nodes; sv_ref.root_ref = scn.ref_root_ref; sv_ref.node_stride = sizeof(DNode);
        sv_ref.oct_stride = 0; sv_ref.top_nodes = nullptr; sv_ref.top_count = 0; sv_ref.fixed_order = 0;
        int32_t* redo_stk = scn.redo_stack + ((size_t)(blockIdx.x * kQS + swave) * (size_t)scn.redo_cap) * 64 + lane;
        for (;;) {
            bool did = false;
            // ---- start paths (main.cu:32-44) while the workgroup is below its target
            if (feed_open && (int32_t)q_load(ctl + kQLive) + 64 <= q.target &&
                q_load(ctl + kQrTail) - q_load(ctl + kQrHead) + 64u <= ring_n) {
                feed_reserve(feed, rp, work_counter, lane);
                const uint32_t avail = feed.end - feed.cur;
                if (avail == 0) {
                    feed_open = false;
                    if (lane == 0) __hip_atomic_fetch_sub(ctl + kQFeedsOpen, 1u, __ATOMIC_RELAXED, PT_Q_WG);
                } else {
                    const uint32_t n = avail < 64u ? avail : 64u;
                    // the workgroup never holds more than q.target paths (what keeps the rings from filling up for good):
                    // count them in first, step back if another S-wave got there at the same time
                    uint32_t room = 0;
                    if (lane == 0) {
                        const int32_t before = (int32_t)__hip_atomic_fetch_add(ctl + kQLive, n, __ATOMIC_RELAXED, PT_Q_WG);
                        room = before + (int32_t)n <= q.target ? 1u : 0u;
                        if (!room) __hip_atomic_fetch_sub(ctl + kQLive, n, __ATOMIC_RELAXED, PT_Q_WG);
                    }
                    room = __builtin_amdgcn_readfirstlane(room);
                    if (room) {
                        // the paths first, their places in the ring second: a T-wave may claim a place the moment it is
                        // reserved and would then wait for all of this
                        QPath p;
                        p.org = ptm::mk(0, 0, 0); p.dir = ptm::mk(0, 0, 1); p.a0 = p.a1 = p.a2 = p.a3 = 0.0f;
                        p.L = ptm::mk(0, 0, 0); p.T = ptm::mk(1, 1, 1); p.rng.state = 0; p.rng.inc = 1; p.depth_flags = 0; p.my_w = 0;
                        if ((uint32_t)lane < n) {
                            const PathStart ps = start_path(rp, feed.region, feed.cur + (uint32_t)lane);
                            p.org = ps.ray.org; p.dir = ps.ray.dir;
                            const ptm::V3 inv = ptm::mk(1.0f / p.dir.x, 1.0f / p.dir.y, 1.0f / p.dir.z);     // trav_begin
                            p.a0 = inv.x; p.a1 = inv.y; p.a2 = inv.z; p.a3 = 0.0f;
                            p.rng = ps.rng;
                            const bool finite = __builtin_isfinite(inv.x) && __builtin_isfinite(inv.y) && __builtin_isfinite(inv.z);
                            p.depth_flags = (fbk && !finite) ? kQRedoBit : 0u;
                            p.my_w = ps.sample_index;
                        }
                        uint32_t pos = 0;
                        if (q_reserve(ctl + kQrHead, n, lane, pos, ring_n)) {
                            if ((uint32_t)lane < n) {
                                q_write(ready_ring, pos + (uint32_t)lane, p, ctl, rl);
                                n_paths++;
                            }
                            feed.cur += n;
                            did = true;
                            if (STATS) dg[10]++;
                        } else if (lane == 0) {             // no room in the ring after all: these items are started later
                            __hip_atomic_fetch_sub(ctl + kQLive, n, __ATOMIC_RELAXED, PT_Q_WG);
                        }
                    }
                }
            }
            // ---- shade a batch of finished segments (radiance.cuh:26-75)
            {
                const uint32_t r_avail = q_load(ctl + kQrTail) - q_load(ctl + kQrHead);
                const uint32_t at_least = ((int32_t)r_avail < q.low) ? 1u : 64u;
                uint32_t pos = 0;
                const uint32_t got = q_claim(ctl + kQsHead, 64u, at_least, lane, pos);
                if (got) {
                    did = true;
                    if (STATS) { dg[7]++; dg[9] += got; }
                    const bool mine = (uint32_t)lane < got;
                    QPath p;
                    p.org = ptm::mk(0, 0, 0); p.dir = ptm::mk(0, 0, 1); p.a0 = p.a1 = p.a2 = p.a3 = 0.0f;
                    p.L = ptm::mk(0, 0, 0); p.T = ptm::mk(1, 1, 1); p.rng.state = 0; p.rng.inc = 1; p.depth_flags = 0; p.my_w = 0;
                    bool cont = false;
                    if (mine) {
                        q_read(shade_ring, pos + (uint32_t)lane, p, ctl, rl);
                        ptd::Ray ray;
                        ray.org = p.org; ray.dir = p.dir;
                        int depth = (int)(p.depth_flags & 0xffffu);
                        const bool primary = depth == 0;
                        ray.tnear = primary ? 0.0f : 1e-4f;
                        ray.tfar = primary ? __builtin_inff() : FLT_MAX;
                        ptd::Hit best;
                        best.t = p.a0; best.u = p.a1; best.v = p.a2; best.prim = __builtin_bit_cast(int32_t, p.a3);
                        if (PT_Q_CHECK && (best.prim < -1 || best.prim >= scn.num_prims || p.my_w >= rp.total_work)) {
                            // cannot happen; if it does, the frame is invalid (error flag) but no address is made from it
                            q_flag_error(ctl, 4u);
                            best.prim = -1;
                            p.my_w = 0;
                        }
                        if (fbk && (p.depth_flags & kQRedoBit)) {
                            // the closest hit of this ray depends on the visit order: traced the reference's way, to completion
                            ptd::TravStats st_redo;
                            st_redo.nodes = 0; st_redo.leaves = 0;
                            best = ptd::intersect<false, false>(sv_ref, ray, redo_stk, st_redo);
                            if (STATS) n_redo++;
                        }
                        if (best.prim < 0) {
                            p.L = p.L + p.T * sv.bg;
                        } else {
                            const ptd::Surface sf = ptd::make_surface<TRI_ONLY>(sv, ray, best);
                            cont = ptd::shade_and_bounce<DIFFUSE_ONLY, false>(sv, sf, ray, p.rng, p.L, p.T, depth, rp.rr_depth, nullptr);
                            depth++;
                            if (depth >= rp.max_depth) cont = false;
                        }
                        if (!cont) {
                            samples[p.my_w] = make_float4(p.L.x, p.L.y, p.L.z, 0.0f);
                        } else {
                            p.org = ray.org; p.dir = ray.dir;
                            const ptm::V3 inv = ptm::mk(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);   // trav_begin
                            p.a0 = inv.x; p.a1 = inv.y; p.a2 = inv.z; p.a3 = 0.0f;
                            const bool finite = __builtin_isfinite(inv.x) && __builtin_isfinite(inv.y) && __builtin_isfinite(inv.z);
                            p.depth_flags = (uint32_t)depth | ((fbk && !finite) ? kQRedoBit : 0u);
                        }
                    }
                    const unsigned long long cont_m = __ballot(mine && cont);
                    const uint32_t n_cont = (uint32_t)__popcll(cont_m);
                    if (lane == 0 && got > n_cont) __hip_atomic_fetch_sub(ctl + kQLive, got - n_cont, __ATOMIC_RELAXED, PT_Q_WG);
                    if (n_cont) {
                        // room in the ready ring: the T-waves take rays out of it whatever the S-waves do, and the workgroup
                        // never holds more paths than fit (q.target), so this wait ends
                        uint32_t rpos = 0, spins = 0;
                        bool placed;
                        while (!(placed = q_reserve(ctl + kQrHead, n_cont, lane, rpos, ring_n))) {
                            if (++spins > kQSpinLimit || q_load(ctl + kQError) != 0u) { q_flag_error(ctl); break; }
                            __builtin_amdgcn_s_sleep(2);
                        }
                        if (placed && mine && cont) q_write(ready_ring, rpos + lane_rank(cont_m), p, ctl, rl);
                    }
                }
            }
            if (!did) {
                if (!feed_open && q_load(ctl + kQFeedsOpen) == 0u && (int32_t)q_load(ctl + kQLive) <= 0) break;
                if (q_load(ctl + kQError) != 0u) break;
                const unsigned long long now = wall_clock64();
                if (idle_since == 0ull) idle_since = now;
                if (now - idle_since > kQWatchdogTicks) { q_flag_error(ctl); break; }
                __builtin_amdgcn_s_sleep(4);
            } else {
                idle_since = 0ull;
            }
        }
    }
    flush_counters<STATS>(counters, lane, n_paths, n_segs, st);
    dg[8] = STATS ? wave_sum(n_redo) : 0ull;                 // segments traced on the caller's tree (as v2's slot 12)
    if (lane == 0) {
        unsigned long long* slot = counter_slot(counters);
        const uint32_t err = q_load(ctl + kQError);
        if (err != 0u) atomicOr(&slot[15], (unsigned long long)err);           // the host turns this into PT_ERR_DEVICE
        if (STATS) {
            atomicAdd(&slot[4], dg[0]); atomicAdd(&slot[5], dg[1]); atomicAdd(&slot[6], dg[2]); atomicAdd(&slot[7], dg[3]);
            atomicAdd(&slot[8], dg[4]); atomicAdd(&slot[9], dg[5]); atomicAdd(&slot[10], dg[6]); atomicAdd(&slot[11], dg[7]);
            atomicAdd(&slot[12], dg[8]); atomicAdd(&slot[13], dg[9]); atomicAdd(&slot[14], dg[10]);
        }
    }
}

}  // namespace ptk
