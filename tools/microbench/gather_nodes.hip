// Microbenchmark: cost of fetching random 64-byte BVH nodes with one ray per lane.
//   A: every lane loads its own node with 4 x global_load_dwordx4 (what inner_step does)
//   B: quad-cooperative: the 4 lanes of a quad load the 4 x 16 B pieces of ONE node per instruction (coalesced 64 B),
//      4 instructions cover the quad's 4 nodes; data is then exchanged inside the quad with DPP
//   C: like A but each lane loads only 16 B (lower bound for "one request per lane")
//   E: 32-byte nodes (a second table with a 32-B stride): 2 x global_load_dwordx4 per lane — what a node with conservatively
//      quantised child boxes would cost ("next candidate")
//   D: quad-cooperative through LDS: the same coalesced 64-B reads as B, but issued as LDS-DMA (global_load_lds_dwordx4:
//      lane l's 16 B land at tile + l*16), so that after 4 instructions quad q's member-m node sits at tile_m + q*64 and
//      every lane reads its own node back with ds_read_b128 — no DPP exchange, no selects
// argv[3] = HOT (0..7): in HOT of 8 steps the next node comes from a 256-node hot set (16 KB, L1-resident) instead of the
// whole table — a traversal revisits the top of the tree, and the kernel's measured L1 hit rate on bunny is 86 %.
// Table size and dependent-chain length are parameters; the address of step i+1 depends on the data of step i,
// like a traversal.   hipcc --offload-arch=gfx950 -O3 gather_nodes.hip -o gather_nodes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ nodes, uint32_t n_nodes, int steps, uint32_t* out, uint32_t hot) {
    __shared__ __attribute__((aligned(16))) float4 tiles[4 /*waves*/][4 /*instr*/][64];
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) % n_nodes;
    float acc = 0.f;
    const int q = threadIdx.x & 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = 0; s < steps; s++) {
        float4 a, b, c, d;
        if (MODE == 0) {
            const float4* p = nodes + (size_t)idx * 4;
            a = p[0]; b = p[1]; c = p[2]; d = p[3];
        } else if (MODE == 2) {
            const float4* p = nodes + (size_t)idx * 4;
            a = p[0]; b = a; c = a; d = a;
        } else if (MODE == 4) {
            const float4* p = nodes + (size_t)idx * 2;          // `nodes` is the 32-B table here; its second piece carries the index in .x
            a = p[0]; d = p[1]; b = a; c = a;
        } else if (MODE == 3) {
            const uint32_t i0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true);
            const uint32_t i1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xf, 0xf, true);
            const uint32_t i2 = __builtin_amdgcn_mov_dpp(idx, 0xaa, 0xf, 0xf, true);
            const uint32_t i3 = __builtin_amdgcn_mov_dpp(idx, 0xff, 0xf, 0xf, true);
            typedef __attribute__((address_space(3))) void lds_void;
            typedef __attribute__((address_space(1))) const void glb_void;
            __builtin_amdgcn_global_load_lds((glb_void*)(nodes + (size_t)i0 * 4 + q), (lds_void*)&tiles[wave][0][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(nodes + (size_t)i1 * 4 + q), (lds_void*)&tiles[wave][1][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(nodes + (size_t)i2 * 4 + q), (lds_void*)&tiles[wave][2][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(nodes + (size_t)i3 * 4 + q), (lds_void*)&tiles[wave][3][0], 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float4* mine = &tiles[wave][q][(lane >> 2) * 4];
            a = mine[0]; b = mine[1]; c = mine[2]; d = mine[3];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is rewritten by the next step's DMA
        } else {
            // node index of quad lane k, broadcast inside the quad
            const uint32_t i0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
            const uint32_t i1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xf, 0xf, true);   // [1,1,1,1]
            const uint32_t i2 = __builtin_amdgcn_mov_dpp(idx, 0xaa, 0xf, 0xf, true);   // [2,2,2,2]
            const uint32_t i3 = __builtin_amdgcn_mov_dpp(idx, 0xff, 0xf, 0xf, true);   // [3,3,3,3]
            const float4 m0 = nodes[(size_t)i0 * 4 + q];     // piece q of node 0
            const float4 m1 = nodes[(size_t)i1 * 4 + q];
            const float4 m2 = nodes[(size_t)i2 * 4 + q];
            const float4 m3 = nodes[(size_t)i3 * 4 + q];
            // lane q needs piece j of node q = register m_q of lane j.  Select m_q locally, then fetch from lane j.
            const float4 mine = q == 0 ? m0 : q == 1 ? m1 : q == 2 ? m2 : m3;   // piece q of MY node (diagonal)
            // rotation r: lane q gets piece (q+r)%4 of its node from lane (q+r)%4, which must offer m_{q} ... the offered
            // register depends on the READER, so each source offers the register of the lane r positions "before" it.
            auto offer = [&](int r) { const int t = (q - r) & 3; return t == 0 ? m0 : t == 1 ? m1 : t == 2 ? m2 : m3; };
            const float4 o1 = offer(1), o2 = offer(2), o3 = offer(3);
#define ROT4(v, ctrl) make_float4( \
    __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (v).x), ctrl, 0xf, 0xf, true)), \
    __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (v).y), ctrl, 0xf, 0xf, true)), \
    __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (v).z), ctrl, 0xf, 0xf, true)), \
    __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (v).w), ctrl, 0xf, 0xf, true)))
            const float4 g1 = ROT4(o1, 0x39);   // quad_perm [1,2,3,0]: lane q reads lane q+1
            const float4 g2 = ROT4(o2, 0x4e);   // [2,3,0,1]
            const float4 g3 = ROT4(o3, 0x93);   // [3,0,1,2]
            // g_r = piece (q+r)%4 of my node; put pieces in order
            const float4 p0 = q == 0 ? mine : q == 1 ? g3 : q == 2 ? g2 : g1;
            const float4 p1 = q == 1 ? mine : q == 2 ? g3 : q == 3 ? g2 : g1;
            const float4 p2 = q == 2 ? mine : q == 3 ? g3 : q == 0 ? g2 : g1;
            const float4 p3 = q == 3 ? mine : q == 0 ? g3 : q == 1 ? g2 : g1;
            a = p0; b = p1; c = p2; d = p3;
        }
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.y + d.z + d.w);   // every dword is used
        const uint32_t h = mix(__builtin_bit_cast(uint32_t, d.x) + idx);  // dependent chain (d.x holds the node's own index)
        idx = ((h & 7u) < hot) ? ((h >> 3) & 255u) : (h >> 3) % n_nodes;
    }
    out[blockIdx.x * 256 + threadIdx.x] = idx + (uint32_t)acc;
}

int main(int argc, char** argv) {
    const uint32_t n_nodes = argc > 1 ? atoi(argv[1]) : 576186;   // bunny: 36.9 MB of nodes
    const int steps = argc > 2 ? atoi(argv[2]) : 64;
    const uint32_t hot = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<float> h((size_t)n_nodes * 16);
    for (uint32_t i = 0; i < n_nodes; i++)
        for (int k = 0; k < 16; k++) h[(size_t)i * 16 + k] = (k == 12) ? __builtin_bit_cast(float, i * 2654435761u) : float(i % 97) + k;
    std::vector<float> h32((size_t)n_nodes * 8);
    for (uint32_t i = 0; i < n_nodes; i++)
        for (int k = 0; k < 8; k++) h32[(size_t)i * 8 + k] = (k == 4) ? __builtin_bit_cast(float, i * 2654435761u) : float(i % 97) + k;
    float4* d32;
    CHECK(hipMalloc(&d32, h32.size() * 4));
    CHECK(hipMemcpy(d32, h32.data(), h32.size() * 4, hipMemcpyHostToDevice));
    float4* d; uint32_t* out;
    const int blocks = 256 * 6;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&out, blocks * 256 * 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<uint32_t> r0(blocks * 256), r1(blocks * 256);
    printf("hot steps: %u of 8\n", hot);
    for (int mode = 0; mode < 5; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) gather<0><<<blocks, 256>>>(d, n_nodes, steps, out, hot);
            else if (mode == 1) gather<1><<<blocks, 256>>>(d, n_nodes, steps, out, hot);
            else if (mode == 2) gather<2><<<blocks, 256>>>(d, n_nodes, steps, out, hot);
            else if (mode == 3) gather<3><<<blocks, 256>>>(d, n_nodes, steps, out, hot);
            else gather<4><<<blocks, 256>>>(d32, n_nodes, steps, out, hot);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CHECK(hipMemcpy(mode == 0 ? r0.data() : r1.data(), out, blocks * 256 * 4, hipMemcpyDeviceToHost));
        const double fetches = double(blocks) * 256 * steps;
        printf("mode %d (%s): %.3f ms, %.2f G node fetches/s, %.1f cycles per wave-step per CU @2.4GHz\n", mode,
               mode == 0 ? "own node, 4 x dwordx4" : mode == 1 ? "quad-cooperative + DPP transpose" : mode == 2 ? "own node, 1 x dwordx4 only" : mode == 3 ? "quad-cooperative LDS-DMA + ds_read" : "32-B nodes, 2 x dwordx4",
               best, fetches / best / 1e6, best * 1e-3 * 2.4e9 / (fetches / 64 / 256));
        if (mode == 1 || mode == 3) { size_t bad = 0; for (size_t i = 0; i < r0.size(); i++) bad += r0[i] != r1[i]; printf("  mode %d vs mode 0 mismatches: %zu\n", mode, bad); }
    }
    return 0;
}
