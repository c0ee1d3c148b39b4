// pt_layout.h — device-side scene layout (what pt_scene_create builds from pt_scene_desc).
//
// The reference walks  BVHNode(56 B) -> child BVHNodes (for their boxes) -> Shape(28 B) ->
// TriangleMesh(80 B) -> int3 -> 3 x float3  with one dependent global load per arrow
// (scene.h:246-301,176-224).  Here every inner node carries BOTH child boxes and child
// references in one 64-byte record (one fetch per inner visit, leaves are not nodes at
// all), and every primitive is pre-gathered into one 48-byte record (one fetch per leaf).
// Topology, child order and visit order are exactly the reference's.
#pragma once
#include <stdint.h>

namespace ptl {

// Child reference: >= 0 -> index of an inner DNode; < 0 -> leaf, primitive index = ~ref.
constexpr int32_t kDone = (int32_t)0x80000000;   // traversal sentinel (never a valid ~prim)

struct alignas(16) DNode {          // 64 B
    float lmin[3], lmax[3];         // left child's AABB
    float rmin[3], rmax[3];         // right child's AABB
    int32_t left, right;            // child references
    int32_t pad0, pad1;
};
static_assert(sizeof(DNode) == 64, "DNode must be 64 bytes");

// Primitive record, 48 B = 3 x float4.
//   triangle: v = p0.xyz p1.xyz p2.xyz          sphere: v = center.xyz radius 0 0 0 0 0
//   info    : bit 31 = sphere flag, bits 0..30 = material id
//   light   : index into the emission table, or -1
struct alignas(16) DPrim {
    float v[9];
    int32_t info;
    int32_t light;
    int32_t pad;
};
static_assert(sizeof(DPrim) == 48, "DPrim must be 48 bytes");

struct alignas(16) DNormals {       // 48 B: vertex normals of a triangle (unused for spheres)
    float n[9];
    float pad[3];
};
static_assert(sizeof(DNormals) == 48, "DNormals must be 48 bytes");

struct alignas(16) DMaterial {      // 32 B
    int32_t type;
    float r, g, b;
    float eta, exponent;
    float pad0, pad1;
};
static_assert(sizeof(DMaterial) == 32, "DMaterial must be 32 bytes");

struct alignas(16) DEmission {      // 16 B: scene.lights[id] as radiance.cuh:36-41 reads it
    float r, g, b;
    int32_t is_area;                // light.type == DIFFUSEAREALIGHT
};

struct alignas(16) DLight {         // 32 B: scene.lights[k] as next-event estimation samples it (PT_RENDER_NEE)
    float r, g, b;                  // area: unused (the emitting primitive's own light entry decides); point: intensity
    int32_t type;                   // PT_LIGHT_POINT / PT_LIGHT_DIFFUSE_AREA
    float px, py, pz;               // point light position
    int32_t shape_id;               // area light: emitting primitive
};
static_assert(sizeof(DLight) == 32, "DLight must be 32 bytes");

struct SceneDev {
    const DNode* nodes;
    const DNode* nodes_oct;         // 8 octant-specialised copies [8][num_nodes] (small scenes only), else nullptr
    const DPrim* prims;
    const DNormals* normals;
    const DMaterial* materials;
    const DEmission* emission;
    const DLight* lights;           // [num_emission] entries of scene.lights[] in order (next-event estimation only)
    int32_t num_nodes, num_prims, num_materials, num_emission;
    int32_t root_ref;
    int32_t stack_cap;              // entries per lane needed (= BVH depth)
    float bg[3];
    // exact traversal on the internal tree: ties on t are settled in the caller's visit order (ref_path / ref_anc below), rays
    // with a zero direction component are traced on the caller's tree instead
    int32_t fallback;               // 1 = `nodes` is the internal tree: both mechanisms are on
    const DNode* ref_nodes;         // the caller's tree (plain layout, global memory)
    int32_t ref_root_ref;
    int32_t redo_cap;               // stack entries per lane of a rerun
    int32_t* redo_stack;            // [waves of the grid][redo_cap][64]
    int32_t fixed_order;            // `nodes` is the internal tree: left child first, stack_cap counts on it
    // ties on t are settled in the caller's tree's visit order (pt_trace.h: ref_visits_first)
    const unsigned long long* ref_path;   // [num_prims] turns from the root to the primitive's leaf in the caller's tree
    const int32_t* ref_anc;         // [num_prims][ref_levels] inner nodes of the caller's tree along that path
    int32_t ref_levels;
};

// Division of a number below 2^30 by a launch constant: q = (n * mul) >> shift, exact for every n < 2^30
// (mul = floor(2^shift / d) + 1 with shift = 30 + ceil(log2 d); the error term n*e / (d * 2^shift) stays below 1/d).
// A 32-bit integer division costs ~35 VALU instructions on gfx950, this one 3.
struct FastDiv {
    uint32_t mul, shift;
};
inline FastDiv make_fastdiv(uint32_t d) {
    uint32_t l = 0;
    while ((1ull << l) < d) l++;
    FastDiv f;
    f.shift = 30 + l;
    f.mul = (uint32_t)((1ull << f.shift) / (d ? d : 1)) + 1u;
    return f;
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t fastdiv(uint32_t n, FastDiv f) { return (uint32_t)(((uint64_t)n * f.mul) >> f.shift); }

struct RenderDev {
    float cam_origin[3], cam_top_left[3], cam_horizontal[3], cam_vertical[3];
    int32_t width, height;
    int32_t row_begin, row_step, num_rows;   // local row r -> image row row_begin + r*row_step
    int32_t spp_pass;               // samples traced by this launch
    int32_t sample_base;            // absolute index of this launch's first sample (sample_offset + pass offset)
    int32_t stream_stride;
    uint64_t seed;
    int32_t max_depth, rr_depth;
    uint32_t total_work;            // num_rows*width*spp_pass
    uint32_t npix;                  // num_rows*width
    int32_t num_regions;            // row bands with their own work counter (XCD affinity), 1..8
    int32_t rows_per_region;        // ceil(num_rows / num_regions)
    uint32_t chunk;                 // work items a wave reserves per atomic (64..256, multiple of 64)
    FastDiv div_width;              // / width
    FastDiv div_npix_full;          // / (rows_per_region * width): pixels of a full band
    FastDiv div_npix_last;          // / pixels of the last, shorter band (region `short_region`)
    int32_t short_region;           // index of the band with fewer than rows_per_region rows, or -1
    FastDiv div_spp;                // / spp_pass
    int32_t row_major;              // work item order inside a band: 0 = sample, row, column   1 = row, sample, column
};

// Bytes between the 8 ray-octant node tables in LDS.  A table of n 64-B nodes is a multiple of 64 B, so every table would
// start at the same two bank positions; one 16-B pad makes the table pitch an odd number of 16-B units, which puts the
// 8 tables at 8 different bank positions (lanes of a wave read nodes of different octants at once).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t oct_table_pitch(uint32_t num_nodes, uint32_t node_stride) {
    const uint32_t bytes = num_nodes * node_stride;
    return bytes + (((bytes / 16u) & 1u) ? 0u : 16u);
}

// LDS carve-up of the trace kernel (all offsets in bytes, 16-B aligned)
struct LdsPlan {
    uint32_t nodes_off, prims_off, normals_off, mats_off, emis_off, stack_off;
    uint32_t total;
    uint32_t top_count;             // residency 3: nodes [0, top_count) — the top of the tree — are also kept in LDS at nodes_off
};

}  // namespace ptl
