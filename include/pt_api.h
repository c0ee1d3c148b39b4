/*
 * pt_api.h — C ABI of the MI355X path-tracing core (libpt_hip.so).
 *
 * This is the drop-in boundary for the reference's offline render path.  The
 * reference (jayHuggie/PathTracer_CUDA_Interactive) has no plugin/FFI API; the
 * seam is "upload the flattened Scene, seed the RNG, launch `render`, sync":
 *
 *   reference call site                       replaced by
 *   ----------------------------------------  ---------------------------------
 *   GPUScene::copyFrom        scene.h:73-119   pt_scene_create
 *   copyTriangleMeshToDevice  shape.cuh:48-59  pt_scene_create (mesh arrays)
 *   copyBVHNodesToDevice2     bvh.cu:83-97     pt_scene_create (node array)
 *   setup_rand<<<>>>          main.cu:54-62,234  (folded into pt_render: O(1) PCG init)
 *   render<<<>>> + sync       main.cu:30-52,258-260   pt_render / pt_render_async
 *   render_progressive<<<>>>  main.cu:64-89,333       pt_render_accumulate
 *   GPUScene::free            scene.h:144-171  pt_scene_destroy
 *   checkCudaErrors -> exit   bbox.cuh:7-17    int status + pt_last_error()
 *
 * All structs are plain C PODs mirroring the reference's host `Scene`
 * (scene.h:17-35): shapes, meshes, materials, lights, BVH nodes.  The caller
 * owns every input array; pt_scene_create copies what it needs (and re-lays it
 * out for the GPU), so inputs may be freed right after it returns.
 *
 * No torch / C++ types cross this boundary.  Thread-compatible: no hidden
 * global state except the thread-local error string.  A scene handle owns one
 * set of scratch buffers, so its renders must not overlap: calls on one handle
 * are issued from one thread at a time, and a render on another stream is
 * enqueued only after the previous one has finished (different handles are
 * independent).
 */
#ifndef PT_API_H
#define PT_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_API_VERSION 1

/* status codes (0 = ok).  The reference exits the process instead (bbox.cuh:9-17). */
enum {
    PT_OK = 0,
    PT_ERR_INVALID_ARG = 1,
    PT_ERR_BAD_SCENE = 2,      /* inconsistent ids / topology / BVH deeper than the stack cap */
    PT_ERR_DEVICE = 3,         /* HIP runtime error */
    PT_ERR_NO_DEVICE = 4,
    PT_ERR_IO = 5,
    PT_ERR_PARSE = 6,
    PT_ERR_UNSUPPORTED = 7
};

/* shape.cuh:61  enum ShapeType {SPHERE, TRIANGLE} */
enum { PT_SHAPE_SPHERE = 0, PT_SHAPE_TRIANGLE = 1 };
/* material.h:27 enum MaterialType { DIFFUSE, MIRROR, PLASTIC, PHONG } */
enum { PT_MAT_DIFFUSE = 0, PT_MAT_MIRROR = 1, PT_MAT_PLASTIC = 2, PT_MAT_PHONG = 3 };
/* light.h:17    enum LightType { POINTLIGHT, DIFFUSEAREALIGHT } */
enum { PT_LIGHT_POINT = 0, PT_LIGHT_DIFFUSE_AREA = 1 };

/* Shape (shape.cuh:63-69): tagged union of Sphere{mat,light,center,radius}
 * and Triangle{face_index, mesh_index}.  Flattened here (no union) so that
 * ctypes / cgo / JNI can bind it without tricks. */
typedef struct pt_shape {
    int32_t type;            /* PT_SHAPE_* */
    int32_t material_id;     /* sphere only (triangles take the mesh's)   */
    int32_t area_light_id;   /* sphere only; -1 = not emissive            */
    float   center[3];       /* sphere */
    float   radius;          /* sphere */
    int32_t face_index;      /* triangle: index into mesh indices         */
    int32_t mesh_index;      /* triangle: index into pt_scene_desc.meshes */
} pt_shape;

/* TriangleMesh (shape.cuh:28-41) minus the device pointers and the UVs
 * (UVs never influence the result: texture.h:69-71 is constant-colour only). */
typedef struct pt_mesh {
    int32_t material_id;
    int32_t area_light_id;       /* -1 = not emissive */
    int32_t num_vertices;
    int32_t num_faces;
    const float*   positions;    /* [num_vertices*3] */
    const int32_t* indices;      /* [num_faces*3]    */
    const float*   normals;      /* [num_vertices*3], REQUIRED (every shipped scene has them) */
} pt_mesh;

/* Material (material.h:8-36).  Textures are constant colours (texture.h:12-15). */
typedef struct pt_material {
    int32_t type;            /* PT_MAT_* */
    float   reflectance[3];
    float   eta;             /* PLASTIC index of refraction */
    float   exponent;        /* PHONG exponent              */
} pt_material;

/* Light (light.h:5-27).  Only DiffuseAreaLight::radiance is ever read by the
 * hot path (radiance.cuh:36-41); point lights are carried for fidelity. */
typedef struct pt_light {
    int32_t type;            /* PT_LIGHT_* */
    int32_t shape_id;        /* area light: emitting shape */
    float   radiance[3];     /* area: radiance; point: intensity */
    float   position[3];     /* point light only */
} pt_light;

/* BVHNode (bvh.cuh:7-15) minus the two unused device pointers. */
typedef struct pt_bvh_node {
    float   bmin[3];
    float   bmax[3];
    int32_t left;            /* -1 on leaves */
    int32_t right;           /* -1 on leaves */
    int32_t prim;            /* shape index on leaves, -1 on inner nodes */
} pt_bvh_node;

/* The flattened scene: what Scene (scene.h:17-35) holds after its ctor ran. */
typedef struct pt_scene_desc {
    int32_t num_shapes;     const pt_shape*    shapes;
    int32_t num_meshes;     const pt_mesh*     meshes;
    int32_t num_materials;  const pt_material* materials;
    int32_t num_lights;     const pt_light*    lights;
    int32_t num_nodes;      const pt_bvh_node* nodes;     /* 2*num_shapes-1, post-order (bvh.cu:16-54) */
    int32_t root;           /* = num_nodes-1 for the reference builder */
    float   background[3];
} pt_scene_desc;

/* Per-render inputs: what `render` takes by value (main.cu:30-31) plus the
 * constants the reference hard-codes (seed 1984 main.cu:61; MAX_DEPTH 50
 * radiance.cuh:12; RR after depth>5 radiance.cuh:68). */
typedef struct pt_render_params {
    float    cam_origin[3];          /* CameraRayData (camera.cuh:21-26) */
    float    cam_top_left[3];
    float    cam_horizontal[3];
    float    cam_vertical[3];
    int32_t  width, height;          /* full image size (u,v and pixel_index use these) */
    int32_t  spp;                    /* samples rendered by this call */
    int32_t  row_begin, row_end;     /* rows [row_begin,row_end) rendered by this call; 0,0 = all.  Multi-GPU shard. */
    int32_t  row_stride;             /* 0/1 = contiguous rows; k>1 = rows row_begin, row_begin+k, ... < row_end (interleaved bands of 1 row) */
    uint64_t seed;                   /* PCG seed (reference literal: 1984) */
    int32_t  max_depth;              /* 0 -> 50 */
    int32_t  rr_depth;               /* Russian roulette when depth > rr_depth; <0 -> 5 */
    int32_t  sample_offset;          /* first absolute sample index of this call (progressive) */
    int32_t  stream_stride;          /* PCG stream = pixel_index*stream_stride + sample_offset + s; 0 -> spp */
    int32_t  traversal;              /* PT_TRAVERSAL_* */
    int32_t  flags;                  /* PT_RENDER_* bits; 0 = the reference's estimator */
} pt_render_params;

/* pt_render_params.flags */
enum {
    /* Next-event estimation — an EXTENSION, the reference samples no light (its point lights are parsed,
     * light.h:5-8, and never used; its occlusion query is dead code, scene.h:306-330).  Every non-specular hit samples one
     * entry of scene.lights[] uniformly (area lights: a uniform point on the emitting primitive; point lights) and
     * connects to it with a shadow ray; emission found by BSDF sampling then counts only on camera rays and after
     * specular bounces.  Same expectation as the reference's estimator when all light comes from area lights, less
     * noise; point lights start to light the scene.  Off by default; every parity and roofline number is taken without it. */
    PT_RENDER_NEE = 1
};

enum {
    PT_TRAVERSAL_DEFAULT = 0,        /* library picks (currently EXACT) */
    PT_TRAVERSAL_EXACT   = 1,        /* visit exactly the nodes the reference visits (no closest-t pruning, scene.h:278-297) */
    PT_TRAVERSAL_PRUNED  = 2         /* skip subtrees whose box entry is beyond the closest hit; same image, fewer visits */
};

/* Work counters of the last pt_render* call on a scene (device-side 64-bit
 * sums).  `segments` = number of intersect() calls = the "samples" of the
 * Msamples/s metric. */
typedef struct pt_counters {
    uint64_t paths;
    uint64_t segments;
    uint64_t node_visits;      /* inner-node pops actually executed by the kernel */
    uint64_t leaf_tests;       /* primitive tests actually executed */
    double   kernel_ms;        /* hipEvent time of the trace kernel(s) of the last call */
    double   resolve_ms;       /* hipEvent time of the per-pixel resolve kernel(s) */
} pt_counters;

typedef struct pt_scene pt_scene;    /* opaque: owns all device memory of one scene on one GPU */

/* Create a device scene on the CURRENT HIP device.  Validates ids/topology
 * (PT_ERR_BAD_SCENE instead of device printf: scene.h:260-263). */
int pt_scene_create(const pt_scene_desc* desc, pt_scene** out);
int pt_scene_destroy(pt_scene* scene);

/* Blocking render.  `fb` receives rows x width x 3 floats, rows = the rows
 * selected by (row_begin,row_end,row_stride), packed in increasing row order;
 * fb[(r*W+i)*3+c], linear radiance, row 0 = top (main.cu:35,50).
 * fb_on_device != 0: fb is a device pointer on the scene's GPU. */
int pt_render(pt_scene* scene, const pt_render_params* p, float* fb, int fb_on_device);

/* Same, enqueued on a caller-provided hipStream_t (NULL = default stream); fb
 * must be a device pointer; returns without synchronising.  What the caller can
 * observe — fb, and accum_dev below — is written in the order of that stream.  The
 * trace kernel itself reads only the scene and writes scratch memory of the handle,
 * so with "frames_in_flight" > 1 (option, default 2) a frame runs on a stream of the
 * handle's own — its resolve behind an event of the caller's stream, the caller's
 * stream behind its resolve — and the trace kernel of call k+1 fills the compute
 * units while the last paths of call k drain; same bits either way. */
int pt_render_async(pt_scene* scene, const pt_render_params* p, float* fb_dev, void* hip_stream);

/* Progressive accumulation (render_progressive, main.cu:64-89): adds the SUM
 * of p->spp new samples (absolute indices sample_offset..) into accum_dev
 * (same layout as fb; overwritten when sample_offset == 0).  Async on stream. */
int pt_render_accumulate(pt_scene* scene, const pt_render_params* p, float* accum_dev, void* hip_stream);

int pt_get_counters(pt_scene* scene, pt_counters* out);   /* synchronises the scene's last stream */

/* HIP-event times of the last render calls on the scene, oldest first: kernel_ms[k] / resolve_ms[k] of up to max_frames calls,
 * *n_out of them written.  The library keeps the events of the last "timing_frames" calls (option, default 1), so a caller
 * may enqueue many frames on a stream without a host sync in between and read every frame's kernel time afterwards
 * (bench.py's timed loop).  Synchronises the scene's last stream. */
int pt_get_frame_times(pt_scene* scene, int max_frames, double* kernel_ms, double* resolve_ms, int* n_out);

/* BVH construction on the device — an alternative producer of `pt_scene_desc.nodes` to the host's
 * construct_bvh (bvh.cu:16-54: object-median split, O(N log^2 N), 10-57 s start-up in README.md:123,132).
 *   PT_BVH_DEVICE_LBVH  Morton order + Karras hierarchy
 *   PT_BVH_DEVICE_SAH   Morton order + per-node surface-area-cost cut, top-down
 * desc->nodes / desc->root are ignored.  out_nodes receives 2*num_shapes-1 nodes in the reference's layout (HOST
 * memory: the same array can be handed to pt_scene_create and to a CPU checker); *out_root the root's index;
 * *out_depth (optional) the depth with leaves counting 1 (computeMaxDepth, bvh.cu:56-65); *out_build_ms (optional)
 * the device time from primitive boxes to finished nodes (uploads and the copy back to the host excluded).
 * Such trees visit fewer nodes per ray than the reference's; closest hits — and so images — are the same except where
 * two primitives tie on t (the first one VISITED wins, scene.h:270).  Parity and roofline numbers use the reference tree. */
enum { PT_BVH_DEVICE_LBVH = 0, PT_BVH_DEVICE_SAH = 1 };
int pt_bvh_build_device(const pt_scene_desc* desc, int method, pt_bvh_node* out_nodes, int32_t* out_root,
                        int32_t* out_depth, double* out_build_ms);

/* The builder behind the library's internal tree (pt_scene_create), exposed so that the tree can be inspected, checked by
 * a CPU oracle, or handed in as the caller's own: top-down, every cut along x, y and z considered at every node,
 * smallest  SA(left) n_left + SA(right) n_right  wins.  Runs on the HOST (no GPU needed), deterministic.
 * Input: the LEAF boxes of desc->nodes (one leaf per shape; the inner nodes of desc are ignored).  Output as for
 * pt_bvh_build_device; out_build_ms = host wall time.  The tree may be deeper than the reference's 64-entry stack
 * allows (scene.h:251) for adversarial input — *out_depth tells; pt_scene_create rejects such a tree as a CALLER's. */
int pt_bvh_build_sweep(const pt_scene_desc* desc, pt_bvh_node* out_nodes, int32_t* out_root, int32_t* out_depth,
                       double* out_build_ms);

/* The same builder run on the GPU (csrc/pt_sweep_build.hip: level-synchronous segmented scans and reductions): the SAME tree,
 * byte for byte — pt_scene_create uses it from 4,096 shapes up.  out_build_ms = device time (sorts to finished nodes, without the
 * upload of the leaf boxes and the copy back).  PT_ERR_UNSUPPORTED for input that runs into the host builder's depth guard
 * (2 log2 n + 16 levels): only the host builder holds the median-cut fallback for such branches. */
int pt_bvh_build_sweep_device(const pt_scene_desc* desc, pt_bvh_node* out_nodes, int32_t* out_root, int32_t* out_depth,
                              double* out_build_ms);

/* Tuning knobs (all optional; none of them changes a bit of the rendered image):
 *   "kernel"        2 (default) decoupled traversal/shading scheduler, 1 segment-synchronous wavefront kernel, 3 paths regrouped
 *                   across the waves of a workgroup through LDS rings (csrc/pt_kernel_q.h; exact traversal without PT_RENDER_NEE,
 *                   anything else runs on 2; "q_target" / "q_swap" / "q_low" are its schedule knobs, 0 = automatic)
 *   "v2_thresh" / "v2_inner" / "v2_minw"   scheduler variant of kernel 2; 0 = automatic (by scene residency).
 *                   v2_inner: < 0 vote burst of -n steps; 1..9 n inner steps + 1 leaf step per burst; >= 100 encodes
 *                   rounds*100 + inner*10 + leaf steps (162 = 6 inner + 2 leaf steps); 1000 + burst = the same burst with
 *                   leaves set aside and tested together (internal tree only).  Only compiled-in variants are
 *                   accepted (PT_ERR_INVALID_ARG otherwise); every variant renders the same bits.
 *   "octants"       1 (default) keep 8 ray-octant node tables in LDS for very small scenes, 0 = one table
 *   "fast_tree"     1 (default) traversal (exact and pruned) on the library's INTERNAL tree where pt_scene_create kept one,
 *                   0 = on the caller's tree.  Same image either way, bit for bit: the reference never prunes, so a leaf is tested iff the
 *                   ray hits the leaf's own box (nested boxes) — any tree over the caller's leaf boxes tests the same leaves;
 *                   ties on t are settled in the caller's visit order (one box test at the node of the caller's tree where
 *                   the two leaves' paths part), rays with a zero direction component are traced on the caller's tree.  The
 *                   internal tree (pt_bvh_build_sweep) is kept when probe rays visit >= 10 % fewer nodes in it, the caller's
 *                   boxes nest, and the scene has 16+ primitives.  DESIGN.md §12.
 *   "top_cache"     1 (default) scenes read from global memory keep the most-visited top nodes of the tree in LDS, 0 = all from memory
 *   "lds_budget_kb" LDS per block for traversal stacks + that cache (0 = 26: 6 resident blocks per CU)
 *   "chunk"         work items a wave reserves per atomic, 64..256 (0 = automatic: 128 for big launches)
 *   "xcd_regions"   0 (default) 8 row bands with XCD affinity, 1 = a single work queue
 *   "item_order"    1 (default) a band is worked through row by row (all samples of a row first), 0 = sample by sample
 *   "force_global"  1 = never stage the scene in LDS
 *   "blocks_per_cu" persistent blocks per CU (0 = occupancy query; at most 32)
 *   "timing_frames" render calls whose HIP events are kept for pt_get_frame_times (0 .. 4096, default 1).  0 = no timing events
 *                   at all (pt_counters.kernel_ms / resolve_ms read 0): four event records less per frame, which an interactive
 *                   loop of 2-spp frames feels (cbox 640x480, host sync per frame: 3,250 -> 3,540 frames/s; scene1 5,300 -> 6,110)
 *   "frames_in_flight" 1 .. 4 (default 2): sets of per-frame scratch memory (per-sample buffer, work and statistics counters)
 *                   the handle rotates through; with more than one, consecutive render calls overlap as described at
 *                   pt_render_async (frames that need several sample passes, and frames that find the GPU idle, run on the
 *                   caller's stream as with 1).  A frame that overlaps leaves its successor room on every CU: all but one of
 *                   the resident blocks with 2 slots (never slower than 1 slot, whatever the caller's submission pattern),
 *                   half of them with 3 or 4 (the fastest for an unbroken stream of frames — bunny 4.41 ms per frame against
 *                   4.52 with 2 and 4.91 with 1 — but a burst that ends leaves its last frame on half a chip);
 *                   an explicit "blocks_per_cu" is used as given.
 *                   kernel_ms of a frame that overlapped with its neighbours includes the time its blocks waited for theirs.
 *   "scratch_bytes" cap of the per-sample scratch buffer (0 = 8 GiB); larger jobs run in sample passes
 *   "stats"         1 = also count node visits / leaf tests (pt_counters), schedule diagnostics ("diag0".."diag7") and the
 *                   launch timeline ("diag8".."diag15", 10-ns ticks; "diag16".."diag271" per-wave histograms; tools/gpu_diag.py)
 * pt_scene_get_info keys: "grid", "lds_bytes", "lds_scene", "residency" (0 global, 1 LDS, 2 LDS + octant tables, 3 global + top of the tree in LDS), "top_nodes",
 * "passes", "occupancy", "blocks_per_cu", "num_cus", "bvh_depth", "scene_bytes", "num_inner_nodes", "device", "vgprs", "vgprs_pruned",
 * "fast_tree" (an internal tree exists), "fast_tree_on" (the next render uses it), "fast_tree_is_callers" (it is the caller's own
 * topology in internal form: the sweep tree did not win the probe), "fast_tree_depth",
 * "fast_tree_cost_permille" (probe-ray node visits, internal / caller's x 1000; 0 = none built), "stack_entries" (per lane),
 * "redo_segments" (with "stats": segments of the last frame traced on the caller's tree), "debug_reruns" (same for pt_debug_intersect),
 * "kernel" / "block_threads" (what the last render ran on), "frames_in_flight", "sweep_on_device" (the internal tree was built on the GPU),
 * "create_us0".."create_us6" (wall microseconds of pt_scene_create: total, primitive records, caller's tree checked and re-laid,
 * internal tree built, ... re-laid, uploads + probe, tie tables). */
int pt_scene_set_option(pt_scene* scene, const char* key, int64_t value);
int pt_scene_get_info(pt_scene* scene, const char* key, int64_t* value);
/* Environment variables the library reads (A/B runs and debugging; results are the same whatever they say):
 *   PT_SWEEP_BUILD=host              pt_scene_create prepares big scenes on the host as in round 2 (default: on the device from 4,096 shapes)
 *   PT_SWEEP_SCANS=rocprim           the device sweep builder's box scans as rocPRIM scans-by-key (default: tiled)
 *   PT_SLOT_STREAM_PRIORITY=normal|high   priority of the handle's own streams (default: lowest) */

/* Deterministic fp32 helpers evaluated ON THE DEVICE, exported so tests can
 * pin device arithmetic against the CPU oracle bit-for-bit.
 * op: 0 sincos(x)->(out0=sin,out1=cos)   1 powf(x,y)->out0
 *     2 pcg32: stream = bit pattern of x, seed = bit pattern of y -> first two float draws */
int pt_debug_math(int op, const float* x, const float* y, float* out0, float* out1, int n);
/* Same functions (the very same source, csrc/pt_math.h) evaluated by the HOST half of the
 * library: lets CPU-only tests pin the shared arithmetic header without a GPU. */
int pt_debug_math_host(int op, const float* x, const float* y, float* out0, float* out1, int n);

/* Closest-hit query for explicit rays (tests / per-ray KATs).
 * rays: n x {org[3], dir[3], tnear, tfar}; out: n x {t,u,v} and prim id (-1 = miss). */
int pt_debug_intersect(pt_scene* scene, const float* rays, int n, int traversal,
                       float* out_tuv, int32_t* out_prim);

const char* pt_last_error(void);
int pt_api_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PT_API_H */
