/*
 * pt_host.h — C ABI of the host-side scene pipeline (libpt_host.so, plain C++17,
 * no GPU dependency).  It keeps the reference's host surface in front of the
 * device boundary (pt_api.h):
 *
 *   reference                                   here
 *   ------------------------------------------  --------------------------------
 *   parse_scene(path)   parse_scene.cpp:862-877  pt_host_scene_load_xml
 *   ParsedScene         parse_scene.h:114-121    builder calls (set_camera/add_*)
 *   Scene::Scene        scene.cpp:11-153         pt_host_scene_finalize (flatten,
 *     construct_bvh     bvh.cu:16-54               per-primitive AABBs, median-split BVH)
 *   compute_camera_ray_data  camera.cuh:28-43    pt_host_camera_ray_data
 *
 * plus a small binary container for parsed scenes (.pts) so that scenes can
 * travel without the XML/OBJ/PLY sources.
 */
#ifndef PT_HOST_H
#define PT_HOST_H

#include "pt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_host_scene pt_host_scene;

/* ParsedCamera (parse_scene.h:9-15) + sample count (parse_scene.h:120) */
typedef struct pt_camera {
    float   lookfrom[3];
    float   lookat[3];
    float   up[3];
    float   vfov;
    int32_t width, height;
    int32_t spp;
} pt_camera;

enum {
    PT_BVH_SORT_TOTAL = 0,      /* centroid on split axis, ties broken by primitive id: host-independent */
    PT_BVH_SORT_REFERENCE = 1   /* bvh.cu:34-37 comparator as is (ties resolved by this libstdc++'s std::sort) */
};

int pt_host_scene_new(pt_host_scene** out);
int pt_host_scene_load_xml(const char* path, pt_host_scene** out);
int pt_host_scene_load_pts(const char* path, pt_host_scene** out);
int pt_host_scene_save_pts(const pt_host_scene* s, const char* path);
int pt_host_scene_destroy(pt_host_scene* s);

int pt_host_scene_set_camera(pt_host_scene* s, const pt_camera* cam);
int pt_host_scene_get_camera(const pt_host_scene* s, pt_camera* cam);
int pt_host_scene_set_background(pt_host_scene* s, const float rgb[3]);
int pt_host_scene_add_material(pt_host_scene* s, const pt_material* m);                 /* returns id >= 0, or -status */
int pt_host_scene_add_point_light(pt_host_scene* s, const float position[3], const float intensity[3]);
/* radiance == NULL: not emissive.  Return shape id >= 0 or -status. */
int pt_host_scene_add_sphere(pt_host_scene* s, const float center[3], float radius, int material_id,
                             const float* radiance);
/* normals == NULL: area-weighted vertex normals are computed as compute_normals.cpp:13-51 does
 * (including its unit_angle quirk). */
int pt_host_scene_add_mesh(pt_host_scene* s, const float* positions, int num_vertices,
                           const int32_t* indices, int num_faces, const float* normals,
                           int material_id, const float* radiance);

/* Scene::Scene: flatten + BVH.  Must be called before get_desc.  Idempotent. */
int pt_host_scene_finalize(pt_host_scene* s, int bvh_sort_mode);
/* Pointers in *out stay valid until the scene is destroyed or re-finalized. */
int pt_host_scene_get_desc(const pt_host_scene* s, pt_scene_desc* out);
int pt_host_scene_bvh_depth(const pt_host_scene* s);       /* computeMaxDepth, bvh.cu:56-65 */

/* compute_camera_ray_data: out = origin, top_left_corner, horizontal, vertical */
void pt_host_camera_ray_data(const pt_camera* cam, int width, int height, float out12[12]);
/* Convenience: fill cam_* / width / height / spp / seed=1984 of *p from the camera (other fields zeroed). */
void pt_host_default_params(const pt_camera* cam, int width, int height, int spp, pt_render_params* p);

/* Image output (the reference never saves its framebuffer: main.cu:258-266).  fb = [height][width][3] floats, row 0 = top.
 * PFM: lossless linear floats.  PPM: the reference's display encoding (sqrt gamma, 8 bit; opengl_display.cpp:104-111). */
int pt_host_write_pfm(const char* path, const float* fb, int width, int height);
int pt_host_write_ppm(const char* path, const float* fb, int width, int height);

const char* pt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
