// C ABI of the host scene pipeline (include/pt_host.h).
#include <cstring>

#include "parsed_scene.h"

using namespace pth;

struct pt_host_scene {
    HostScene s;
};

static thread_local std::string g_err;

template <class F>
static int guard(F&& fn) {
    try {
        return fn();
    } catch (const Error& e) {
        g_err = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_err = e.what();
        return PT_ERR_INVALID_ARG;
    }
}

extern "C" {

const char* pt_host_last_error(void) { return g_err.c_str(); }

int pt_host_scene_new(pt_host_scene** out) {
    if (!out) return PT_ERR_INVALID_ARG;
    *out = new pt_host_scene();
    return PT_OK;
}

int pt_host_scene_destroy(pt_host_scene* s) {
    delete s;
    return PT_OK;
}

static int load_with(const char* path, pt_host_scene** out, void (*loader)(const std::string&, HostScene&)) {
    if (!path || !out) { g_err = "null argument"; return PT_ERR_INVALID_ARG; }
    *out = nullptr;
    pt_host_scene* h = new pt_host_scene();
    int rc = guard([&] { loader(path, h->s); return PT_OK; });
    if (rc != PT_OK) { delete h; return rc; }
    *out = h;
    return PT_OK;
}

int pt_host_scene_load_xml(const char* path, pt_host_scene** out) { return load_with(path, out, load_xml); }
int pt_host_scene_load_pts(const char* path, pt_host_scene** out) { return load_with(path, out, load_pts); }

int pt_host_scene_save_pts(const pt_host_scene* s, const char* path) {
    if (!s || !path) { g_err = "null argument"; return PT_ERR_INVALID_ARG; }
    return guard([&] { save_pts(s->s, path); return PT_OK; });
}

int pt_host_scene_set_camera(pt_host_scene* s, const pt_camera* cam) {
    if (!s || !cam) return PT_ERR_INVALID_ARG;
    s->s.camera = *cam;
    return PT_OK;
}

int pt_host_scene_get_camera(const pt_host_scene* s, pt_camera* cam) {
    if (!s || !cam) return PT_ERR_INVALID_ARG;
    *cam = s->s.camera;
    return PT_OK;
}

int pt_host_scene_set_background(pt_host_scene* s, const float rgb[3]) {
    if (!s || !rgb) return PT_ERR_INVALID_ARG;
    s->s.background = {rgb[0], rgb[1], rgb[2]};
    return PT_OK;
}

int pt_host_scene_add_material(pt_host_scene* s, const pt_material* m) {
    if (!s || !m || m->type < PT_MAT_DIFFUSE || m->type > PT_MAT_PHONG) { g_err = "bad material"; return -PT_ERR_INVALID_ARG; }
    s->s.materials.push_back(*m);
    s->s.finalized = false;
    return int(s->s.materials.size()) - 1;
}

int pt_host_scene_add_point_light(pt_host_scene* s, const float position[3], const float intensity[3]) {
    if (!s || !position || !intensity) return -PT_ERR_INVALID_ARG;
    ParsedLight l;
    l.type = PT_LIGHT_POINT;
    l.position = {position[0], position[1], position[2]};
    l.value = {intensity[0], intensity[1], intensity[2]};
    s->s.lights.push_back(l);
    s->s.finalized = false;
    return int(s->s.lights.size()) - 1;
}

static void attach_emitter(HostScene& hs, ParsedShape& sh, const float* radiance) {
    if (!radiance) return;
    sh.area_light_id = int(hs.lights.size());
    ParsedLight l;
    l.type = PT_LIGHT_DIFFUSE_AREA;
    l.value = {radiance[0], radiance[1], radiance[2]};
    l.shape_id = int(hs.shapes.size());
    hs.lights.push_back(l);
}

int pt_host_scene_add_sphere(pt_host_scene* s, const float center[3], float radius, int material_id,
                             const float* radiance) {
    if (!s || !center) return -PT_ERR_INVALID_ARG;
    ParsedShape sh;
    sh.type = PT_SHAPE_SPHERE;
    sh.material_id = material_id;
    sh.center = {center[0], center[1], center[2]};
    sh.radius = radius;
    attach_emitter(s->s, sh, radiance);
    s->s.shapes.push_back(std::move(sh));
    s->s.finalized = false;
    return int(s->s.shapes.size()) - 1;
}

int pt_host_scene_add_mesh(pt_host_scene* s, const float* positions, int num_vertices, const int32_t* indices,
                           int num_faces, const float* normals, int material_id, const float* radiance) {
    if (!s || !positions || !indices || num_vertices <= 0 || num_faces <= 0) { g_err = "bad mesh arguments"; return -PT_ERR_INVALID_ARG; }
    ParsedShape sh;
    sh.type = PT_SHAPE_TRIANGLE;
    sh.material_id = material_id;
    sh.positions.resize(num_vertices);
    std::memcpy(sh.positions.data(), positions, sizeof(f3) * size_t(num_vertices));
    sh.indices.resize(num_faces);
    std::memcpy(sh.indices.data(), indices, sizeof(i3) * size_t(num_faces));
    for (const i3& f : sh.indices)
        for (int v : {f.x, f.y, f.z})
            if (v < 0 || v >= num_vertices) { g_err = "face index out of range"; return -PT_ERR_BAD_SCENE; }
    if (normals) {
        sh.normals.resize(num_vertices);
        std::memcpy(sh.normals.data(), normals, sizeof(f3) * size_t(num_vertices));
    } else {
        sh.normals = compute_normals(sh.positions, sh.indices);
    }
    attach_emitter(s->s, sh, radiance);
    s->s.shapes.push_back(std::move(sh));
    s->s.finalized = false;
    return int(s->s.shapes.size()) - 1;
}

int pt_host_scene_finalize(pt_host_scene* s, int bvh_sort_mode) {
    if (!s) return PT_ERR_INVALID_ARG;
    if (bvh_sort_mode != PT_BVH_SORT_TOTAL && bvh_sort_mode != PT_BVH_SORT_REFERENCE) { g_err = "bad sort mode"; return PT_ERR_INVALID_ARG; }
    if (s->s.finalized && s->s.bvh_sort_mode == bvh_sort_mode) return PT_OK;
    return guard([&] { s->s.finalize(bvh_sort_mode); return PT_OK; });
}

int pt_host_scene_get_desc(const pt_host_scene* s, pt_scene_desc* out) {
    if (!s || !out) return PT_ERR_INVALID_ARG;
    if (!s->s.finalized) { g_err = "scene not finalized"; return PT_ERR_INVALID_ARG; }
    const HostScene& h = s->s;
    std::memset(out, 0, sizeof *out);
    out->num_shapes = int(h.flat_shapes.size());       out->shapes = h.flat_shapes.data();
    out->num_meshes = int(h.flat_meshes.size());       out->meshes = h.flat_meshes.data();
    out->num_materials = int(h.materials.size());      out->materials = h.materials.data();
    out->num_lights = int(h.flat_lights.size());       out->lights = h.flat_lights.data();
    out->num_nodes = int(h.nodes.size());              out->nodes = h.nodes.data();
    out->root = h.root;
    out->background[0] = h.background.x; out->background[1] = h.background.y; out->background[2] = h.background.z;
    return PT_OK;
}

int pt_host_scene_bvh_depth(const pt_host_scene* s) { return (s && s->s.finalized) ? s->s.depth : -1; }

// camera.cuh:28-43
void pt_host_camera_ray_data(const pt_camera* cam, int width, int height, float out12[12]) {
    float aspect_ratio = float(width) / float(height);
    float viewport_height = float(2.0 * tanf(radians(cam->vfov / 2)));
    float viewport_width = aspect_ratio * viewport_height;
    f3 lookfrom{cam->lookfrom[0], cam->lookfrom[1], cam->lookfrom[2]};
    f3 lookat{cam->lookat[0], cam->lookat[1], cam->lookat[2]};
    f3 up{cam->up[0], cam->up[1], cam->up[2]};
    f3 cam_dir = normalize(lookat - lookfrom);
    f3 right = normalize(cross(cam_dir, up));
    f3 new_up = cross(right, cam_dir);
    f3 origin = lookfrom;
    f3 horizontal = viewport_width * right;
    f3 vertical = viewport_height * new_up;
    f3 top_left = origin - horizontal / float(2) + vertical / float(2) + cam_dir;
    const f3 v[4] = {origin, top_left, horizontal, vertical};
    std::memcpy(out12, v, sizeof v);
}

void pt_host_default_params(const pt_camera* cam, int width, int height, int spp, pt_render_params* p) {
    std::memset(p, 0, sizeof *p);
    float d[12];
    pt_host_camera_ray_data(cam, width, height, d);
    std::memcpy(p->cam_origin, d, 12);
    std::memcpy(p->cam_top_left, d + 3, 12);
    std::memcpy(p->cam_horizontal, d + 6, 12);
    std::memcpy(p->cam_vertical, d + 9, 12);
    p->width = width;
    p->height = height;
    p->spp = spp;
    p->seed = 1984;          // main.cu:61
    p->max_depth = 50;       // radiance.cuh:12
    p->rr_depth = 5;         // radiance.cuh:68
}

}  // extern "C"
