// Host scene model: the ParsedScene (parse_scene.h:114-121) equivalent plus the
// flattened Scene (scene.h:17-35) produced by finalize().
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/pt_host.h"
#include "vecmath.h"

namespace pth {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

struct ParsedLight {            // parse_scene.h:59-69
    int type;                   // PT_LIGHT_*
    f3 position{0, 0, 0};       // point
    f3 value{1, 1, 1};          // intensity (point) / radiance (area)
    int shape_id = -1;          // area
};

struct ParsedShape {            // parse_scene.h:75-95
    int type;                   // PT_SHAPE_SPHERE | PT_SHAPE_TRIANGLE (= a whole mesh here)
    int material_id = -1;
    int area_light_id = -1;
    f3 center{0, 0, 0};
    float radius = 1;
    std::vector<f3> positions;
    std::vector<i3> indices;
    std::vector<f3> normals;
};

struct HostScene {
    // ---- parsed ----
    pt_camera camera{{0, 0, 0}, {0, 0, -1}, {0, 1, 0}, 45.0f, 256, 256, 16};   // parse_scene.cpp:12-13,794-810
    f3 background{0.5f, 0.5f, 0.5f};
    std::vector<pt_material> materials;
    std::vector<ParsedLight> lights;
    std::vector<ParsedShape> shapes;

    // ---- flattened (finalize) ----
    bool finalized = false;
    int bvh_sort_mode = PT_BVH_SORT_TOTAL;
    std::vector<pt_shape> flat_shapes;
    std::vector<pt_mesh> flat_meshes;         // pointers into the three pools below
    std::vector<std::vector<float>> mesh_positions;
    std::vector<std::vector<int32_t>> mesh_indices;
    std::vector<std::vector<float>> mesh_normals;
    std::vector<pt_light> flat_lights;
    std::vector<pt_bvh_node> nodes;
    int root = -1;
    int depth = 0;

    void finalize(int sort_mode);             // scene_build.cpp
};

std::vector<f3> compute_normals(const std::vector<f3>& vertices, const std::vector<i3>& indices);

// mesh_io.cpp
void load_obj(const std::string& path, const Mat4& to_world, ParsedShape& out);
void load_ply(const std::string& path, const Mat4& to_world, ParsedShape& out);
// scene_xml.cpp
void load_xml(const std::string& path, HostScene& out);
// pts_io.cpp
void save_pts(const HostScene& s, const std::string& path);
void load_pts(const std::string& path, HostScene& out);

}  // namespace pth
