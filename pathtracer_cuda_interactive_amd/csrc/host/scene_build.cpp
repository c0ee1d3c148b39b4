// Scene flattening + BVH build: the host producer of the hot path's inputs.
// Follows Scene::Scene (scene.cpp:11-153) and construct_bvh (bvh.cu:16-54).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <future>
#include <limits>

#include "parsed_scene.h"

namespace pth {

// Angle between two unit vectors as compute_normals.cpp:4-10 evaluates it, bit for bit: the chord form 2 asin(|b - a| / 2)
// for acute angles and — the reference's own slip, kept on purpose — (pi - 2) asin(|b + a| / 2) where
// pi - 2 asin(...) was meant.  Shading normals of OBJ meshes without `vn` lines depend on it.
static inline float corner_weight(f3 a, f3 b) {
    const bool obtuse = dot(a, b) < 0;
    const float half_chord = float(0.5) * length(obtuse ? b + a : b - a);
    return obtuse ? (kPi - 2) * asinf(half_chord) : 2 * asinf(half_chord);
}

// Vertex normals weighted by the corner angles of the faces around each vertex (compute_normals.cpp:13-51), same operation
// order as the reference: faces in index order, corners 0, 1, 2 of a face, unit face normal from corner 0's edges.
std::vector<f3> compute_normals(const std::vector<f3>& vertices, const std::vector<i3>& indices) {
    std::vector<f3> sum(vertices.size(), f3{0, 0, 0});
    for (const i3& tri : indices) {
        const int at[3] = {tri.x, tri.y, tri.z};
        const f3 p[3] = {vertices[at[0]], vertices[at[1]], vertices[at[2]]};
        f3 face = cross(p[1] - p[0], p[2] - p[0]);
        const float twice_area = length(face);
        if (twice_area == 0) continue;                   // degenerate face: contributes nothing (a NaN area falls through, as there)
        face = face / twice_area;
        for (int c = 0; c < 3; c++) {
            const f3 to_next = normalize(p[(c + 1) % 3] - p[c]), to_prev = normalize(p[(c + 2) % 3] - p[c]);
            sum[at[c]] = sum[at[c]] + face * corner_weight(to_next, to_prev);
        }
    }
    for (f3& n : sum) {
        const float len = length(n);
        if (len != 0) n = n / len;
        else n = f3{0, 0, 0};
    }
    return sum;
}

namespace {

struct BoxId {
    f3 lo, hi;
    int id;
};

struct BvhBuilder {
    std::vector<BoxId>& boxes;
    std::vector<pt_bvh_node>& pool;     // pre-sized to 2N-1; a subtree with n leaves owns 2n-1 consecutive slots (post-order)
    int sort_mode;
    std::atomic<int> max_depth{0};
    std::atomic<int> tasks_in_flight{0};

    static float center(const BoxId& b, int axis) {   // bvh.cu:5-6: (p_max + p_min) / 2.0f
        switch (axis) {
            case 0: return (b.hi.x + b.lo.x) * (1.0f / 2.0f);
            case 1: return (b.hi.y + b.lo.y) * (1.0f / 2.0f);
            default: return (b.hi.z + b.lo.z) * (1.0f / 2.0f);
        }
    }

    void note_depth(int depth) {
        int cur = max_depth.load(std::memory_order_relaxed);
        while (depth > cur && !max_depth.compare_exchange_weak(cur, depth, std::memory_order_relaxed)) {}
    }

    // bvh.cu:16-54 on the sub-range [lo,hi) in place (the reference copies the range and sorts the copy: same input
    // sequence, same result).  The node pool is post-order exactly like the reference's push_back order: the subtree
    // of [lo,hi) fills pool[base, base + 2(hi-lo)-1) with its root last, so independent subtrees can be built by
    // different threads and still land at the reference's indices.  Returns the index of the subtree's root.
    int build(size_t lo, size_t hi, int depth, size_t base) {
        note_depth(depth);
        const size_t n = hi - lo;
        const size_t self = base + 2 * n - 2;
        if (n == 1) {
            pt_bvh_node node;
            node.left = node.right = -1;
            node.prim = boxes[lo].id;
            std::memcpy(node.bmin, &boxes[lo].lo, 12);
            std::memcpy(node.bmax, &boxes[lo].hi, 12);
            pool[self] = node;
            return int(self);
        }
        const float inf = std::numeric_limits<float>::infinity();
        f3 bmin{inf, inf, inf}, bmax{-inf, -inf, -inf};           // bbox.cuh:19-26
        for (size_t k = lo; k < hi; k++) {                          // merge: bbox.cuh:104-114
            bmin = {tmin(bmin.x, boxes[k].lo.x), tmin(bmin.y, boxes[k].lo.y), tmin(bmin.z, boxes[k].lo.z)};
            bmax = {tmax(bmax.x, boxes[k].hi.x), tmax(bmax.y, boxes[k].hi.y), tmax(bmax.z, boxes[k].hi.z)};
        }
        f3 ext = bmax - bmin;                                       // largest_axis: bbox.cuh:93-102
        int axis = (ext.x > ext.y && ext.x > ext.z) ? 0 : ((ext.y > ext.x && ext.y > ext.z) ? 1 : 2);
        if (sort_mode == PT_BVH_SORT_REFERENCE) {
            std::sort(boxes.begin() + lo, boxes.begin() + hi,
                      [axis](const BoxId& a, const BoxId& b) { return center(a, axis) < center(b, axis); });
        } else {
            std::sort(boxes.begin() + lo, boxes.begin() + hi, [axis](const BoxId& a, const BoxId& b) {
                float ca = center(a, axis), cb = center(b, axis);
                if (ca < cb) return true;
                if (cb < ca) return false;
                return a.id < b.id;
            });
        }
        const size_t mid = lo + n / 2;
        const size_t left_base = base, right_base = base + 2 * (mid - lo) - 1;
        int left_id, right_id;
        // big subtrees: build the left half on another thread (bounded number of tasks)
        if (n > 32768 && tasks_in_flight.load(std::memory_order_relaxed) < 16) {
            tasks_in_flight.fetch_add(1, std::memory_order_relaxed);
            auto fut = std::async(std::launch::async, [&, lo, mid, depth, left_base] { return build(lo, mid, depth + 1, left_base); });
            right_id = build(mid, hi, depth + 1, right_base);
            left_id = fut.get();
            tasks_in_flight.fetch_sub(1, std::memory_order_relaxed);
        } else {
            left_id = build(lo, mid, depth + 1, left_base);
            right_id = build(mid, hi, depth + 1, right_base);
        }
        pt_bvh_node node;
        std::memcpy(node.bmin, &bmin, 12);
        std::memcpy(node.bmax, &bmax, 12);
        node.left = left_id;
        node.right = right_id;
        node.prim = -1;
        pool[self] = node;
        return int(self);
    }
};

}  // namespace

void HostScene::finalize(int sort_mode) {
    flat_shapes.clear(); flat_meshes.clear(); flat_lights.clear(); nodes.clear();
    mesh_positions.clear(); mesh_indices.clear(); mesh_normals.clear();
    bvh_sort_mode = sort_mode;

    auto area_radiance = [&](int light_id) -> f3 {
        if (light_id < 0 || light_id >= int(lights.size()) || lights[light_id].type != PT_LIGHT_DIFFUSE_AREA)
            throw Error(PT_ERR_BAD_SCENE, "shape refers to a light that is not an area light");
        return lights[light_id].value;
    };
    auto push_area_light = [&](int shape_index, f3 rad) {
        pt_light l{};
        l.type = PT_LIGHT_DIFFUSE_AREA;
        l.shape_id = shape_index;
        l.radiance[0] = rad.x; l.radiance[1] = rad.y; l.radiance[2] = rad.z;
        flat_lights.push_back(l);
    };

    // scene.cpp:27-91: shapes in order; one DiffuseAreaLight entry per emissive primitive
    for (const ParsedShape& ps : shapes) {
        if (ps.material_id < 0 || ps.material_id >= int(materials.size()))
            throw Error(PT_ERR_BAD_SCENE, "shape without a valid material");
        if (ps.type == PT_SHAPE_SPHERE) {
            if (ps.area_light_id >= 0) push_area_light(int(flat_shapes.size()), area_radiance(ps.area_light_id));
            pt_shape s{};
            s.type = PT_SHAPE_SPHERE;
            s.material_id = ps.material_id;
            s.area_light_id = ps.area_light_id;
            s.center[0] = ps.center.x; s.center[1] = ps.center.y; s.center[2] = ps.center.z;
            s.radius = ps.radius;
            s.face_index = -1; s.mesh_index = -1;
            flat_shapes.push_back(s);
        } else {
            if (ps.normals.size() != ps.positions.size())
                throw Error(PT_ERR_BAD_SCENE, "mesh needs one normal per vertex (SURVEY H5a)");
            int mesh_index = int(mesh_positions.size());
            mesh_positions.emplace_back(reinterpret_cast<const float*>(ps.positions.data()),
                                        reinterpret_cast<const float*>(ps.positions.data()) + 3 * ps.positions.size());
            mesh_indices.emplace_back(reinterpret_cast<const int32_t*>(ps.indices.data()),
                                      reinterpret_cast<const int32_t*>(ps.indices.data()) + 3 * ps.indices.size());
            mesh_normals.emplace_back(reinterpret_cast<const float*>(ps.normals.data()),
                                      reinterpret_cast<const float*>(ps.normals.data()) + 3 * ps.normals.size());
            for (const i3& f : ps.indices)
                for (int v : {f.x, f.y, f.z})
                    if (v < 0 || v >= int(ps.positions.size())) throw Error(PT_ERR_BAD_SCENE, "face index out of range");
            pt_mesh m{};
            m.material_id = ps.material_id;
            m.area_light_id = ps.area_light_id;
            m.num_vertices = int(ps.positions.size());
            m.num_faces = int(ps.indices.size());
            flat_meshes.push_back(m);
            f3 rad{0, 0, 0};
            if (ps.area_light_id >= 0) rad = area_radiance(ps.area_light_id);
            for (int face = 0; face < int(ps.indices.size()); face++) {
                if (ps.area_light_id >= 0) push_area_light(int(flat_shapes.size()), rad);
                pt_shape s{};
                s.type = PT_SHAPE_TRIANGLE;
                s.material_id = -1; s.area_light_id = -1;
                s.face_index = face; s.mesh_index = mesh_index;
                flat_shapes.push_back(s);
            }
        }
    }
    for (size_t k = 0; k < flat_meshes.size(); k++) {   // pools no longer move: take the addresses
        flat_meshes[k].positions = mesh_positions[k].data();
        flat_meshes[k].indices = mesh_indices[k].data();
        flat_meshes[k].normals = mesh_normals[k].data();
    }
    // scene.cpp:114-120: point lights go after the per-primitive area-light entries
    for (const ParsedLight& pl : lights) {
        if (pl.type != PT_LIGHT_POINT) continue;
        pt_light l{};
        l.type = PT_LIGHT_POINT;
        l.shape_id = -1;
        l.radiance[0] = pl.value.x; l.radiance[1] = pl.value.y; l.radiance[2] = pl.value.z;
        l.position[0] = pl.position.x; l.position[1] = pl.position.y; l.position[2] = pl.position.z;
        flat_lights.push_back(l);
    }
    if (flat_shapes.empty()) throw Error(PT_ERR_BAD_SCENE, "scene has no shapes");

    // scene.cpp:124-145: one AABB per primitive
    std::vector<BoxId> boxes(flat_shapes.size());
    for (int i = 0; i < int(boxes.size()); i++) {
        const pt_shape& s = flat_shapes[i];
        if (s.type == PT_SHAPE_SPHERE) {
            f3 c{s.center[0], s.center[1], s.center[2]};
            boxes[i] = {f3{c.x - s.radius, c.y - s.radius, c.z - s.radius},
                        f3{c.x + s.radius, c.y + s.radius, c.z + s.radius}, i};
        } else {
            const float* P = mesh_positions[s.mesh_index].data();
            const int32_t* I = &mesh_indices[s.mesh_index][3 * size_t(s.face_index)];
            f3 p0{P[3 * I[0]], P[3 * I[0] + 1], P[3 * I[0] + 2]};
            f3 p1{P[3 * I[1]], P[3 * I[1] + 1], P[3 * I[1] + 2]};
            f3 p2{P[3 * I[2]], P[3 * I[2] + 1], P[3 * I[2] + 2]};
            // float3_min/max: scene.h:466-472 (fminf/fmaxf per component)
            f3 lo{tmin(tmin(p0.x, p1.x), p2.x), tmin(tmin(p0.y, p1.y), p2.y), tmin(tmin(p0.z, p1.z), p2.z)};
            f3 hi{tmax(tmax(p0.x, p1.x), p2.x), tmax(tmax(p0.y, p1.y), p2.y), tmax(tmax(p0.z, p1.z), p2.z)};
            boxes[i] = {lo, hi, i};
        }
    }
    nodes.assign(2 * boxes.size() - 1, pt_bvh_node{});
    BvhBuilder b{boxes, nodes, sort_mode};
    root = b.build(0, boxes.size(), 1, 0);
    depth = b.max_depth.load();        // computeMaxDepth (bvh.cu:56-65): leaves count 1
    finalized = true;
}

}  // namespace pth
