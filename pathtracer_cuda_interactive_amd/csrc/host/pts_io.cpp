// .pts — binary container for a parsed scene (camera, background, materials,
// lights, shapes with post-transform mesh arrays).  Little-endian, versioned.
// It is the state right before Scene::Scene (scene.cpp:11): loading a .pts and
// calling finalize() reproduces exactly what loading the XML/OBJ/PLY would.
//
//   magic "PTSCENE1" | u32 version | pt_camera (52 B) | f32 background[3]
//   u32 n_materials | n x { i32 type, f32 rgb[3], f32 eta, f32 exponent }
//   u32 n_lights    | n x { i32 type, f32 position[3], f32 value[3], i32 shape_id }
//   u32 n_shapes    | n x { i32 type, i32 material_id, i32 area_light_id,
//                           sphere: f32 center[3], f32 radius
//                           mesh  : u32 nv, u32 nf, f32 pos[nv*3], i32 idx[nf*3], f32 nrm[nv*3] }
#include <cstdio>
#include <cstring>

#include "parsed_scene.h"

namespace pth {
namespace {

const char kMagic[8] = {'P', 'T', 'S', 'C', 'E', 'N', 'E', '1'};
const uint32_t kVersion = 1;

struct Writer {
    FILE* f;
    const std::string& path;
    void raw(const void* p, size_t n) {
        if (n && std::fwrite(p, 1, n, f) != n) throw Error(PT_ERR_IO, "write failed: " + path);
    }
    template <class T> void put(const T& v) { raw(&v, sizeof(T)); }
};
struct Reader {
    FILE* f;
    const std::string& path;
    void raw(void* p, size_t n) {
        if (n && std::fread(p, 1, n, f) != n) throw Error(PT_ERR_IO, "truncated scene file: " + path);
    }
    template <class T> T get() { T v; raw(&v, sizeof(T)); return v; }
};
struct FileCloser {
    FILE* f;
    ~FileCloser() { if (f) std::fclose(f); }
};

}  // namespace

void save_pts(const HostScene& s, const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error(PT_ERR_IO, "cannot create " + path);
    FileCloser fc{f};
    Writer w{f, path};
    w.raw(kMagic, 8);
    w.put(kVersion);
    w.put(s.camera);
    w.put(s.background);
    w.put(uint32_t(s.materials.size()));
    for (const pt_material& m : s.materials) w.put(m);
    w.put(uint32_t(s.lights.size()));
    for (const ParsedLight& l : s.lights) { w.put(int32_t(l.type)); w.put(l.position); w.put(l.value); w.put(int32_t(l.shape_id)); }
    w.put(uint32_t(s.shapes.size()));
    for (const ParsedShape& sh : s.shapes) {
        w.put(int32_t(sh.type)); w.put(int32_t(sh.material_id)); w.put(int32_t(sh.area_light_id));
        if (sh.type == PT_SHAPE_SPHERE) { w.put(sh.center); w.put(sh.radius); }
        else {
            w.put(uint32_t(sh.positions.size())); w.put(uint32_t(sh.indices.size()));
            w.raw(sh.positions.data(), sh.positions.size() * sizeof(f3));
            w.raw(sh.indices.data(), sh.indices.size() * sizeof(i3));
            w.raw(sh.normals.data(), sh.normals.size() * sizeof(f3));
        }
    }
}

void load_pts(const std::string& path, HostScene& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw Error(PT_ERR_IO, "cannot open " + path);
    FileCloser fc{f};
    Reader r{f, path};
    char magic[8];
    r.raw(magic, 8);
    if (std::memcmp(magic, kMagic, 8) != 0) throw Error(PT_ERR_PARSE, "not a .pts scene: " + path);
    if (r.get<uint32_t>() != kVersion) throw Error(PT_ERR_UNSUPPORTED, "unsupported .pts version: " + path);
    out.camera = r.get<pt_camera>();
    out.background = r.get<f3>();
    const uint32_t kMax = 1u << 28;
    uint32_t nm = r.get<uint32_t>();
    if (nm > kMax) throw Error(PT_ERR_PARSE, "corrupt .pts (materials)");
    out.materials.resize(nm);
    for (pt_material& m : out.materials) m = r.get<pt_material>();
    uint32_t nl = r.get<uint32_t>();
    if (nl > kMax) throw Error(PT_ERR_PARSE, "corrupt .pts (lights)");
    out.lights.resize(nl);
    for (ParsedLight& l : out.lights) { l.type = r.get<int32_t>(); l.position = r.get<f3>(); l.value = r.get<f3>(); l.shape_id = r.get<int32_t>(); }
    uint32_t ns = r.get<uint32_t>();
    if (ns > kMax) throw Error(PT_ERR_PARSE, "corrupt .pts (shapes)");
    out.shapes.resize(ns);
    for (ParsedShape& sh : out.shapes) {
        sh.type = r.get<int32_t>(); sh.material_id = r.get<int32_t>(); sh.area_light_id = r.get<int32_t>();
        if (sh.type == PT_SHAPE_SPHERE) { sh.center = r.get<f3>(); sh.radius = r.get<float>(); }
        else if (sh.type == PT_SHAPE_TRIANGLE) {
            uint32_t nv = r.get<uint32_t>(), nf = r.get<uint32_t>();
            if (nv > kMax || nf > kMax) throw Error(PT_ERR_PARSE, "corrupt .pts (mesh sizes)");
            sh.positions.resize(nv); sh.indices.resize(nf); sh.normals.resize(nv);
            r.raw(sh.positions.data(), nv * sizeof(f3));
            r.raw(sh.indices.data(), nf * sizeof(i3));
            r.raw(sh.normals.data(), nv * sizeof(f3));
        } else throw Error(PT_ERR_PARSE, "corrupt .pts (shape type)");
    }
}

}  // namespace pth
