// Mitsuba-0.6 XML subset -> HostScene.  Accepts what the reference's parser accepts
// (parse_scene.cpp:792-860) for the features the hot path supports: perspective
// sensor, diffuse/mirror/plastic/phong (optionally wrapped in twosided), rgb/srgb
// colours, obj/ply/sphere/rectangle shapes with toWorld transforms, area emitters,
// point emitters, background radiance, <default> + $name substitution.
// Not supported (PT_ERR_UNSUPPORTED): bitmap textures, blinn BSDFs (the reference
// parses then silently drops them, shifting material ids: SURVEY H5d), serialized meshes.
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>

#include "parsed_scene.h"
#include "xml_lite.h"

namespace pth {
namespace {

using DefaultMap = std::map<std::string, std::string>;

std::string lower(std::string s) {
    for (char& c : s) c = char(std::tolower((unsigned char)c));
    return s;
}

// "$name" -> default value (parse_scene.cpp:63-74)
const std::string& subst(const std::string& value, const DefaultMap& dm) {
    if (!value.empty() && value[0] == '$') {
        auto it = dm.find(value.substr(1));
        if (it == dm.end()) throw Error(PT_ERR_PARSE, "Reference default variable " + value + " not found.");
        return it->second;
    }
    return value;
}

float to_float(const std::string& raw, const DefaultMap& dm) {
    const std::string& v = subst(raw, dm);
    char* end = nullptr;
    float f = std::strtof(v.c_str(), &end);
    if (end == v.c_str()) throw Error(PT_ERR_PARSE, "expected a number, got '" + v + "'");
    return f;
}

int to_int(const std::string& raw, const DefaultMap& dm) {
    const std::string& v = subst(raw, dm);
    char* end = nullptr;
    long n = std::strtol(v.c_str(), &end, 10);
    if (end == v.c_str()) throw Error(PT_ERR_PARSE, "expected an integer, got '" + v + "'");
    return int(n);
}

bool to_bool(const std::string& raw, const DefaultMap& dm) {
    const std::string& v = subst(raw, dm);
    if (v == "true") return true;
    if (v == "false") return false;
    throw Error(PT_ERR_PARSE, "parse_boolean failed");
}

std::vector<float> to_floats(const std::string& raw, const DefaultMap& dm) {   // split on (,| )+
    const std::string& v = subst(raw, dm);
    std::vector<float> out;
    size_t i = 0;
    while (i < v.size()) {
        while (i < v.size() && (v[i] == ',' || v[i] == ' ')) i++;
        if (i >= v.size()) break;
        size_t b = i;
        while (i < v.size() && v[i] != ',' && v[i] != ' ') i++;
        std::string tok = v.substr(b, i - b);
        char* end = nullptr;
        float f = std::strtof(tok.c_str(), &end);
        if (end == tok.c_str()) throw Error(PT_ERR_PARSE, "expected a number, got '" + tok + "'");
        out.push_back(f);
    }
    return out;
}

f3 to_vec3(const std::string& raw, const DefaultMap& dm) {   // parse_scene.cpp:44-61
    std::vector<float> l = to_floats(raw, dm);
    if (l.size() == 1) return {l[0], l[0], l[0]};
    if (l.size() == 3) return {l[0], l[1], l[2]};
    throw Error(PT_ERR_PARSE, "parse_vector3 failed");
}

f3 srgb_to_rgb(f3 c) {   // parse_scene.cpp:29-36
    auto conv = [](float v) {
        return v <= float(0.04045) ? v / float(12.92) : powf((v + float(0.055)) / float(1.055), float(2.4));
    };
    return {conv(c.x), conv(c.y), conv(c.z)};
}

f3 to_srgb(const std::string& raw, const DefaultMap& dm) {   // parse_scene.cpp:139-156
    const std::string& v = subst(raw, dm);
    if (v.size() != 7 || v[0] != '#') throw Error(PT_ERR_PARSE, "Unknown SRGB format: " + v);
    char* end = nullptr;
    long enc = std::strtol(v.c_str() + 1, &end, 16);
    if (*end != '\0') throw Error(PT_ERR_PARSE, "Invalid SRGB value: " + v);
    return {((enc & 0xFF0000) >> 16) / 255.0f, ((enc & 0x00FF00) >> 8) / 255.0f, (enc & 0x0000FF) / 255.0f};
}

f3 parse_color(const XmlNode& n, const DefaultMap& dm) {   // parse_scene.cpp:430-452
    if (n.name == "rgb") return to_vec3(n.attr("value"), dm);
    if (n.name == "srgb") return srgb_to_rgb(to_srgb(n.attr("value"), dm));
    if (n.name == "ref" || n.name == "texture") throw Error(PT_ERR_UNSUPPORTED, "image textures are not supported");
    throw Error(PT_ERR_PARSE, "Unknown spectrum texture type:" + n.name);
}

f3 parse_intensity(const XmlNode& n, const DefaultMap& dm) {   // parse_scene.cpp:454-466
    if (n.name == "rgb") return to_vec3(n.attr("value"), dm);
    if (n.name == "srgb") return srgb_to_rgb(to_srgb(n.attr("value"), dm));
    return {1, 1, 1};
}

Mat4 parse_transform(const XmlNode& node, const DefaultMap& dm) {   // parse_scene.cpp:186-267
    Mat4 t = Mat4::identity();
    auto xyz = [&](const XmlNode& c, float dflt) {
        f3 v{dflt, dflt, dflt};
        if (c.has("x")) v.x = to_float(c.attr("x"), dm);
        if (c.has("y")) v.y = to_float(c.attr("y"), dm);
        if (c.has("z")) v.z = to_float(c.attr("z"), dm);
        return v;
    };
    for (auto& cp : node.children) {
        const XmlNode& c = *cp;
        std::string name = lower(c.name);
        if (name == "scale") {
            f3 v = xyz(c, 1.0f);
            if (c.has("value")) v = to_vec3(c.attr("value"), dm);
            t = scale(v) * t;
        } else if (name == "translate") {
            f3 v = xyz(c, 0.0f);
            if (c.has("value")) v = to_vec3(c.attr("value"), dm);
            t = translate(v) * t;
        } else if (name == "rotate") {
            f3 v = xyz(c, 0.0f);
            float angle = c.has("angle") ? to_float(c.attr("angle"), dm) : 0.0f;
            t = rotate(angle, v) * t;
        } else if (name == "lookat") {
            t = look_at(to_vec3(c.attr("origin"), dm), to_vec3(c.attr("target"), dm), to_vec3(c.attr("up"), dm)) * t;
        } else if (name == "matrix") {
            std::vector<float> l = to_floats(c.attr("value"), dm);
            if (l.size() != 16) throw Error(PT_ERR_PARSE, "parse_matrix4x4 failed");
            Mat4 m;
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m(i, j) = l[i * 4 + j];
            t = m * t;
        }
    }
    return t;
}

void parse_sensor(const XmlNode& node, const DefaultMap& dm, pt_camera& cam) {   // parse_scene.cpp:305-384
    f3 from{0, 0, 0}, at{0, 0, -1}, up{0, 1, 0};
    float fov = 45.0f;
    int width = 256, height = 256, spp = 16;
    enum { AX_X, AX_Y, AX_DIAG, AX_SMALLER, AX_LARGER } axis = AX_X;
    if (node.attr("type") != "perspective") throw Error(PT_ERR_UNSUPPORTED, "Unsupported sensor: " + node.attr("type"));
    for (auto& cp : node.children) {
        const XmlNode& c = *cp;
        const std::string& name = c.attr("name");
        if (name == "fov") fov = to_float(c.attr("value"), dm);
        else if (name == "toWorld" || name == "to_world") {
            for (auto& gp : c.children) {
                if (lower(gp->name) != "lookat") throw Error(PT_ERR_UNSUPPORTED, "Only support LookAt transform in a sensor.");
                from = to_vec3(gp->attr("origin"), dm);
                at = to_vec3(gp->attr("target"), dm);
                up = to_vec3(gp->attr("up"), dm);
            }
        } else if (name == "fovAxis" || name == "fov_axis") {
            const std::string& v = c.attr("value");
            if (v == "x") axis = AX_X; else if (v == "y") axis = AX_Y; else if (v == "diagonal") axis = AX_DIAG;
            else if (v == "smaller") axis = AX_SMALLER; else if (v == "larger") axis = AX_LARGER;
            else throw Error(PT_ERR_PARSE, "Unknown fovAxis value: " + v);
        }
    }
    for (auto& cp : node.children) {
        const XmlNode& c = *cp;
        if (c.name == "film") {
            for (auto& gp : c.children) {
                const std::string& n = gp->attr("name");
                if (n == "width") width = to_int(gp->attr("value"), dm);
                else if (n == "height") height = to_int(gp->attr("value"), dm);
            }
        } else if (c.name == "sampler") {
            for (auto& gp : c.children) {
                const std::string& n = gp->attr("name");
                if (n == "sampleCount" || n == "sample_count") spp = to_int(gp->attr("value"), dm);
            }
        }
    }
    // convert to vertical FOV (parse_scene.cpp:364-375)
    if (axis == AX_X || (axis == AX_SMALLER && width < height) || (axis == AX_LARGER && height < width)) {
        fov = degrees(2 * atanf(tanf(radians(fov) / 2) * height / float(width)));
    } else if (axis == AX_DIAG) {
        float aspect = float(height) / width;
        float diagonal = 2 * tanf(radians(fov) / 2);
        float h = diagonal / sqrtf(1 + 1 / (aspect * aspect));
        fov = degrees(2 * atanf(h / 2));
    }
    cam = pt_camera{{from.x, from.y, from.z}, {at.x, at.y, at.z}, {up.x, up.y, up.z}, fov, width, height, spp};
}

// returns (id, material); parse_scene.cpp:468-561
std::pair<std::string, pt_material> parse_bsdf(const XmlNode& node, const DefaultMap& dm, const std::string& parent_id = "") {
    const std::string& type = node.attr("type");
    std::string id = node.has("id") ? node.attr("id") : parent_id;
    if (type == "twosided") {
        for (auto& c : node.children)
            if (c->name == "bsdf") return parse_bsdf(*c, dm, id);
        throw Error(PT_ERR_PARSE, "twosided bsdf without a nested bsdf");
    }
    pt_material m{};
    f3 refl{0.5f, 0.5f, 0.5f};
    m.eta = 1.5f;
    m.exponent = 5.0f;
    if (type == "diffuse") m.type = PT_MAT_DIFFUSE;
    else if (type == "mirror") { m.type = PT_MAT_MIRROR; refl = {1, 1, 1}; }
    else if (type == "plastic") m.type = PT_MAT_PLASTIC;
    else if (type == "phong") m.type = PT_MAT_PHONG;
    else if (type == "blinn" || type == "blinnphong" || type == "blinn_microfacet" || type == "blinnphong_microfacet")
        throw Error(PT_ERR_UNSUPPORTED, "Blinn BSDFs are not supported by the render path (reference drops them: scene.cpp:96-112)");
    else throw Error(PT_ERR_PARSE, "Unknown BSDF: " + type);
    for (auto& c : node.children) {
        const std::string& name = c->attr("name");
        if (name == "reflectance") refl = parse_color(*c, dm);
        else if (m.type == PT_MAT_PLASTIC && (name == "ior" || name == "eta")) m.eta = to_float(c->attr("value"), dm);
        else if (m.type == PT_MAT_PHONG && (name == "exponent" || name == "alpha")) m.exponent = to_float(c->attr("value"), dm);
    }
    m.reflectance[0] = refl.x; m.reflectance[1] = refl.y; m.reflectance[2] = refl.z;
    return {id, m};
}

struct ShapeCommon {
    std::string filename;
    Mat4 to_world = Mat4::identity();
    bool face_normals = false;
};

ShapeCommon parse_shape_common(const XmlNode& node, const DefaultMap& dm) {
    ShapeCommon sc;
    for (auto& c : node.children) {
        const std::string& name = c->attr("name");
        if (name == "filename") sc.filename = subst(c->attr("value"), dm);
        else if ((name == "toWorld" || name == "to_world") && c->name == "transform") sc.to_world = parse_transform(*c, dm);
        else if (name == "faceNormals" || name == "face_normals") sc.face_normals = to_bool(c->attr("value"), dm);
    }
    return sc;
}

void parse_shape(const XmlNode& node, const DefaultMap& dm, const std::string& base_dir,
                 std::map<std::string, int>& material_map, HostScene& out) {   // parse_scene.cpp:591-790
    int material_id = -1;
    for (auto& c : node.children) {
        if (c->name == "ref") {
            if (!c->has("id")) throw Error(PT_ERR_PARSE, "Material reference id not specified.");
            auto it = material_map.find(c->attr("id"));
            if (it == material_map.end()) throw Error(PT_ERR_PARSE, "Material reference " + c->attr("id") + " not found.");
            material_id = it->second;
        } else if (c->name == "bsdf") {
            auto [name, m] = parse_bsdf(*c, dm);
            if (!name.empty()) material_map[name] = int(out.materials.size());
            material_id = int(out.materials.size());
            out.materials.push_back(m);
        }
    }
    ParsedShape shape;
    const std::string& type = node.attr("type");
    auto resolve = [&](const std::string& f) { return (!f.empty() && f[0] == '/') ? f : base_dir + "/" + f; };
    if (type == "obj" || type == "ply") {
        ShapeCommon sc = parse_shape_common(node, dm);
        shape.type = PT_SHAPE_TRIANGLE;
        if (type == "obj") load_obj(resolve(sc.filename), sc.to_world, shape);
        else load_ply(resolve(sc.filename), sc.to_world, shape);
        if (sc.face_normals)
            throw Error(PT_ERR_UNSUPPORTED, "faceNormals=true is not supported (render path requires vertex normals, SURVEY H5a)");
        if (shape.normals.empty()) shape.normals = compute_normals(shape.positions, shape.indices);
    } else if (type == "sphere") {
        shape.type = PT_SHAPE_SPHERE;
        for (auto& c : node.children) {
            const std::string& name = c->attr("name");
            if (name == "center")
                shape.center = {to_float(c->attr("x"), dm), to_float(c->attr("y"), dm), to_float(c->attr("z"), dm)};
            else if (name == "radius") shape.radius = to_float(c->attr("value"), dm);
        }
    } else if (type == "rectangle") {   // parse_scene.cpp:727-761
        shape.type = PT_SHAPE_TRIANGLE;
        Mat4 to_world = Mat4::identity();
        bool flip = false;
        shape.positions = {{-1, -1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 1, 0}};
        shape.indices = {{0, 1, 2}, {0, 2, 3}};
        shape.normals = {{0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}};
        for (auto& c : node.children) {
            const std::string& name = c->attr("name");
            if ((name == "toWorld" || name == "to_world") && c->name == "transform") to_world = parse_transform(*c, dm);
            else if (name == "flipNormals" || name == "flip_normals") flip = to_bool(c->attr("value"), dm);
        }
        if (flip) for (f3& n : shape.normals) n = -n;
        for (f3& p : shape.positions) p = xform_point(to_world, p);
        Mat4 inv = inverse(to_world);
        for (f3& n : shape.normals) n = xform_normal(inv, n);
    } else if (type == "serialized") {
        throw Error(PT_ERR_UNSUPPORTED, "serialized meshes are not supported");
    } else {
        throw Error(PT_ERR_PARSE, "Unknown shape:" + type);
    }
    shape.material_id = material_id;
    for (auto& c : node.children) {   // area emitter: parse_scene.cpp:768-784
        if (c->name != "emitter") continue;
        f3 radiance{1, 1, 1};
        for (auto& g : c->children)
            if (g->attr("name") == "radiance") radiance = parse_intensity(*g, dm);
        shape.area_light_id = int(out.lights.size());
        ParsedLight l;
        l.type = PT_LIGHT_DIFFUSE_AREA;
        l.value = radiance;
        l.shape_id = int(out.shapes.size());
        out.lights.push_back(l);
    }
    out.shapes.push_back(std::move(shape));
}

}  // namespace

void load_xml(const std::string& path, HostScene& out) {
    std::ifstream ifs(path, std::ios::binary);
    if (!ifs) throw Error(PT_ERR_IO, "cannot open scene file " + path);
    std::stringstream ss;
    ss << ifs.rdbuf();
    std::unique_ptr<XmlNode> doc = xml_parse(ss.str());
    const XmlNode* scene = doc->child("scene");
    if (!scene) throw Error(PT_ERR_PARSE, "no <scene> element in " + path);
    size_t slash = path.find_last_of('/');
    std::string base_dir = slash == std::string::npos ? "." : path.substr(0, slash);

    DefaultMap dm;
    std::map<std::string, int> material_map;
    for (auto& cp : scene->children) {   // parse_scene.cpp:812-853
        const XmlNode& c = *cp;
        if (c.name == "default") {
            if (c.has("name") && c.has("value")) dm[c.attr("name")] = c.attr("value");
        } else if (c.name == "sensor") {
            parse_sensor(c, dm, out.camera);
        } else if (c.name == "bsdf") {
            auto [name, m] = parse_bsdf(c, dm);
            if (!name.empty()) {
                material_map[name] = int(out.materials.size());
                out.materials.push_back(m);
            }
        } else if (c.name == "emitter") {   // parse_scene.cpp:563-589
            if (c.attr("type") != "point") throw Error(PT_ERR_PARSE, "Unknown emitter: " + c.attr("type"));
            ParsedLight l;
            l.type = PT_LIGHT_POINT;
            for (auto& g : c.children) {
                const std::string& name = g->attr("name");
                if (name == "position") {
                    if (g->has("x")) l.position.x = to_float(g->attr("x"), dm);
                    if (g->has("y")) l.position.y = to_float(g->attr("y"), dm);
                    if (g->has("z")) l.position.z = to_float(g->attr("z"), dm);
                } else if (name == "intensity") l.value = parse_intensity(*g, dm);
            }
            out.lights.push_back(l);
        } else if (c.name == "shape") {
            parse_shape(c, dm, base_dir, material_map, out);
        } else if (c.name == "texture") {
            throw Error(PT_ERR_UNSUPPORTED, "image textures are not supported");
        } else if (c.name == "background") {
            for (auto& g : c.children)
                if (g->attr("name") == "radiance") out.background = parse_intensity(*g, dm);
        }
    }
}

}  // namespace pth
