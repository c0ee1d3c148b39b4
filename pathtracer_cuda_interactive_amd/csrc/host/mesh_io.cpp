// OBJ / PLY mesh readers.  Behaviour follows parse_obj.cpp:111-203 and
// parse_ply.cpp:9-123 (positions transformed by to_world, normals by the inverse
// transpose, quads fanned as (0,1,2),(0,2,3), OBJ vertices de-duplicated on the
// (v,vt,vn) triple).  UVs are not kept: they cannot influence the result
// (texture.h:69-71).
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

#include "parsed_scene.h"

namespace pth {
namespace {

std::string trim(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) b++;
    while (e > b && std::isspace((unsigned char)s[e - 1])) e--;
    return s.substr(b, e - b);
}

// "v/vt/vn" -> (v, vt, vn), missing fields 0 (parse_obj.cpp:28-44)
std::tuple<int, int, int> split_face(const std::string& s) {
    int f[3] = {0, 0, 0};
    size_t pos = 0;
    for (int k = 0; k < 3 && pos <= s.size(); k++) {
        size_t slash = s.find('/', pos);
        std::string tok = s.substr(pos, slash == std::string::npos ? std::string::npos : slash - pos);
        if (!tok.empty()) f[k] = std::atoi(tok.c_str());
        if (slash == std::string::npos) break;
        pos = slash + 1;
    }
    return {f[0], f[1], f[2]};
}

}  // namespace

void load_obj(const std::string& path, const Mat4& to_world, ParsedShape& mesh) {
    std::ifstream ifs(path);
    if (!ifs.is_open()) throw Error(PT_ERR_IO, "Unable to open the obj file " + path);
    std::vector<f3> pos_pool, nor_pool;
    std::map<std::tuple<int, int, int>, int> vertex_map;
    const Mat4 inv = inverse(to_world);
    bool any_normal = false, any_missing_normal = false;

    auto vertex_id = [&](const std::tuple<int, int, int>& key) -> int {   // parse_obj.cpp:66-109
        auto it = vertex_map.find(key);
        if (it != vertex_map.end()) return it->second;
        int id = int(mesh.positions.size());
        int v = std::get<0>(key), vn = std::get<2>(key);
        int vi = v > 0 ? v - 1 : int(pos_pool.size()) + v;
        if (vi < 0 || vi >= int(pos_pool.size())) throw Error(PT_ERR_PARSE, "OBJ vertex index out of range in " + path);
        mesh.positions.push_back(xform_point(to_world, pos_pool[vi]));
        if (vn != 0) {
            int ni = vn > 0 ? vn - 1 : int(nor_pool.size()) + vn;
            if (ni < 0 || ni >= int(nor_pool.size())) throw Error(PT_ERR_PARSE, "OBJ normal index out of range in " + path);
            mesh.normals.push_back(xform_normal(inv, nor_pool[ni]));
            any_normal = true;
        } else {
            any_missing_normal = true;
        }
        vertex_map[key] = id;
        return id;
    };

    std::string raw;
    while (std::getline(ifs, raw)) {
        std::string line = trim(raw);
        if (line.empty() || line[0] == '#') continue;
        std::stringstream ss(line);
        std::string token;
        ss >> token;
        if (token == "v") {
            float x = 0, y = 0, z = 0, w = 1;
            ss >> x >> y >> z;
            float w_in;
            if (ss >> w_in) w = w_in;
            pos_pool.push_back(f3{x, y, z} / w);
        } else if (token == "vn") {
            float x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            nor_pool.push_back(normalize(f3{x, y, z}));
        } else if (token == "f") {
            std::string i0, i1, i2, i3s, i4;
            ss >> i0 >> i1 >> i2;
            if (i2.empty()) throw Error(PT_ERR_PARSE, "OBJ face with fewer than 3 vertices in " + path);
            int a = vertex_id(split_face(i0));
            int b = vertex_id(split_face(i1));
            int c = vertex_id(split_face(i2));
            mesh.indices.push_back({a, b, c});
            if (ss >> i3s) {
                int d = vertex_id(split_face(i3s));
                mesh.indices.push_back({a, c, d});
            }
            if (ss >> i4) throw Error(PT_ERR_UNSUPPORTED, "The object file contains n-gon (n>4) that we do not support.");
        }
    }
    if (any_normal && any_missing_normal)
        throw Error(PT_ERR_UNSUPPORTED, "OBJ mixes vertices with and without normals: " + path);
}

namespace {

struct PlyProp {
    std::string name;
    std::string type;        // scalar type, or list item type
    bool is_list = false;
    std::string count_type;  // list only
};
struct PlyElem {
    std::string name;
    size_t count = 0;
    std::vector<PlyProp> props;
};

int type_size(const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    throw Error(PT_ERR_PARSE, "PLY: unknown type " + t);
}

double read_scalar_bin(const unsigned char*& p, const unsigned char* end, const std::string& t) {
    int n = type_size(t);
    if (p + n > end) throw Error(PT_ERR_PARSE, "PLY: truncated data");
    double v = 0;
    if (t == "char" || t == "int8") { int8_t x; std::memcpy(&x, p, 1); v = x; }
    else if (t == "uchar" || t == "uint8") { uint8_t x; std::memcpy(&x, p, 1); v = x; }
    else if (t == "short" || t == "int16") { int16_t x; std::memcpy(&x, p, 2); v = x; }
    else if (t == "ushort" || t == "uint16") { uint16_t x; std::memcpy(&x, p, 2); v = x; }
    else if (t == "int" || t == "int32") { int32_t x; std::memcpy(&x, p, 4); v = x; }
    else if (t == "uint" || t == "uint32") { uint32_t x; std::memcpy(&x, p, 4); v = x; }
    else if (t == "float" || t == "float32") { float x; std::memcpy(&x, p, 4); v = x; }
    else { double x; std::memcpy(&x, p, 8); v = x; }
    p += n;
    return v;
}

}  // namespace

void load_ply(const std::string& path, const Mat4& to_world, ParsedShape& mesh) {
    std::ifstream ifs(path, std::ios::binary);
    if (!ifs) throw Error(PT_ERR_IO, "Unable to open the ply file " + path);
    std::string line;
    std::getline(ifs, line);
    if (trim(line) != "ply") throw Error(PT_ERR_PARSE, "not a PLY file: " + path);
    std::vector<PlyElem> elems;
    bool binary = false;
    while (std::getline(ifs, line)) {
        std::stringstream ss(trim(line));
        std::string tok;
        ss >> tok;
        if (tok == "format") {
            std::string fmt;
            ss >> fmt;
            if (fmt == "binary_little_endian") binary = true;
            else if (fmt == "ascii") binary = false;
            else throw Error(PT_ERR_UNSUPPORTED, "PLY format not supported: " + fmt);
        } else if (tok == "element") {
            PlyElem e;
            ss >> e.name >> e.count;
            elems.push_back(e);
        } else if (tok == "property") {
            if (elems.empty()) throw Error(PT_ERR_PARSE, "PLY: property before element");
            PlyProp p;
            std::string t;
            ss >> t;
            if (t == "list") { p.is_list = true; ss >> p.count_type >> p.type >> p.name; }
            else { p.type = t; ss >> p.name; }
            elems.back().props.push_back(p);
        } else if (tok == "end_header") {
            break;
        }
    }
    std::vector<unsigned char> blob;
    std::stringstream ascii;
    if (binary) blob.assign(std::istreambuf_iterator<char>(ifs), std::istreambuf_iterator<char>());
    else ascii << ifs.rdbuf();
    const unsigned char* p = blob.data();
    const unsigned char* end = blob.data() + blob.size();
    auto next = [&](const std::string& t) -> double {
        if (binary) return read_scalar_bin(p, end, t);
        double v;
        if (!(ascii >> v)) throw Error(PT_ERR_PARSE, "PLY: truncated ascii data");
        return v;
    };

    const Mat4 inv = inverse(to_world);
    bool have_normals = false;
    for (const PlyElem& e : elems) {
        if (e.name == "vertex") {
            int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1;
            for (int k = 0; k < int(e.props.size()); k++) {
                const std::string& n = e.props[k].name;
                if (e.props[k].is_list) throw Error(PT_ERR_UNSUPPORTED, "PLY: list property on vertex");
                if (n == "x") ix = k; else if (n == "y") iy = k; else if (n == "z") iz = k;
                else if (n == "nx") inx = k; else if (n == "ny") iny = k; else if (n == "nz") inz = k;
            }
            if (ix < 0 || iy < 0 || iz < 0) throw Error(PT_ERR_PARSE, "Vertex positions not found in " + path);
            have_normals = inx >= 0 && iny >= 0 && inz >= 0;
            mesh.positions.resize(e.count);
            if (have_normals) mesh.normals.resize(e.count);
            std::vector<double> row(e.props.size());
            for (size_t v = 0; v < e.count; v++) {
                for (size_t k = 0; k < e.props.size(); k++) row[k] = next(e.props[k].type);
                mesh.positions[v] = xform_point(to_world, f3{float(row[ix]), float(row[iy]), float(row[iz])});
                if (have_normals)
                    mesh.normals[v] = xform_normal(inv, f3{float(row[inx]), float(row[iny]), float(row[inz])});
            }
        } else if (e.name == "face") {
            mesh.indices.reserve(e.count);
            for (size_t f = 0; f < e.count; f++) {
                for (const PlyProp& pr : e.props) {
                    if (!pr.is_list) { next(pr.type); continue; }
                    int n = int(next(pr.count_type));
                    bool is_idx = pr.name == "vertex_indices" || pr.name == "vertex_index";
                    if (is_idx && n != 3) throw Error(PT_ERR_UNSUPPORTED, "PLY: only triangle faces are supported");
                    int idx[3] = {0, 0, 0};
                    for (int k = 0; k < n; k++) {
                        double v = next(pr.type);
                        if (is_idx && k < 3) idx[k] = int(v);
                    }
                    if (is_idx) mesh.indices.push_back({idx[0], idx[1], idx[2]});
                }
            }
        } else {
            for (size_t r = 0; r < e.count; r++)
                for (const PlyProp& pr : e.props) {
                    if (!pr.is_list) { next(pr.type); continue; }
                    int n = int(next(pr.count_type));
                    for (int k = 0; k < n; k++) next(pr.type);
                }
        }
    }
    if (mesh.indices.empty()) throw Error(PT_ERR_PARSE, "Vertex indices not found in " + path);
}

}  // namespace pth
