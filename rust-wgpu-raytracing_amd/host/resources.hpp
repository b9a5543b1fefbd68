// Host-side mirror of the reference's asset loader (pure CPU).
//   load_string / load_binary / load_texture   /root/reference/src/resources.rs:20-66
//   load_model_compute                         /root/reference/src/resources.rs:163-264
// The reference resolves files under the compile-time OUT_DIR/res (:29-31,49-51);
// here the `res` directory is a run-time argument.  tobj 3.2.5 (not vendored) is
// replaced by the OBJ/MTL reader below, which reproduces the behaviour the
// reference relies on: `triangulate` (fan) + `single_index` (one vertex per
// unique v/vt/vn triple, first-use order; faces in file order), one model per
// o/g/usemtl group, Ka/Kd/Ks/Ns/map_Kd/map_Bump from the MTL.
#pragma once

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "model.hpp"

namespace rwr {
namespace resources {

struct Error {
    int code = RWR_OK;  // RWR_ERR_IO / RWR_ERR_PARSE
    std::string message;
    explicit operator bool() const { return code != RWR_OK; }
};

inline std::string join_path(const std::string &dir, const std::string &file)
{
    if (dir.empty()) return file;
    return dir.back() == '/' ? dir + file : dir + "/" + file;
}

// resources.rs:39-57
inline Error load_binary(const std::string &res_dir, const std::string &file_name, std::vector<uint8_t> &out)
{
    const std::string path = join_path(res_dir, file_name);
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return {RWR_ERR_IO, "cannot open " + path + ": " + std::strerror(errno)};
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz < 0) { std::fclose(f); return {RWR_ERR_IO, "cannot size " + path}; }
    out.resize((size_t)sz);
    const size_t got = sz ? std::fread(out.data(), 1, (size_t)sz, f) : 0;
    std::fclose(f);
    if (got != (size_t)sz) return {RWR_ERR_IO, "short read on " + path};
    return {};
}

// resources.rs:20-37
inline Error load_string(const std::string &res_dir, const std::string &file_name, std::string &out)
{
    std::vector<uint8_t> bytes;
    Error e = load_binary(res_dir, file_name, bytes);
    if (e) return e;
    out.assign(bytes.begin(), bytes.end());
    return {};
}

// resources.rs:59-66
inline Error load_texture(const std::string &res_dir, const std::string &file_name, texture::Texture &out)
{
    std::vector<uint8_t> data;
    Error e = load_binary(res_dir, file_name, data);
    if (e) return e;
    std::string err;
    if (!texture::Texture::from_bytes(data.data(), data.size(), out, err)) return {RWR_ERR_PARSE, file_name + ": " + err};
    return {};
}

namespace detail {

inline std::vector<std::string> split_ws(const std::string &s)
{
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r')) i++;
        size_t j = i;
        while (j < s.size() && s[j] != ' ' && s[j] != '\t' && s[j] != '\r') j++;
        if (j > i) out.emplace_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

inline bool parse_float(const std::string &tok, float &out)
{
    char *end = nullptr;
    errno = 0;
    out = std::strtof(tok.c_str(), &end);
    return end != tok.c_str() && *end == '\0';
}

inline bool parse_int(const std::string &tok, long &out)
{
    char *end = nullptr;
    out = std::strtol(tok.c_str(), &end, 10);
    return end != tok.c_str() && *end == '\0';
}

struct RawMesh {
    std::string name;
    std::vector<float> positions, texcoords, normals;
    std::vector<uint32_t> indices;
    std::string material_name;
    bool has_material = false;
};

struct RawMaterial {
    std::string name;
    float ambient[3] = {0, 0, 0}, diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0};
    float shininess = 0.0f;
    std::string diffuse_texture, normal_texture;
};

inline Error parse_mtl(const std::string &text, std::vector<RawMaterial> &out)
{
    std::istringstream in(text);
    std::string line;
    RawMaterial *cur = nullptr;
    int lineno = 0;
    while (std::getline(in, line)) {
        lineno++;
        const auto tok = split_ws(line);
        if (tok.empty() || tok[0][0] == '#') continue;
        const std::string &key = tok[0];
        auto rest = [&]() {
            std::string r;
            for (size_t i = 1; i < tok.size(); i++) r += (i > 1 ? " " : "") + tok[i];
            return r;
        };
        auto rgb = [&](float (&dst)[3]) -> bool {
            if (tok.size() < 4) return false;
            return parse_float(tok[1], dst[0]) && parse_float(tok[2], dst[1]) && parse_float(tok[3], dst[2]);
        };
        if (key == "newmtl") {
            out.emplace_back();
            cur = &out.back();
            cur->name = rest();
            continue;
        }
        if (!cur) continue;
        bool ok = true;
        if (key == "Ka") ok = rgb(cur->ambient);
        else if (key == "Kd") ok = rgb(cur->diffuse);
        else if (key == "Ks") ok = rgb(cur->specular);
        else if (key == "Ns") ok = tok.size() >= 2 && parse_float(tok[1], cur->shininess);
        else if (key == "map_Kd") cur->diffuse_texture = tok.size() >= 2 ? tok.back() : "";
        else if (key == "map_Bump" || key == "map_bump" || key == "bump") cur->normal_texture = tok.size() >= 2 ? tok.back() : "";
        if (!ok) return {RWR_ERR_PARSE, "MTL line " + std::to_string(lineno) + ": bad value for " + key};
    }
    return {};
}

inline Error parse_obj(const std::string &text, std::vector<RawMesh> &meshes, std::vector<std::string> &mtllibs)
{
    std::vector<float> pos, tex, nor;
    using Key = std::tuple<long, long, long>;
    std::vector<std::vector<Key>> cur_faces;
    std::string cur_name = "unnamed_object", cur_mtl;
    bool has_mtl = false;

    auto flush = [&]() {
        if (cur_faces.empty()) return;
        RawMesh mesh;
        mesh.name = cur_name;
        mesh.material_name = cur_mtl;
        mesh.has_material = has_mtl;
        std::map<Key, uint32_t> index_map;
        for (const auto &face : cur_faces) {
            for (size_t i = 1; i + 1 < face.size(); i++) {  // fan triangulation
                const Key tri[3] = {face[0], face[i], face[i + 1]};
                for (const Key &vert : tri) {
                    auto it = index_map.find(vert);
                    if (it != index_map.end()) {
                        mesh.indices.push_back(it->second);
                        continue;
                    }
                    const long v = std::get<0>(vert), vt = std::get<1>(vert), vn = std::get<2>(vert);
                    mesh.positions.insert(mesh.positions.end(), {pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]});
                    if (!tex.empty() && vt >= 0) mesh.texcoords.insert(mesh.texcoords.end(), {tex[2 * vt], tex[2 * vt + 1]});
                    if (!nor.empty() && vn >= 0) mesh.normals.insert(mesh.normals.end(), {nor[3 * vn], nor[3 * vn + 1], nor[3 * vn + 2]});
                    const uint32_t next = (uint32_t)index_map.size();
                    mesh.indices.push_back(next);
                    index_map.emplace(vert, next);
                }
            }
        }
        meshes.push_back(std::move(mesh));
        cur_faces.clear();
    };

    std::istringstream in(text);
    std::string line;
    int lineno = 0;
    while (std::getline(in, line)) {
        lineno++;
        const auto tok = split_ws(line);
        if (tok.empty() || tok[0][0] == '#') continue;
        const std::string &key = tok[0];
        auto bad = [&](const char *what) { return Error{RWR_ERR_PARSE, "OBJ line " + std::to_string(lineno) + ": " + what}; };
        auto rest = [&]() {
            std::string r;
            for (size_t i = 1; i < tok.size(); i++) r += (i > 1 ? " " : "") + tok[i];
            return r;
        };
        if (key == "v" || key == "vn") {
            float f[3];
            if (tok.size() < 4 || !parse_float(tok[1], f[0]) || !parse_float(tok[2], f[1]) || !parse_float(tok[3], f[2])) return bad("bad vertex");
            auto &dst = key == "v" ? pos : nor;
            dst.insert(dst.end(), {f[0], f[1], f[2]});
        } else if (key == "vt") {
            float u, v = 0.0f;
            if (tok.size() < 2 || !parse_float(tok[1], u) || (tok.size() > 2 && !parse_float(tok[2], v))) return bad("bad texcoord");
            tex.insert(tex.end(), {u, v});
        } else if (key == "f") {
            std::vector<Key> face;
            for (size_t i = 1; i < tok.size(); i++) {
                long idx[3] = {0, 0, 0};
                bool have[3] = {false, false, false};
                size_t start = 0;
                for (int k = 0; k < 3 && start <= tok[i].size(); k++) {
                    const size_t slash = tok[i].find('/', start);
                    const std::string part = tok[i].substr(start, slash == std::string::npos ? std::string::npos : slash - start);
                    if (!part.empty()) {
                        if (!parse_int(part, idx[k])) return bad("bad face index");
                        have[k] = true;
                    }
                    if (slash == std::string::npos) break;
                    start = slash + 1;
                }
                if (!have[0]) return bad("face corner without a position index");
                const long counts[3] = {(long)pos.size() / 3, (long)tex.size() / 2, (long)nor.size() / 3};
                long fixed[3] = {-1, -1, -1};
                for (int k = 0; k < 3; k++) {
                    if (!have[k]) continue;
                    fixed[k] = idx[k] > 0 ? idx[k] - 1 : counts[k] + idx[k];
                    if (fixed[k] < 0 || fixed[k] >= counts[k]) return bad("face index out of range");
                }
                face.emplace_back(fixed[0], fixed[1], fixed[2]);
            }
            if (face.size() >= 3) cur_faces.push_back(std::move(face));
        } else if (key == "o" || key == "g") {
            flush();
            cur_name = tok.size() > 1 ? rest() : "unnamed_object";
        } else if (key == "usemtl") {
            const std::string name = rest();
            if (!cur_faces.empty() && (!has_mtl || name != cur_mtl)) flush();
            cur_mtl = name;
            has_mtl = true;
        } else if (key == "mtllib") {
            mtllibs.push_back(rest());
        }
    }
    flush();
    return {};
}

}  // namespace detail

// resources.rs:163-264 — OBJ + MTL + diffuse textures -> Model with
// ModelVertexSmall / ModelFaceSmall buffers per mesh.
inline Error load_model_compute(const std::string &res_dir, const std::string &file_name, model::Model &out)
{
    std::string obj_text;
    Error e = load_string(res_dir, file_name, obj_text);
    if (e) return e;
    std::vector<detail::RawMesh> raw_meshes;
    std::vector<std::string> mtllibs;
    e = detail::parse_obj(obj_text, raw_meshes, mtllibs);
    if (e) return e;

    std::vector<detail::RawMaterial> raw_materials;
    for (const auto &lib : mtllibs) {
        std::string mat_text;
        e = load_string(res_dir, lib, mat_text);  // the reference unwrap()s this load (resources.rs:181)
        if (e) return e;
        e = detail::parse_mtl(mat_text, raw_materials);
        if (e) return e;
    }

    out = model::Model{};
    for (const auto &m : raw_materials) {
        model::Material mat;
        mat.name = m.name;
        mat.diffuse_texture_file = m.diffuse_texture;
        mat.normal_texture_file = m.normal_texture;
        if (m.diffuse_texture.empty()) return {RWR_ERR_IO, "material '" + m.name + "' has no map_Kd texture"};
        e = load_texture(res_dir, m.diffuse_texture, mat.diffuse_texture);  // resources.rs:189
        if (e) return e;
        if (!m.normal_texture.empty()) {  // extension (normal-mapped shading); a missing or unreadable map is no error
            texture::Texture nm;
            if (!load_texture(res_dir, m.normal_texture, nm)) mat.normal_texture = std::move(nm);
        }
        for (int k = 0; k < 3; k++) { mat.ambient[k] = m.ambient[k]; mat.diffuse[k] = m.diffuse[k]; mat.specular[k] = m.specular[k]; }
        mat.shininess = m.shininess;
        out.materials.push_back(std::move(mat));
    }

    for (const auto &m : raw_meshes) {
        model::Mesh mesh;
        mesh.name = file_name;  // resources.rs:254
        const size_t n_verts = m.positions.size() / 3;
        // resources.rs:226 indexes texcoords unconditionally: a mesh without vt panics there.
        if (m.texcoords.size() < 2 * n_verts)
            return {RWR_ERR_PARSE, "mesh '" + m.name + "' lacks texture coordinates for every vertex (index out of bounds in the reference)"};
        mesh.vertex_buffer.reserve(n_verts);
        for (size_t i = 0; i < n_verts; i++) {
            const float p[3] = {m.positions[3 * i], m.positions[3 * i + 1], m.positions[3 * i + 2]};
            const float t[2] = {m.texcoords[2 * i], m.texcoords[2 * i + 1]};
            mesh.vertex_buffer.push_back(make_vertex(p, t));
        }
        for (size_t i = 0; i + 2 < m.indices.size(); i += 3) mesh.index_buffer.push_back(make_face(m.indices[i], m.indices[i + 1], m.indices[i + 2]));
        mesh.num_elements = (uint32_t)m.indices.size();
        mesh.material = 0;
        if (m.has_material)
            for (size_t k = 0; k < raw_materials.size(); k++)
                if (raw_materials[k].name == m.material_name) mesh.material = k;
        out.meshes.push_back(std::move(mesh));
    }
    return {};
}

}  // namespace resources
}  // namespace rwr
