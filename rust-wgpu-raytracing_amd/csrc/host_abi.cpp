// C-ABI wrappers over the pure-CPU host surface (camera maths, controller,
// OBJ/MTL/texture loader, image decoding, instance grid).  The implementations
// live in ../host/*.hpp so that C++ callers can use them directly; these entry
// points serve FFI callers (Rust, ctypes).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../host/camera.hpp"
#include "../host/circle_camera_control.hpp"
#include "../host/resources.hpp"
#include "rwr_internal.h"

using namespace rwr;

struct rwr_model {
    model::Model model;
};

extern "C" {

int rwr_camera_build_inv_uniform(const rwr_camera *camera, rwr_camera_inv_uniform *out)
{
    if (!camera || !out) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    CameraInvUniform u;
    if (!u.update_view_proj(Camera::from_c(*camera)))
        return set_error(RWR_ERR_INVALID_ARGUMENT, "camera matrix is singular (the reference panics in invert().unwrap())");
    std::memcpy(out, static_cast<const rwr_camera_inv_uniform *>(&u), sizeof *out);
    return RWR_OK;
}

int rwr_circle_controller_update(float speed, uint32_t pressed_keys, rwr_camera *camera)
{
    if (!camera) return set_error(RWR_ERR_INVALID_ARGUMENT, "camera is NULL");
    CircleCameraController ctl(speed);
    ctl.set_pressed_mask(pressed_keys);
    Camera c = Camera::from_c(*camera);
    ctl.update_camera(c);
    *camera = c.to_c();
    return RWR_OK;
}

int rwr_load_model_compute(const char *res_dir, const char *file_name, rwr_model **out_model)
{
    if (!res_dir || !file_name || !out_model) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_model = nullptr;
    rwr_model *m = new (std::nothrow) rwr_model();
    if (!m) return set_error(RWR_ERR_IO, "out of memory");
    resources::Error e = resources::load_model_compute(res_dir, file_name, m->model);
    if (e) {
        delete m;
        return set_error(e.code, "%s", e.message.c_str());
    }
    // TriangleList consumes meshes[0] / materials[0] without checking (triangle_list.rs:212,229)
    if (m->model.meshes.empty() || m->model.materials.empty()) {
        delete m;
        return set_error(RWR_ERR_PARSE, "%s: no mesh or no material (index out of bounds in the reference)", file_name);
    }
    *out_model = m;
    return RWR_OK;
}

void rwr_model_free(rwr_model *model) { delete model; }

int rwr_model_info(const rwr_model *model, uint32_t *n_meshes, uint32_t *n_materials, uint32_t *n_verts, uint32_t *n_faces,
                   uint32_t *tex_w, uint32_t *tex_h)
{
    if (!model) return set_error(RWR_ERR_INVALID_ARGUMENT, "model is NULL");
    const auto &m = model->model;
    if (n_meshes) *n_meshes = (uint32_t)m.meshes.size();
    if (n_materials) *n_materials = (uint32_t)m.materials.size();
    if (n_verts) *n_verts = (uint32_t)m.meshes[0].vertex_buffer.size();
    if (n_faces) *n_faces = (uint32_t)m.meshes[0].index_buffer.size();
    if (tex_w) *tex_w = m.materials[0].diffuse_texture.width;
    if (tex_h) *tex_h = m.materials[0].diffuse_texture.height;
    return RWR_OK;
}

const rwr_model_vertex_small *rwr_model_vertices(const rwr_model *model) { return model ? model->model.meshes[0].vertex_buffer.data() : nullptr; }
const rwr_model_face_small *rwr_model_faces(const rwr_model *model) { return model ? model->model.meshes[0].index_buffer.data() : nullptr; }
const uint8_t *rwr_model_texture_rgba8(const rwr_model *model) { return model ? model->model.materials[0].diffuse_texture.rgba.data() : nullptr; }

const rwr_material_data *rwr_model_material(const rwr_model *model)
{
    // MaterialData::new(materials[0].ambient, .diffuse, .specular) — triangle_list.rs:212
    static thread_local rwr_material_data md;
    if (!model) return nullptr;
    const auto &m = model->model.materials[0];
    md = rwr_material_data{{m.ambient[0], m.ambient[1], m.ambient[2]}, 0.0f, {m.diffuse[0], m.diffuse[1], m.diffuse[2]}, 0.0f,
                           {m.specular[0], m.specular[1], m.specular[2]}, 0.0f};
    return &md;
}

int rwr_model_part_count(const rwr_model *model, uint32_t *n_parts)
{
    if (!model || !n_parts) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    *n_parts = (uint32_t)model->model.meshes.size();
    return RWR_OK;
}

int rwr_model_part(const rwr_model *model, uint32_t part, const rwr_model_vertex_small **verts, uint32_t *n_verts,
                   const rwr_model_face_small **faces, uint32_t *n_faces, rwr_material_data *material,
                   const uint8_t **rgba8, uint32_t *tex_w, uint32_t *tex_h)
{
    if (!model) return set_error(RWR_ERR_INVALID_ARGUMENT, "model is NULL");
    const auto &m = model->model;
    if (part >= m.meshes.size()) return set_error(RWR_ERR_INVALID_ARGUMENT, "part %u out of range (%zu meshes)", part, m.meshes.size());
    const auto &mesh = m.meshes[part];
    if (mesh.material >= m.materials.size()) return set_error(RWR_ERR_PARSE, "mesh %u references material %zu of %zu", part, mesh.material, m.materials.size());
    const auto &mat = m.materials[mesh.material];   // mesh.material = material_id.unwrap_or(0), resources.rs:257
    if (verts) *verts = mesh.vertex_buffer.data();
    if (n_verts) *n_verts = (uint32_t)mesh.vertex_buffer.size();
    if (faces) *faces = mesh.index_buffer.data();
    if (n_faces) *n_faces = (uint32_t)mesh.index_buffer.size();
    if (material)
        *material = rwr_material_data{{mat.ambient[0], mat.ambient[1], mat.ambient[2]}, 0.0f, {mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]}, 0.0f,
                                      {mat.specular[0], mat.specular[1], mat.specular[2]}, 0.0f};
    if (rgba8) *rgba8 = mat.diffuse_texture.rgba.data();
    if (tex_w) *tex_w = mat.diffuse_texture.width;
    if (tex_h) *tex_h = mat.diffuse_texture.height;
    return RWR_OK;
}

int rwr_model_part_normal_map(const rwr_model *model, uint32_t part, const uint8_t **rgba8, uint32_t *tex_w, uint32_t *tex_h)
{
    if (!model || !rgba8 || !tex_w || !tex_h) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    const auto &m = model->model;
    if (part >= m.meshes.size()) return set_error(RWR_ERR_INVALID_ARGUMENT, "part %u out of range (%zu meshes)", part, m.meshes.size());
    if (m.meshes[part].material >= m.materials.size()) return set_error(RWR_ERR_PARSE, "mesh %u references a missing material", part);
    const auto &nm = m.materials[m.meshes[part].material].normal_texture;
    *rgba8 = nm.rgba.empty() ? nullptr : nm.rgba.data();
    *tex_w = nm.width;
    *tex_h = nm.height;
    return RWR_OK;
}

int rwr_scene_upload_model_all(rwr_context *ctx, const rwr_model *model)
{
    if (!ctx || !model) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = rwr_scene_clear(ctx);
    for (uint32_t i = 0; rc == RWR_OK && i < model->model.meshes.size(); i++) {
        const rwr_model_vertex_small *v; const rwr_model_face_small *f; const uint8_t *t;
        uint32_t nv, nf, tw, th;
        rwr_material_data md;
        rc = rwr_model_part(model, i, &v, &nv, &f, &nf, &md, &t, &tw, &th);
        if (rc == RWR_OK && nf == 0) continue;   // an empty mesh adds no part
        if (rc == RWR_OK) rc = rwr_scene_add_mesh(ctx, v, nv, f, nf, &md, t, tw, th);
        const uint8_t *nm = nullptr;
        uint32_t nw = 0, nh = 0;
        if (rc == RWR_OK) rc = rwr_model_part_normal_map(model, i, &nm, &nw, &nh);
        if (rc == RWR_OK && nm) {   // the part just added is the scene's last
            uint32_t n_parts = 0;
            rc = rwr_scene_part_count(ctx, &n_parts);
            if (rc == RWR_OK) rc = rwr_scene_set_normal_map(ctx, n_parts - 1u, nm, nw, nh);
        }
    }
    if (rc == RWR_OK) rc = rwr_scene_commit(ctx);
    return rc;
}

int rwr_scene_upload_model(rwr_context *ctx, const rwr_model *model)
{
    if (!ctx || !model) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    const auto &mesh = model->model.meshes[0];
    const auto &tex = model->model.materials[0].diffuse_texture;
    int rc = rwr_scene_upload_mesh(ctx, mesh.vertex_buffer.data(), (uint32_t)mesh.vertex_buffer.size(), mesh.index_buffer.data(),
                                   (uint32_t)mesh.index_buffer.size(), rwr_model_material(model), tex.rgba.data(), tex.width, tex.height);
    const auto &nm = model->model.materials[0].normal_texture;   // extension: only RWR_FLAG_NORMAL_MAP renders look at it
    if (rc == RWR_OK && !nm.rgba.empty() && !mesh.index_buffer.empty()) rc = rwr_scene_set_normal_map(ctx, 0, nm.rgba.data(), nm.width, nm.height);
    return rc;
}

int rwr_decode_image_rgba8(const uint8_t *bytes, size_t n_bytes, uint8_t **out_rgba, uint32_t *out_w, uint32_t *out_h)
{
    if (!bytes || !out_rgba || !out_w || !out_h) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_rgba = nullptr;
    codec::Image img;
    std::string err;
    if (!codec::decode_image(bytes, n_bytes, img, err)) return set_error(RWR_ERR_PARSE, "%s", err.c_str());
    uint8_t *p = static_cast<uint8_t *>(std::malloc(img.rgba.size() ? img.rgba.size() : 1));
    if (!p) return set_error(RWR_ERR_IO, "out of memory");
    std::memcpy(p, img.rgba.data(), img.rgba.size());
    *out_rgba = p;
    *out_w = img.width;
    *out_h = img.height;
    return RWR_OK;
}

void rwr_free(void *p) { std::free(p); }

int rwr_write_png_rgba8(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height, int flip_vertical,
                        int encode_srgb)
{
    if (!path || !rgba8 || width == 0 || height == 0) return set_error(RWR_ERR_INVALID_ARGUMENT, "bad PNG arguments");
    uint8_t lut[256];
    for (int i = 0; i < 256; i++) {
        const double l = i / 255.0;
        const double e = l <= 0.0031308 ? 12.92 * l : 1.055 * std::pow(l, 1.0 / 2.4) - 0.055;
        lut[i] = encode_srgb ? (uint8_t)std::lround(e * 255.0) : (uint8_t)i;
    }
    std::vector<uint8_t> img((size_t)width * height * 4);
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t *src = rgba8 + (size_t)(flip_vertical ? height - 1 - y : y) * width * 4;
        uint8_t *dst = img.data() + (size_t)y * width * 4;
        for (uint32_t x = 0; x < width; x++) {
            dst[4 * x + 0] = lut[src[4 * x + 0]];
            dst[4 * x + 1] = lut[src[4 * x + 1]];
            dst[4 * x + 2] = lut[src[4 * x + 2]];
            dst[4 * x + 3] = src[4 * x + 3];
        }
    }
    const std::vector<uint8_t> png = codec::encode_png_rgba8(img.data(), width, height);
    FILE *f = std::fopen(path, "wb");
    if (!f) return set_error(RWR_ERR_IO, "cannot open %s for writing", path);
    const size_t put = std::fwrite(png.data(), 1, png.size(), f);
    std::fclose(f);
    if (put != png.size()) return set_error(RWR_ERR_IO, "short write on %s", path);
    return RWR_OK;
}

int rwr_make_instance_grid(uint32_t per_row, float space_between, rwr_instance_raw *out)
{
    if (!out || per_row == 0) return set_error(RWR_ERR_INVALID_ARGUMENT, "bad instance grid arguments");
    // lib.rs:400-421
    for (uint32_t z = 0; z < per_row; z++) {
        for (uint32_t x = 0; x < per_row; x++) {
            const float px = space_between * ((float)x - (float)per_row / 2.0f);
            const float pz = space_between * ((float)z - (float)per_row / 2.0f);
            Instance inst;
            inst.position = Vector3(px, 0.0f, pz);
            inst.rotation = inst.position.is_zero() ? Quaternion::from_axis_angle(Vector3::unit_z(), 0.0f)
                                                    : Quaternion::from_axis_angle(inst.position.normalize(), 45.0f);
            out[z * per_row + x] = inst.to_raw();
        }
    }
    return RWR_OK;
}

}  // extern "C"
