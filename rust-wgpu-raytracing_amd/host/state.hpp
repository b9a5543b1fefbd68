// Frame orchestrator — the role of `State` in /root/reference/src/lib.rs:223-1231, with the wgpu
// plumbing replaced by one rwr_context (C ABI, include/rwr_hip.h).
//   State::new     lib.rs:260-770   (camera literal 352-361, spheres 532-534, scene 559-568)
//   State::resize  lib.rs:772-989
//   State::input   lib.rs:990-992
//   State::update  lib.rs:994-1010
//   State::render  lib.rs:1012-1230
// There is no window: `size` is given, events come from the caller, and `render` leaves the frame
// in device memory (read it back or write a PNG with `present`).
#pragma once

#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "camera.hpp"
#include "circle_camera_control.hpp"
#include "resources.hpp"

namespace rwr {

// The reference's two model objects as State holds them (src/models/): each owns wgpu buffers and a compute pipeline
// there; here they are what those buffers contain — the pipelines are the fused HIP kernel behind rwr_render.
namespace models {

// models::sphere::Sphere (sphere.rs:4-133): the 16-byte uniform, uploaded with rwr_scene_set_spheres
class Sphere {
public:
    Sphere(float radius, Vector3 center) : data_{{center.x, center.y, center.z}, radius} {}  // sphere.rs:18-23
    const SphereBufferData &get_buffer() const { return data_; }                              // sphere.rs:123-125

private:
    SphereBufferData data_;
};

// models::triangle_list::TriangleList (triangle_list.rs:6-250): takes the Model by value (:79), derives MaterialData from
// materials[0] (:212) and exposes meshes[0]'s vertex / index buffers and materials[0]'s texture (:228-246) — the pieces
// rwr_scene_upload_mesh receives
class TriangleList {
public:
    explicit TriangleList(model::Model m) : model_(std::move(m))
    {
        const model::Material &mat = model_.materials.at(0);
        material_ = MaterialData{{mat.ambient[0], mat.ambient[1], mat.ambient[2]}, 0.0f,
                                 {mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]}, 0.0f,
                                 {mat.specular[0], mat.specular[1], mat.specular[2]}, 0.0f};
    }
    const std::vector<ModelVertexSmall> &get_vertex_buffer() const { return model_.meshes.at(0).vertex_buffer; }
    const std::vector<ModelFaceSmall> &get_index_buffer() const { return model_.meshes.at(0).index_buffer; }
    const MaterialData &get_material_buffer() const { return material_; }
    const texture::Texture &get_texture() const { return model_.materials.at(0).diffuse_texture; }

private:
    model::Model model_;
    MaterialData material_;
};

}  // namespace models

struct RwrFailure : std::runtime_error {
    int code;
    RwrFailure(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc)
{
    if (rc != RWR_OK) throw RwrFailure(rc, rwr_last_error_string());
}

class State {
public:
    // State::new(window) — the scene file is hard-coded in the reference (lib.rs:560); here it is an argument.
    State(uint32_t width, uint32_t height, const std::string &res_dir, const std::string &scene = "suzanne_lowpoly.obj",
          int device_id = 0)
        : size_{width, height},
          camera_controller_(std::make_unique<CircleCameraController>(0.2f)),  // lib.rs:361
          sphere_(0.4f, Vector3(0.6f, 0.5f, -4.0f)),                           // lib.rs:532
          sphere_front_(0.4f, Vector3(0.4f, 0.4f, -3.0f)),                     // lib.rs:534
          triangle_list_(load(res_dir, scene))
    {
        if (width == 0 || height == 0) throw RwrFailure(RWR_ERR_INVALID_ARGUMENT, "zero-sized window");
        camera_.eye = Point3(0.0f, 0.0f, 0.0f);  // lib.rs:352-360
        camera_.target = Point3(0.0f, 0.0f, -1.0f);
        camera_.up = Vector3::unit_y();
        camera_.aspect = (float)width / (float)height;
        camera_.fovy = 60.0f;
        camera_.znear = 0.1f;
        camera_.zfar = 100.0f;
        if (!camera_inv_uniform_.update_view_proj(camera_)) throw RwrFailure(RWR_ERR_INVALID_ARGUMENT, "singular camera");
        camera_uniform_.update_view_proj(camera_);

        check(rwr_ctx_create(device_id, &ctx_));
        try {
            const auto &tex = triangle_list_.get_texture();
            check(rwr_scene_upload_mesh(ctx_, triangle_list_.get_vertex_buffer().data(), (uint32_t)triangle_list_.get_vertex_buffer().size(),
                                        triangle_list_.get_index_buffer().data(), (uint32_t)triangle_list_.get_index_buffer().size(),
                                        &triangle_list_.get_material_buffer(), tex.rgba.data(), tex.width, tex.height));
            const SphereBufferData spheres[2] = {sphere_.get_buffer(), sphere_front_.get_buffer()};  // pass order: lib.rs:1106-1148
            check(rwr_scene_set_spheres(ctx_, spheres, 2));
            const rwr_screen screen{width, height};
            check(rwr_resize(ctx_, &screen));
        } catch (...) {
            rwr_ctx_destroy(ctx_);
            throw;
        }
    }
    ~State() { rwr_ctx_destroy(ctx_); }
    State(const State &) = delete;
    State &operator=(const State &) = delete;

    // lib.rs:772-989.  The reference recomputes camera.aspect from the OLD size (lib.rs:774 runs
    // before 776-777); that quirk is reproduced so that a scripted resize sequence matches.
    void resize(uint32_t new_width, uint32_t new_height)
    {
        if (new_width > 0 && new_height > 0) {
            camera_.aspect = (float)size_.width / (float)size_.height;
            size_ = rwr_screen{new_width, new_height};
            check(rwr_resize(ctx_, &size_));
        }
    }
    bool input(const KeyboardInput &event) { return camera_controller_->process_events(event); }  // lib.rs:990-992
    void update()                                                                                 // lib.rs:994-1010
    {
        camera_controller_->update_camera(camera_);
        camera_uniform_.update_view_proj(camera_);
        if (!camera_inv_uniform_.update_view_proj(camera_)) throw RwrFailure(RWR_ERR_INVALID_ARGUMENT, "singular camera");
    }
    void render(const rwr_render_params *params = nullptr) { check(rwr_render(ctx_, &camera_inv_uniform_, params)); }  // lib.rs:1012-1230

    // the blit's job (screenquad.wgsl + sRGB swapchain, lib.rs:1186-1224): framebuffer -> PNG
    void present(const std::string &png_path)
    {
        std::vector<uint8_t> rgba((size_t)size_.width * size_.height * 4);
        check(rwr_readback(ctx_, rgba.data(), nullptr, nullptr, nullptr, nullptr));
        check(rwr_write_png_rgba8(png_path.c_str(), rgba.data(), size_.width, size_.height, 1, 1));
    }

    Camera &camera() { return camera_; }
    const CameraInvUniform &camera_inv_uniform() const { return camera_inv_uniform_; }
    rwr_context *context() { return ctx_; }
    rwr_screen size() const { return size_; }

private:
    static model::Model load(const std::string &res_dir, const std::string &scene)
    {
        model::Model m;
        const resources::Error e = resources::load_model_compute(res_dir, scene, m);  // lib.rs:559-566 (.unwrap())
        if (e) throw RwrFailure(e.code, e.message);
        if (m.meshes.empty() || m.materials.empty()) throw RwrFailure(RWR_ERR_PARSE, scene + ": no mesh or no material");
        return m;
    }

    rwr_screen size_;
    Camera camera_;
    std::unique_ptr<CameraController> camera_controller_;
    CameraUniform camera_uniform_;
    CameraInvUniform camera_inv_uniform_;
    models::Sphere sphere_, sphere_front_;
    models::TriangleList triangle_list_;
    rwr_context *ctx_ = nullptr;
};

}  // namespace rwr
