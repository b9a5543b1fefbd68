// models::triangle_list::TriangleList — /root/reference/src/models/triangle_list/triangle_list.rs:6-250.
// Takes the Model by value (:79), derives MaterialData from materials[0] (:212) and exposes
// meshes[0]'s vertex / index buffers and materials[0]'s texture (:228-246) — exactly the pieces
// rwr_scene_upload_mesh receives.
#pragma once

#include <utility>

#include "../model.hpp"

namespace rwr {
namespace models {

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
}  // namespace rwr
