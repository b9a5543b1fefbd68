// Host-side mirror of the reference's model/texture types (CPU side only; the
// GPU copies live inside the rwr_context).
//   ModelVertexSmall / ModelFaceSmall   /root/reference/src/model.rs:45-79
//   Material / Mesh / Model             /root/reference/src/model.rs:108-128
//   Texture::from_bytes / from_image    /root/reference/src/texture.rs:98-166
#pragma once

#include <array>
#include <string>
#include <vector>

#include "../../include/rwr_hip.h"
#include "image_codec.hpp"

namespace rwr {

using ModelVertexSmall = rwr_model_vertex_small;
using ModelFaceSmall = rwr_model_face_small;
using MaterialData = rwr_material_data;       // triangle_list.rs:24-33
using SphereBufferData = rwr_sphere_buffer_data;  // sphere.rs:10-15
static_assert(sizeof(ModelVertexSmall) == 32 && sizeof(ModelFaceSmall) == 16 && sizeof(MaterialData) == 48, "POD layout");

inline ModelVertexSmall make_vertex(const float (&position)[3], const float (&tex_coords)[2])
{
    return ModelVertexSmall{{position[0], position[1], position[2]}, 0.0f, {tex_coords[0], tex_coords[1]}, {0.0f, 0.0f}};
}
inline ModelFaceSmall make_face(uint32_t a, uint32_t b, uint32_t c) { return ModelFaceSmall{{a, b, c}, 0u}; }

namespace texture {
// texture.rs: the decoded RGBA8 image that the reference uploads as
// Rgba8UnormSrgb with a ClampToEdge / mag-Linear / min-Nearest sampler (:122,151-159).
struct Texture {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;  // row 0 = top row of the image file

    // Texture::from_bytes (:98-106).  On failure returns false with `err` set
    // (the reference propagates image::ImageError through anyhow).
    static bool from_bytes(const uint8_t *bytes, size_t n, Texture &out, std::string &err)
    {
        codec::Image img;
        if (!codec::decode_image(bytes, n, img, err)) return false;
        out.width = img.width;
        out.height = img.height;
        out.rgba = std::move(img.rgba);
        return true;
    }
};
}  // namespace texture

namespace model {

struct Material {
    std::string name;
    texture::Texture diffuse_texture;
    std::string diffuse_texture_file;
    std::string normal_texture_file;  // map_Bump: parsed, not consumed by the reference shader (resources.rs:189)
    texture::Texture normal_texture;  // extension: the decoded map_Bump image when the file is there (RWR_FLAG_NORMAL_MAP);
                                      // empty otherwise — the reference never opens it, so its absence is not an error
    std::array<float, 3> ambient{0, 0, 0};
    std::array<float, 3> diffuse{0, 0, 0};
    std::array<float, 3> specular{0, 0, 0};
    float shininess = 0.0f;
};

struct Mesh {
    std::string name;
    std::vector<ModelVertexSmall> vertex_buffer;  // STORAGE buffer contents, resources.rs:240-244
    std::vector<ModelFaceSmall> index_buffer;     // resources.rs:246-250
    uint32_t num_elements = 0;                    // indices.len(), resources.rs:256
    size_t material = 0;                          // material_id.unwrap_or(0), resources.rs:257
};

struct Model {
    std::vector<Mesh> meshes;
    std::vector<Material> materials;
};

}  // namespace model
}  // namespace rwr
