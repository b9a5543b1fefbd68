// models::sphere::Sphere — /root/reference/src/models/sphere/sphere.rs:4-133.
// The reference object owns a 16-byte uniform buffer + a compute pipeline built from
// sphere/compute.wgsl; here it is the uniform's contents only: the "pipeline" is the fused
// HIP kernel behind rwr_render, and the buffer is uploaded with rwr_scene_set_spheres.
#pragma once

#include "../camera.hpp"
#include "../model.hpp"

namespace rwr {
namespace models {

class Sphere {
public:
    Sphere(float radius, Vector3 center) : data_{{center.x, center.y, center.z}, radius} {}  // sphere.rs:18-23
    const SphereBufferData &get_buffer() const { return data_; }                              // sphere.rs:123-125

private:
    SphereBufferData data_;
};

}  // namespace models
}  // namespace rwr
