// models::triangle::Triangle — /root/reference/src/models/triangle/triangle.rs:4-130 (the single-triangle
// model; the reference builds the type but State never creates or dispatches one).  The reference object
// owns a 48-byte uniform buffer + a compute pipeline built from triangle/compute.wgsl; here it is the
// uniform's contents only: the pass lives in librwr_hip.so (kernels_dormant.hip) and the buffer is
// uploaded with rwr_scene_set_triangles.
#pragma once

#include "../../../include/rwr_hip.h"
#include "../camera.hpp"

namespace rwr {
namespace models {

struct TriangleBufferData {  // triangle.rs:10-19
    float p0[3]; float pad0;
    float p1[3]; float pad1;
    float p2[3]; float pad2;
};
static_assert(sizeof(TriangleBufferData) == sizeof(rwr_triangle_buffer_data), "same layout as the C ABI POD");

class Triangle {
public:
    Triangle(Vector3 p0, Vector3 p1, Vector3 p2)  // triangle.rs:37; TriangleBufferData::new, triangle.rs:22-33
        : data_{{p0.x, p0.y, p0.z}, 0.0f, {p1.x, p1.y, p1.z}, 0.0f, {p2.x, p2.y, p2.z}, 0.0f} {}
    const TriangleBufferData &get_buffer() const { return data_; }

private:
    TriangleBufferData data_;
};

}  // namespace models
}  // namespace rwr
