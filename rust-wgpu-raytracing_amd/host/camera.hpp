// Host-side mirror of the reference's camera surface (pure CPU, f32).
//   Camera                      /root/reference/src/camera.rs:3-30
//   OPENGL_TO_WGPU_MATRIX       /root/reference/src/lib.rs:31-37
//   CameraUniform               /root/reference/src/lib.rs:66-84
//   CameraInvUniform            /root/reference/src/lib.rs:86-112
// The reference leans on cgmath 0.18.0 (not vendored); its look_at_rh,
// perspective and Matrix4::invert are restated here from their published
// formulas, with Rust's evaluation order (no fused multiply-add: build with
// -ffp-contract=off).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <optional>

#include "../../include/rwr_hip.h"

namespace rwr {

struct Vector3 {
    float x = 0, y = 0, z = 0;
    Vector3() = default;
    Vector3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static Vector3 unit_y() { return {0.0f, 1.0f, 0.0f}; }
    static Vector3 unit_z() { return {0.0f, 0.0f, 1.0f}; }
    Vector3 operator+(Vector3 o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vector3 operator-(Vector3 o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vector3 operator*(float s) const { return {x * s, y * s, z * s}; }
    Vector3 operator-() const { return {-x, -y, -z}; }
    Vector3 &operator+=(Vector3 o) { return *this = *this + o; }
    Vector3 &operator-=(Vector3 o) { return *this = *this - o; }
    float dot(Vector3 o) const { return x * o.x + y * o.y + z * o.z; }
    Vector3 cross(Vector3 o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    float magnitude() const { return std::sqrt(dot(*this)); }
    // cgmath InnerSpace::normalize = normalize_to(1) = self * (1 / magnitude)
    Vector3 normalize() const { return *this * (1.0f / magnitude()); }
    bool is_zero() const { return x == 0.0f && y == 0.0f && z == 0.0f; }
};
using Point3 = Vector3;

// Column-major 4x4, m[col][row] — the layout of cgmath::Matrix4<f32> and of
// `[[f32; 4]; 4]`.
struct Matrix4 {
    float m[4][4] = {};

    static Matrix4 identity()
    {
        Matrix4 r;
        for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f;
        return r;
    }
    // Matrix4::new(c0r0, c0r1, ..., c3r3)
    static Matrix4 from_cols(const float (&c)[16])
    {
        Matrix4 r;
        std::memcpy(r.m, c, sizeof r.m);
        return r;
    }
    static Matrix4 from_translation(Vector3 v)
    {
        Matrix4 r = identity();
        r.m[3][0] = v.x; r.m[3][1] = v.y; r.m[3][2] = v.z;
        return r;
    }
    // Matrix4::look_to_rh(eye, dir, up); look_at_rh(eye, center, up) = look_to_rh(eye, center - eye, up)
    static Matrix4 look_at_rh(Point3 eye, Point3 center, Vector3 up)
    {
        const Vector3 f = (center - eye).normalize();
        const Vector3 s = f.cross(up).normalize();
        const Vector3 u = s.cross(f);
        const float c[16] = {s.x, u.x, -f.x, 0.0f, s.y, u.y, -f.y, 0.0f, s.z, u.z, -f.z, 0.0f,
                             -eye.dot(s), -eye.dot(u), eye.dot(f), 1.0f};
        return from_cols(c);
    }
    Matrix4 operator*(const Matrix4 &b) const
    {
        Matrix4 r;
        for (int c = 0; c < 4; c++)
            for (int i = 0; i < 4; i++)
                r.m[c][i] = m[0][i] * b.m[c][0] + m[1][i] * b.m[c][1] + m[2][i] * b.m[c][2] + m[3][i] * b.m[c][3];
        return r;
    }
    float determinant() const
    {
        float det = 0.0f;
        for (int c = 0; c < 4; c++) det += m[c][0] * cofactor(c, 0);
        return det;
    }
    // SquareMatrix::invert: adjugate / determinant; nullopt when singular
    // (the reference unwrap()s it, camera.rs:22-23,28-29).
    std::optional<Matrix4> invert() const
    {
        const float det = determinant();
        if (det == 0.0f || !std::isfinite(det)) return std::nullopt;
        const float inv_det = 1.0f / det;
        Matrix4 r;
        for (int c = 0; c < 4; c++)
            for (int i = 0; i < 4; i++) r.m[c][i] = cofactor(i, c) * inv_det;
        return r;
    }

private:
    float cofactor(int col, int row) const
    {
        float s[3][3];
        int cc = 0;
        for (int c = 0; c < 4; c++) {
            if (c == col) continue;
            int rr = 0;
            for (int r = 0; r < 4; r++) {
                if (r == row) continue;
                s[cc][rr++] = m[c][r];
            }
            cc++;
        }
        const float d = s[0][0] * (s[1][1] * s[2][2] - s[2][1] * s[1][2]) -
                        s[1][0] * (s[0][1] * s[2][2] - s[2][1] * s[0][2]) +
                        s[2][0] * (s[0][1] * s[1][2] - s[1][1] * s[0][2]);
        return ((col + row) & 1) ? -d : d;
    }
};

// cgmath::perspective(Deg(fovy), aspect, near, far)
inline Matrix4 perspective_deg(float fovy_deg, float aspect, float near, float far)
{
    const float fovy_rad = fovy_deg * (float)(3.14159265358979323846 / 180.0);
    const float f = 1.0f / std::tan(fovy_rad / 2.0f);
    Matrix4 r;
    r.m[0][0] = f / aspect;
    r.m[1][1] = f;
    r.m[2][2] = (far + near) / (near - far);
    r.m[2][3] = -1.0f;
    r.m[3][2] = (2.0f * far * near) / (near - far);
    return r;
}

// lib.rs:31-37
inline Matrix4 opengl_to_wgpu_matrix()
{
    const float c[16] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.0f, 0.5f, 1.0f};
    return Matrix4::from_cols(c);
}

// camera.rs:3-30
struct Camera {
    Point3 eye;
    Point3 target;
    Vector3 up;
    float aspect = 1.0f;
    float fovy = 60.0f;
    float znear = 0.1f;
    float zfar = 100.0f;

    Matrix4 build_view_projection_matrix() const
    {
        const Matrix4 view = Matrix4::look_at_rh(eye, target, up);
        const Matrix4 proj = perspective_deg(fovy, aspect, znear, zfar);
        return proj * view;
    }
    std::optional<Matrix4> build_view_inv_matrix() const { return Matrix4::look_at_rh(eye, target, up).invert(); }
    std::optional<Matrix4> build_proj_inv_matrix() const { return perspective_deg(fovy, aspect, znear, zfar).invert(); }

    static Camera from_c(const rwr_camera &c)
    {
        Camera k;
        k.eye = {c.eye[0], c.eye[1], c.eye[2]};
        k.target = {c.target[0], c.target[1], c.target[2]};
        k.up = {c.up[0], c.up[1], c.up[2]};
        k.aspect = c.aspect; k.fovy = c.fovy; k.znear = c.znear; k.zfar = c.zfar;
        return k;
    }
    rwr_camera to_c() const
    {
        return rwr_camera{{eye.x, eye.y, eye.z}, {target.x, target.y, target.z}, {up.x, up.y, up.z}, aspect, fovy, znear, zfar};
    }
};

// lib.rs:66-84 — uploaded every frame by the reference but bound to no shader.
struct CameraUniform {
    float view_proj[4][4];
    CameraUniform() { std::memcpy(view_proj, Matrix4::identity().m, sizeof view_proj); }
    void update_view_proj(const Camera &camera)
    {
        const Matrix4 vp = opengl_to_wgpu_matrix() * camera.build_view_projection_matrix();
        std::memcpy(view_proj, vp.m, sizeof view_proj);
    }
};

// lib.rs:86-112 — the uniform the compute shaders read (binding g0 b3).
struct CameraInvUniform : rwr_camera_inv_uniform {
    CameraInvUniform()
    {
        std::memcpy(viewmodel_inv, Matrix4::identity().m, sizeof viewmodel_inv);
        std::memcpy(proj_inv, Matrix4::identity().m, sizeof proj_inv);
        origin[0] = origin[1] = origin[2] = 0.0f;
        _padding = 0;
    }
    // Returns false where the reference would panic on `.invert().unwrap()`.
    bool update_view_proj(const Camera &camera)
    {
        const auto vi = camera.build_view_inv_matrix();
        const auto pi = camera.build_proj_inv_matrix();
        if (!vi || !pi) return false;
        std::memcpy(viewmodel_inv, vi->m, sizeof viewmodel_inv);
        const Matrix4 gp = opengl_to_wgpu_matrix() * *pi;  // NOT inverse(G * P): lib.rs:109
        std::memcpy(proj_inv, gp.m, sizeof proj_inv);
        origin[0] = camera.eye.x; origin[1] = camera.eye.y; origin[2] = camera.eye.z;
        return true;
    }
};
static_assert(sizeof(CameraInvUniform) == 144, "CameraInvUniform must stay 144 B");

// Quaternion::from_axis_angle + Matrix4::from(Quaternion), for Instance::to_raw (lib.rs:114-127).
struct Quaternion {
    float s = 1.0f;
    Vector3 v;
    static Quaternion from_axis_angle(Vector3 axis, float angle_deg)
    {
        const float half = angle_deg * (float)(3.14159265358979323846 / 180.0) * 0.5f;
        Quaternion q;
        q.s = std::cos(half);
        q.v = axis * std::sin(half);
        return q;
    }
    Matrix4 to_matrix() const
    {
        const float x2 = v.x + v.x, y2 = v.y + v.y, z2 = v.z + v.z;
        const float xx2 = x2 * v.x, xy2 = x2 * v.y, xz2 = x2 * v.z;
        const float yy2 = y2 * v.y, yz2 = y2 * v.z, zz2 = z2 * v.z;
        const float sy2 = y2 * s, sz2 = z2 * s, sx2 = x2 * s;
        const float c[16] = {1.0f - yy2 - zz2, xy2 + sz2, xz2 - sy2, 0.0f,
                             xy2 - sz2, 1.0f - xx2 - zz2, yz2 + sx2, 0.0f,
                             xz2 + sy2, yz2 - sx2, 1.0f - xx2 - yy2, 0.0f,
                             0.0f, 0.0f, 0.0f, 1.0f};
        return Matrix4::from_cols(c);
    }
};

// Instance / InstanceRaw, lib.rs:114-134.
struct Instance {
    Vector3 position;
    Quaternion rotation;
    rwr_instance_raw to_raw() const
    {
        const Matrix4 model = Matrix4::from_translation(position) * rotation.to_matrix();
        rwr_instance_raw r;
        std::memcpy(r.model, model.m, sizeof r.model);
        return r;
    }
};

}  // namespace rwr
