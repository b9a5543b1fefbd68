// Fused primary-ray kernel: ONE launch replaces the reference's three clears,
// two sphere passes, two depth copies and the mesh pass
// (/root/reference/src/lib.rs:1024-1184; shaders
//  src/models/sphere/compute.wgsl:114-158 and
//  src/models/triangle_list/compute.wgsl:177-240).
//
// MI355X mapping (DESIGN.md §"Kernels"):
//  * one wave64 = one 8x8 pixel tile, one lane per pixel (the reference runs one
//    single-lane workgroup per pixel); a 256-thread workgroup = 32x8 pixels, so
//    every 128-byte line of the RGBA8 and R32F targets is written whole by one
//    workgroup (one XCD L2), 4 B per lane, each pixel exactly once — the
//    "clear" is the miss value of the same store.
//  * depth compositing between the passes lives in registers; the depth
//    ping-pong textures and their copies disappear.
//  * per wave, lanes first test 64 faces at a time against the tile's ray
//    frustum (all primary rays share the camera origin), __ballot the
//    survivors, and the wave then walks the set bits in ascending face order:
//    the face index is wave-uniform, so its 128-byte TriRecord arrives through
//    scalar loads and sits in SGPRs while 64 rays are tested against it.
//    Ascending order + strict '<' reproduces the reference's lowest-index tie
//    rule; skipped faces are ones no ray of the tile can hit, so the result is
//    bit-identical to the brute-force loop (checked against RWR_FLAG_NO_CULL).
#include "rwr_device.h"

namespace rwr {

// ---------------------------------------------------------------------------
// Scene prebake: ModelVertexSmall[] + ModelFaceSmall[] (+ rigid instances)
// -> TriRecord[] + FaceUV[].  One thread per (instance, face).
__global__ void __launch_bounds__(256)
k_prebake(const rwr_model_vertex_small *__restrict__ verts, const rwr_model_face_small *__restrict__ faces,
          uint32_t n_faces, const rwr_instance_raw *__restrict__ instances, uint32_t n_instances,
          TriRecord *__restrict__ tris, FaceUV *__restrict__ face_uv)
{
    const uint32_t total = n_faces * (n_instances ? n_instances : 1u);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t inst = i / n_faces, f = i - inst * n_faces;
    const rwr_model_face_small face = faces[f];
    const rwr_model_vertex_small v0 = verts[face.indices[0]];
    const rwr_model_vertex_small v1 = verts[face.indices[1]];
    const rwr_model_vertex_small v2 = verts[face.indices[2]];
    f3 p0 = ld3(v0.position), p1 = ld3(v1.position), p2 = ld3(v2.position);
    if (n_instances) {
        // world = model * vec4(p, 1), WGSL mat*vec order
        const rwr_instance_raw &m = instances[inst];
        float w;
        f3 q;
        mat4_mul(m.model, p0.x, p0.y, p0.z, 1.0f, q.x, q.y, q.z, w); p0 = q;
        mat4_mul(m.model, p1.x, p1.y, p1.z, 1.0f, q.x, q.y, q.z, w); p1 = q;
        mat4_mul(m.model, p2.x, p2.y, p2.z, 1.0f, q.x, q.y, q.z, w); p2 = q;
    }
    const f3 v0v1 = sub3(p1, p0), v0v2 = sub3(p2, p0);
    const f3 N = cross3(v0v1, v0v2);
    const f3 e1 = sub3(p2, p1), e2 = sub3(p0, p2);
    TriRecord T;
    T.p0[0] = p0.x; T.p0[1] = p0.y; T.p0[2] = p0.z; T.d = -dot3(N, p0);
    T.p1[0] = p1.x; T.p1[1] = p1.y; T.p1[2] = p1.z; T.denom = dot3(N, N);
    T.p2[0] = p2.x; T.p2[1] = p2.y; T.p2[2] = p2.z; T.pad0 = 0.0f;
    T.N[0] = N.x; T.N[1] = N.y; T.N[2] = N.z; T.pad1 = 0.0f;
    T.e0[0] = v0v1.x; T.e0[1] = v0v1.y; T.e0[2] = v0v1.z; T.pad2 = 0.0f;
    T.e1[0] = e1.x; T.e1[1] = e1.y; T.e1[2] = e1.z; T.pad3 = 0.0f;
    T.e2[0] = e2.x; T.e2[1] = e2.y; T.e2[2] = e2.z; T.pad4 = 0.0f;
    T.pad5[0] = T.pad5[1] = T.pad5[2] = T.pad5[3] = 0.0f;
    tris[i] = T;
    FaceUV U;
    U.uv0[0] = v0.tex_coords[0]; U.uv0[1] = v0.tex_coords[1];
    U.uv1[0] = v1.tex_coords[0]; U.uv1[1] = v1.tex_coords[1];
    U.uv2[0] = v2.tex_coords[0]; U.uv2[1] = v2.tex_coords[1];
    U.pad[0] = U.pad[1] = 0.0f;
    face_uv[i] = U;
}

hipError_t launch_prebake(hipStream_t s, const rwr_model_vertex_small *verts, const rwr_model_face_small *faces,
                          uint32_t n_faces, const rwr_instance_raw *instances, uint32_t n_instances,
                          TriRecord *tris, FaceUV *face_uv)
{
    const uint32_t total = n_faces * (n_instances ? n_instances : 1u);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prebake, dim3((total + 255) / 256), dim3(256), 0, s, verts, faces, n_faces, instances,
                       n_instances, tris, face_uv);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Tile frustum: the four side planes (through the shared ray origin) of the
// pyramid spanned by an 8x8 pixel tile.  Normals point inward.
struct TileFrustum {
    f3 n[4];
    float l1[4];  // |n|_1, for the rounding margin
};

RWR_DEV TileFrustum make_tile_frustum(const rwr_camera_inv_uniform &cam, float x0, float y0, float x1, float y1,
                                      float width, float height)
{
    const f3 c00 = ray_dir_unnormalized(cam, x0, y0, width, height);
    const f3 c10 = ray_dir_unnormalized(cam, x1, y0, width, height);
    const f3 c11 = ray_dir_unnormalized(cam, x1, y1, width, height);
    const f3 c01 = ray_dir_unnormalized(cam, x0, y1, width, height);
    const f3 mid = ray_dir_unnormalized(cam, 0.5f * (x0 + x1), 0.5f * (y0 + y1), width, height);
    TileFrustum fr;
    fr.n[0] = cross3(c00, c10);
    fr.n[1] = cross3(c10, c11);
    fr.n[2] = cross3(c11, c01);
    fr.n[3] = cross3(c01, c00);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (dot3(fr.n[k], mid) < 0.0f) fr.n[k] = neg3(fr.n[k]);
        fr.l1[k] = fabsf(fr.n[k].x) + fabsf(fr.n[k].y) + fabsf(fr.n[k].z);
    }
    return fr;
}

// True when the triangle lies entirely outside one side plane, with a margin
// (relative 2e-5 on L1 norms, >100x the f32 rounding of the hit test) so that no
// face the exact test could accept for a ray inside the tile is ever dropped.
// NaNs compare false => "keep".
RWR_DEV bool tile_culls_triangle(const TileFrustum &fr, f3 q0, f3 q1, f3 q2)
{
    constexpr float kRel = 2e-5f;
    const float m0 = kRel * (fabsf(q0.x) + fabsf(q0.y) + fabsf(q0.z));
    const float m1 = kRel * (fabsf(q1.x) + fabsf(q1.y) + fabsf(q1.z));
    const float m2 = kRel * (fabsf(q2.x) + fabsf(q2.y) + fabsf(q2.z));
    bool culled = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float s0 = dot3(fr.n[k], q0), s1 = dot3(fr.n[k], q1), s2 = dot3(fr.n[k], q2);
        culled |= (s0 < -m0 * fr.l1[k]) && (s1 < -m1 * fr.l1[k]) && (s2 < -m2 * fr.l1[k]);
    }
    return culled;
}

// ---------------------------------------------------------------------------
template <bool AUX, bool CULL>
__global__ void __launch_bounds__(256)
k_primary(const FrameParams p, const TriRecord *__restrict__ tris, const FaceUV *__restrict__ face_uv,
          const uint32_t *__restrict__ tex, const float *__restrict__ srgb_lut, const Targets tg)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = srgb_lut[threadIdx.x];
    __syncthreads();

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t tile_x0 = blockIdx.x * 32u + wave * 8u;
    const uint32_t tile_y0 = p.row_begin + blockIdx.y * 8u;
    const uint32_t px = tile_x0 + (lane & 7u), py = tile_y0 + (lane >> 3);
    const bool in_range = (px < p.width) && (py < p.row_end);

    const f3 O = ld3(p.cam.origin);
    const f3 D = pixel_to_ray_dir(p.cam, px, py, 0.5f, 0.5f, p.width, p.height);

    // Framebuffer state of this pixel, as the reference's cleared textures hold it.
    float depth_tex = 0.0f;                       // depth_texture_* after the clear (lib.rs:1024-1104)
    float cr = 0.0f, cg = 0.0f, cb = 0.0f, ca = 0.0f;  // screen_texture after the clear
    int32_t obj = -1;
    float hit_t = 0.0f;

    // -- analytic sphere passes, in order (lib.rs:1106-1173) -----------------
    for (uint32_t s = 0; s < p.n_spheres; s++) {
        float t;
        f3 n;
        if (sphere_ray_intersect(ld3(p.spheres[s].center), p.spheres[s].radius, O, D, t, n)) {
            const float current_depth = 1.0f - depth_tex;  // sphere/compute.wgsl:130
            const float depth = to_non_linear_depth(t);
            if (!(depth >= current_depth)) {
                const f3 c = shade_sphere(n, D);
                cr = c.x; cg = c.y; cb = c.z; ca = 2.0f;
                depth_tex = 1.0f - depth;
                obj = -2 - (int32_t)s;
                hit_t = t;
            }
        }
    }

    // -- mesh pass (lib.rs:1174-1184) -----------------------------------------
    MeshHit best;
    best.have = false; best.t = 0.0f; best.u = 0.0f; best.v = 0.0f; best.ndotd = 0.0f; best.idx = 0u;

    if (p.n_tris) {
        TileFrustum fr;
        if (CULL) {
            const float fx0 = (float)tile_x0, fy0 = (float)tile_y0;
            fr = make_tile_frustum(p.cam, fx0, fy0, fx0 + 8.0f, fy0 + 8.0f, (float)p.width, (float)p.height);
        }
        for (uint32_t base = 0; base < p.n_tris; base += 64u) {
            const uint32_t j = base + lane;
            bool keep = j < p.n_tris;
            if (CULL && keep) {
                const TriRecord &T = tris[j];
                keep = !tile_culls_triangle(fr, sub3(ld3(T.p0), O), sub3(ld3(T.p1), O), sub3(ld3(T.p2), O));
            }
            unsigned long long mask = __ballot(keep);
            while (mask) {
                const uint32_t b = (uint32_t)__builtin_ctzll(mask);
                mask &= mask - 1ull;
                const uint32_t idx = base + b;  // wave-uniform: record comes in through scalar loads
                intersect_and_select(tris[idx], idx, O, D, best);
            }
        }
    }

    if (best.have) {
        const float current_depth = 1.0f - depth_tex;  // compute.wgsl:210
        const float depth = to_non_linear_depth(best.t);
        if (!(depth >= current_depth)) {
            const TriRecord &T = tris[best.idx];
            f3 N = ld3(T.N);
            if (best.ndotd > 0.0f) N = neg3(N);          // compute.wgsl:140-142
            const float u = best.u / T.denom, v = best.v / T.denom;
            const f3 n = normalize3(N);
            const f3 c = shade_mesh(face_uv[best.idx], u, v, 1.0f - u - v, n, D, p.ambient, p.specular, tex, p.tex_w,
                                    p.tex_h, s_lut, nullptr);
            cr = c.x; cg = c.y; cb = c.z; ca = 2.0f;
            depth_tex = 1.0f - depth;
            obj = (int32_t)best.idx;
            hit_t = best.t;
        }
    }

    if (in_range) {
        const size_t o = (size_t)py * p.width + px;
        reinterpret_cast<uint32_t *>(tg.color)[o] = pack_rgba8(cr, cg, cb, ca);
        tg.depth[o] = depth_tex;
        if (AUX) {
            reinterpret_cast<float4 *>(tg.color_f32)[o] = make_float4(cr, cg, cb, ca);
            tg.obj_id[o] = obj;
            tg.hit_t[o] = hit_t;
        }
    }
}

hipError_t launch_primary(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const FaceUV *face_uv,
                          const uint32_t *tex, const float *srgb_lut, const Targets &tg)
{
    if (fp.row_end <= fp.row_begin || fp.width == 0) return hipSuccess;
    const dim3 grid((fp.width + 31u) / 32u, (fp.row_end - fp.row_begin + 7u) / 8u);
    const dim3 block(256);
    const bool aux = (fp.flags & RWR_FLAG_AUX_OUTPUTS) != 0;
    const bool cull = (fp.flags & RWR_FLAG_NO_CULL) == 0;
    if (aux && cull) hipLaunchKernelGGL((k_primary<true, true>), grid, block, 0, s, fp, tris, face_uv, tex, srgb_lut, tg);
    else if (aux) hipLaunchKernelGGL((k_primary<true, false>), grid, block, 0, s, fp, tris, face_uv, tex, srgb_lut, tg);
    else if (cull) hipLaunchKernelGGL((k_primary<false, true>), grid, block, 0, s, fp, tris, face_uv, tex, srgb_lut, tg);
    else hipLaunchKernelGGL((k_primary<false, false>), grid, block, 0, s, fp, tris, face_uv, tex, srgb_lut, tg);
    return hipGetLastError();
}

}  // namespace rwr
