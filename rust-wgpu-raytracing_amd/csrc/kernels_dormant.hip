// The dormant parts of the reference, as one plain frame kernel:
//   * the single-triangle model, /root/reference/src/models/triangle/{triangle.rs, compute.wgsl} — built
//     but never dispatched by State::render;
//   * pixelToRay_ortho, defined in all three shaders (triangle_list/compute.wgsl:166-174,
//     sphere/compute.wgsl:103-111, triangle/compute.wgsl:143-151) and called by none.
// A frame that uses either (rwr_scene_set_triangles, RWR_FLAG_ORTHO_RAYS) is rendered here: one pixel
// per lane, every pass brute force in the reference's order (spheres, then single triangles — this
// project's choice of position — then the mesh), the same literal arithmetic as the frame kernels.  No
// culling: orthographic rays do not share an origin, and this path is not on the benchmark.
#include "rwr_primary.h"

namespace rwr {

// triangleRayIntersect of the single-triangle model (triangle/compute.wgsl:65-125): plane + three
// inclusive edge tests; the HitRecord's normal is N flipped towards the ray and NOT normalised.
RWR_DEV bool single_triangle_intersect(const rwr_triangle_buffer_data &tr, f3 O, f3 D, float &t_out, float &ndotd_out)
{
    const f3 p0 = ld3(tr.p0), p1 = ld3(tr.p1), p2 = ld3(tr.p2);
    const f3 N = cross3(sub3(p1, p0), sub3(p2, p0));
    const float ndotd = dot3(N, D);
    bool hit = !(fabsf(ndotd) < kEpsilon);
    const float d = -dot3(N, p0);
    const float t = -(dot3(N, O) + d) / ndotd;
    hit &= !(t < 0.0f);
    const f3 P = along(O, t, D);
    hit &= !(dot3(N, cross3(sub3(p1, p0), sub3(P, p0))) < 0.0f);
    hit &= !(dot3(N, cross3(sub3(p2, p1), sub3(P, p1))) < 0.0f);
    hit &= !(dot3(N, cross3(sub3(p0, p2), sub3(P, p2))) < 0.0f);
    t_out = t;
    ndotd_out = ndotd;
    return hit;
}

template <bool AUX>
__global__ void __launch_bounds__(256)
k_primary_dormant(const FrameParams p, const SingleTriangles st, const TriRecord *__restrict__ tris, const ShadeRec *__restrict__ shade,
                  const float4 *__restrict__ tex, const Targets tg)
{
    const uint32_t px = blockIdx.x * 32u + (threadIdx.x & 31u), py = p.row_begin + blockIdx.y * p.row_pitch + (threadIdx.x >> 5);
    const bool in_range = (px < p.width) && (py < p.row_end);

    f3 O = ld3(p.cam.origin), D;
    if (p.flags & RWR_FLAG_ORTHO_RAYS) {  // pixelToRay_ortho
        const float x_nds = 2.0f * ((float)px + 0.5f) / (float)p.width - 1.0f;
        const float y_nds = 2.0f * ((float)py + 0.5f) / (float)p.height - 1.0f;
        O = mk3(O.x + x_nds * 5.0f, O.y + y_nds * 5.0f, O.z + 0.0f);
        D = mk3(0.0f, 0.0f, -1.0f);
    } else {
        D = pixel_to_ray_dir(p.cam, px, py, 0.5f, 0.5f, p.width, p.height);
    }

    float depth_tex = 0.0f, win_t = 0.0f, win_ndotd = 0.0f;
    int32_t obj = -1;
    for (uint32_t s = 0; s < p.n_spheres; s++) {  // sphere/compute.wgsl:114-158
        float t;
        if (sphere_ray_intersect_t(ld3(p.spheres[s].center), p.spheres[s].radius, O, D, t)) {
            const float depth = to_non_linear_depth(t);
            if (!(depth >= 1.0f - depth_tex)) { depth_tex = 1.0f - depth; obj = -2 - (int32_t)s; win_t = t; }
        }
    }
    for (uint32_t k = 0; k < st.n; k++) {  // triangle/compute.wgsl:153-195
        float t, ndotd;
        if (single_triangle_intersect(st.t[k], O, D, t, ndotd)) {
            const float depth = to_non_linear_depth(t);
            if (!(depth >= 1.0f - depth_tex)) { depth_tex = 1.0f - depth; obj = -10 - (int32_t)k; win_t = t; win_ndotd = ndotd; }
        }
    }
    MeshHit best;
    best.have = false; best.t = 0.0f; best.u = 0.0f; best.v = 0.0f; best.ndotd = 0.0f; best.idx = 0u;
    for (uint32_t i = 0; i < p.n_tris; i++) intersect_and_select(tris[i], i, O, D, best);  // compute.wgsl:186-202
    if (best.have) {
        const float depth = to_non_linear_depth(best.t);
        if (!(depth >= 1.0f - depth_tex)) { depth_tex = 1.0f - depth; obj = (int32_t)best.idx; win_t = best.t; }
    }

    float cr = 0.0f, cg = 0.0f, cb = 0.0f, ca = 0.0f;
    if (obj <= -10) {
        // triangle/compute.wgsl:171-187: the sphere's shading with the un-normalised, ray-facing N
        const rwr_triangle_buffer_data &tr = st.t[-10 - obj];
        f3 N = cross3(sub3(ld3(tr.p1), ld3(tr.p0)), sub3(ld3(tr.p2), ld3(tr.p0)));
        if (win_ndotd > 0.0f) N = neg3(N);
        const f3 c = shade_sphere(N, D);
        cr = c.x; cg = c.y; cb = c.z; ca = 2.0f;
    } else if (obj != -1) {
        const f3 c = ((p.flags & RWR_FLAG_NORMAL_MAP) ? shade_winner<true>(p, obj, win_t, best.u, best.v, best.ndotd, shade, tex, O, D) : shade_winner<false>(p, obj, win_t, best.u, best.v, best.ndotd, shade, tex, O, D)).colour;
        cr = c.x; cg = c.y; cb = c.z; ca = 2.0f;
    }
    if (in_range) {
        const size_t o = (size_t)py * p.width + px;
        reinterpret_cast<uint32_t *>(tg.color)[o] = pack_rgba8(cr, cg, cb, ca);
        tg.depth[o] = depth_tex;
        if (AUX) {
            reinterpret_cast<float4 *>(tg.color_f32)[o] = make_float4(cr, cg, cb, ca);
            tg.obj_id[o] = obj;
            tg.hit_t[o] = win_t;
        }
    }
}

hipError_t launch_primary_dormant(hipStream_t s, const FrameParams &fp, const SingleTriangles &st, const TriRecord *tris,
                                  const ShadeRec *shade, const float4 *tex, const Targets &tg)
{
    if (fp.row_end <= fp.row_begin || fp.width == 0) return hipSuccess;
    const dim3 grid((fp.width + 31u) / 32u, band_strips(fp));
    if (fp.flags & RWR_FLAG_AUX_OUTPUTS) hipLaunchKernelGGL(k_primary_dormant<true>, grid, dim3(256), 0, s, fp, st, tris, shade, tex, tg);
    else hipLaunchKernelGGL(k_primary_dormant<false>, grid, dim3(256), 0, s, fp, st, tris, shade, tex, tg);
    return hipGetLastError();
}

}  // namespace rwr
