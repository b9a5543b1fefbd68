// Per-frame records and tables (FrameParams::ray_colp / ray_row / tnum, FrameTri): the work of one 256-thread block, shared
// by k_frame_setup (kernels_primary.hip: its own launch ahead of every render kernel) and by the frame kernel's fused form
// (kernels_primary_p2.hip: the first workgroups of the frame's ONE launch make the records, the others wait for them).
// Blocks [0, nb_tris) make one face each per thread (the culling record of rwr_cull.h and the ray-independent numerator of
// the plane distance), the blocks after them fill the ray tables, one column pair or one row per thread.
#pragma once

#include "rwr_cull.h"
#include "rwr_device.h"

namespace rwr {

RWR_DEV void frame_setup_block(uint32_t block, uint32_t n_blocks, const CullConsts &cc, const rwr_camera_inv_uniform &cam, uint32_t width,
                               uint32_t height, const CullRec *__restrict__ cull, const TriRecord *__restrict__ tris, uint32_t n_tris,
                               uint32_t nb_tris, const FrameSetupOut &out)
{
    {   // what the frame wants zeroed: all blocks together
        const uint32_t t = block * 256u + threadIdx.x, stride = n_blocks * 256u;
        for (uint32_t i = t; i < out.n_zero_a; i += stride) out.zero_a[i] = 0u;
        for (uint32_t i = t; i < out.n_zero_b; i += stride) out.zero_b[i] = 0u;
    }
    if (block >= nb_tris) {
        const uint32_t e = (block - nb_tris) * 256u + threadIdx.x;
        const float(&p)[4][4] = cam.proj_inv;
        if (e < out.ray_pairs) {
            // compute.wgsl:151-152 for columns 2e and 2e + 1 (pixel centre: + 0.5), then the first term of :155
            const float xa = 2.0f * ((float)(2u * e) + 0.5f) / (float)width - 1.0f;
            const float xb = 2.0f * ((float)(2u * e + 1u) + 0.5f) / (float)width - 1.0f;
            out.ray_colp[2u * e] = make_float4(p[0][0] * xa, p[0][0] * xb, p[0][1] * xa, p[0][1] * xb);
            out.ray_colp[2u * e + 1u] = make_float4(p[0][2] * xa, p[0][2] * xb, xa, xb);
        } else if (e - out.ray_pairs < out.ray_rows) {
            const uint32_t y = e - out.ray_pairs;
            const float ya = 2.0f * ((float)y + 0.5f) / (float)height - 1.0f;
            out.ray_row[y] = make_float4(p[1][0] * ya, p[1][1] * ya, p[1][2] * ya, ya);
        }
        return;
    }
    const uint32_t i = block * 256u + threadIdx.x;
    if (i >= n_tris) return;
    out.tnum[i] = -(dot3(ld3(tris[i].N), ld3(cam.origin)) + tris[i].d);  // compute.wgsl:99-102
    FrameTri T;
    if (cc.enabled) {
        T = make_frame_tri(cc, cull[i]);
    } else {
        const float inf = __builtin_inff();
        T.bx0 = -inf; T.by0 = -inf; T.bx1 = inf; T.by1 = inf;
        T.ea[0] = T.ea[1] = T.ea[2] = inf;
        T.ex[0] = T.ex[1] = T.ex[2] = 0.0f;
        T.ey[0] = T.ey[1] = T.ey[2] = 0.0f;
        T.me0 = T.me1 = T.me2 = 0.0f;
    }
    out.ftris[i] = T;
}

}  // namespace rwr
