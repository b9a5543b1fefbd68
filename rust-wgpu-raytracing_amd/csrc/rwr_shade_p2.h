// Pair-wide mesh shading shared by the two-pixels-per-lane kernels (the fused frame kernel,
// kernels_primary_p2.hip, and the first stage of the wavefront integrator, kernels_wf_primary.hip).
#pragma once

#include "rwr_device_p2.h"

namespace rwr {

// Mesh shading (triangle_list/compute.wgsl:217-234; colour path, tolerance 1e-4, not bit-exact) of the
// winners of both pixels of a lane: per-pixel record loads, dot products and texture taps
// (rwr_device.h), then the pair-wide steps — half vector, x^32, the final multiply-adds — as
// packed instructions.  obj < 0 (no mesh winner) shades face 0; the caller drops that result.
// UNIFORM: the wave ran the exact test on exactly one face (most tiles), so every mesh winner is that
// face; its record `uS` was fetched by scalar loads together with the TriRecord, which takes one
// dependent memory round trip (and six vector loads) out of the tile's critical path — at 1080p the
// frame kernel spends as long on the latency chain of its first and last waves as on arithmetic.
// (tr, tg, tb): the filtered diffuse texel of each pixel — the surface's albedo, which the wavefront integrator
// carries along the bounce ray.
// NMAP: normal-mapped light terms where the face's material has a map (extension, RWR_FLAG_NORMAL_MAP).
// P: FrameParams, or FrameParams in the kernel-argument (constant) address space — a caller inside a long loop reads the fields
// through a pointer it has just re-derived, so that they are loaded where they are used instead of living in scalar registers
// for the whole loop (kernels_wf_primary.hip).
template <bool MULTI, bool UNIFORM, bool NMAP = false, typename P = FrameParams>
RWR_DEV void shade_mesh_pair(const P &p, const ShadeRec *__restrict__ shade, const float4 *__restrict__ tex,
                             i2 obj, const ShadeRec &uS, const MeshHit2 &best, v3 D, f2 &cr, f2 &cg, f2 &cb,
                             f2 &tr, f2 &tg, f2 &tb)
{
    static_assert(!(NMAP && UNIFORM), "the normal-mapped path fetches per-pixel records");
    const v3 h = sub3(splat3(mesh_light_dir()), D);          // :229, un-normalised
    const f2 hh = fma2(h.z, h.z, fma2(h.y, h.y, h.x * h.x));
    const f2 rh = f2{__builtin_amdgcn_rsqf(hh.x), __builtin_amdgcn_rsqf(hh.y)};
    f2 ndl, hn;
    f2 kar = splat(p.ambient[0]), kag = splat(p.ambient[1]), kab = splat(p.ambient[2]);
    f2 ksr = splat(p.specular[0]), ksg = splat(p.specular[1]), ksb = splat(p.specular[2]);
    TexTaps taps[2];
    const float4 *texk[2] = {tex, tex};
    // phase 1, both pixels: record loads, light terms, tap addresses
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const ShadeRec &S = UNIFORM ? uS : shade[(uint32_t)max(k ? obj.y : obj.x, 0)];
        HalfVec hv;
        hv.h = lane3(h, k);
        hv.rh = k ? rh.y : rh.x;
        float a, c;
        mesh_light_terms(S, k ? best.ndotd.y : best.ndotd.x, hv, a, c);
        if (k) { ndl.y = a; hn.y = c; } else { ndl.x = a; hn.x = c; }
        const f2 pos = mesh_texel_pos(S, k ? best.u.y : best.u.x, k ? best.v.y : best.v.x);
        if (NMAP) {
            const MaterialRec &Mn = p.materials[S.material];
            if (Mn.nmap) {
                const f3 cn = tex_sample_bilinear(Mn.nmap, Mn.nmap_w * 16u, (float)(Mn.nmap_w - 1u), (float)(Mn.nmap_h - 1u), nmap_texel_pos(Mn, pos));
                mesh_light_terms_nm(S, p.tangents[(uint32_t)max(k ? obj.y : obj.x, 0)], cn, k ? best.ndotd.y : best.ndotd.x, hv, a, c);
                if (k) { ndl.y = a; hn.y = c; } else { ndl.x = a; hn.x = c; }
            }
        }
        if (MULTI) {  // per-face material (extension)
            const MaterialRec &M = p.materials[S.material];
            taps[k] = tex_taps(M.tex_w * 16u, M.wmax, M.hmax, pos);
            texk[k] = M.tex;
            if (k) { kar.y = M.ambient[0]; kag.y = M.ambient[1]; kab.y = M.ambient[2]; ksr.y = M.specular[0]; ksg.y = M.specular[1]; ksb.y = M.specular[2]; }
            else { kar.x = M.ambient[0]; kag.x = M.ambient[1]; kab.x = M.ambient[2]; ksr.x = M.specular[0]; ksg.x = M.specular[1]; ksb.x = M.specular[2]; }
        } else {
            taps[k] = tex_taps(p.tex_w * 16u, p.tex_wmax, p.tex_hmax, pos);
        }
    }
    // phase 2, one pixel after the other (the scheduling barriers keep 12, not 24, texel registers live)
    __builtin_amdgcn_sched_barrier(0);
    {
        const f3 t = tex_filter(texk[0], taps[0]);
        tr.x = t.x; tg.x = t.y; tb.x = t.z;
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        const f3 t = tex_filter(texk[1], taps[1]);
        tr.y = t.x; tg.y = t.y; tb.y = t.z;
    }
    f2 sp = hn * hn;  // pow(., 32) by five squarings (rwr_device.h pow32)
    sp = sp * sp; sp = sp * sp; sp = sp * sp; sp = sp * sp;
    cr = fma2(ksr, sp, fma2(tr, ndl, kar));                  // :231-233
    cg = fma2(ksg, sp, fma2(tg, ndl, kag));
    cb = fma2(ksb, sp, fma2(tb, ndl, kab));
}

template <bool MULTI, bool UNIFORM, bool NMAP = false, typename P = FrameParams>
RWR_DEV void shade_mesh_pair(const P &p, const ShadeRec *__restrict__ shade, const float4 *__restrict__ tex,
                             i2 obj, const ShadeRec &uS, const MeshHit2 &best, v3 D, f2 &cr, f2 &cg, f2 &cb)
{
    f2 tr, tg, tb;
    shade_mesh_pair<MULTI, UNIFORM, NMAP>(p, shade, tex, obj, uS, best, D, cr, cg, cb, tr, tg, tb);
}

}  // namespace rwr
