// Primary-ray visibility for one 32x8-pixel workgroup (4 waves of 8x8 pixels, one
// lane per pixel): the reference's sphere passes and mesh pass with its depth
// test (/root/reference/src/lib.rs:1106-1184), composited in registers.
// Shared by the fused frame kernel (kernels_primary.hip) and the first stage of
// the wavefront integrator (kernels_wavefront.hip).
#pragma once

#include "rwr_cull.h"
#include "rwr_device.h"

namespace rwr {

// LDS a workgroup needs for primary_visibility.
struct PrimaryShared {
    uint32_t cand[256];   // block-level candidate faces of the current batch, ascending
    uint32_t wave_cnt[4];
};

// What the pixel shows after all passes.
struct PrimaryHit {
    float depth_tex;  // value of depth_texture_output (0 = cleared)
    int32_t obj;      // >= 0 face, -1 nothing, -2-k sphere k
    float t;          // distance of the winning hit (0 if none)
    MeshHit mesh;     // valid when obj >= 0
};

// Must be called by all 256 threads of the workgroup (contains barriers).
// O, D: this lane's ray.  (tile_x0, tile_y0): the wave's 8x8 tile; blk_x0: the workgroup's 32-wide block.
template <bool CULL, bool COUNT>
RWR_DEV void primary_visibility(const FrameParams &p, const TriRecord *__restrict__ tris,
                                const FrameTri *__restrict__ ftris, PrimaryShared &sh, uint32_t blk_x0,
                                uint32_t tile_x0, uint32_t tile_y0, f3 O, f3 D, PrimaryHit &r, uint32_t &dbg_listed,
                                uint32_t &dbg_tested)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    r.depth_tex = 0.0f;  // depth_texture_* after the clear (lib.rs:1024-1104)
    r.obj = -1;
    r.t = 0.0f;

    // -- analytic sphere passes, in order (lib.rs:1106-1173) -----------------
    const float tx0 = (float)tile_x0, ty0 = (float)tile_y0;
    for (uint32_t s = 0; s < p.n_spheres; s++) {
        // wave-uniform: the tile lies outside the sphere's conservative silhouette bounds
        if (CULL && ((tx0 + 8.0f < p.sphere_rect[s][0]) || (tx0 > p.sphere_rect[s][2]) ||
                     (ty0 + 8.0f < p.sphere_rect[s][1]) || (ty0 > p.sphere_rect[s][3])))
            continue;
        float t;
        if (sphere_ray_intersect_t(ld3(p.spheres[s].center), p.spheres[s].radius, O, D, t)) {
            const float current_depth = 1.0f - r.depth_tex;  // sphere/compute.wgsl:130
            const float depth = to_non_linear_depth(t);
            if (!(depth >= current_depth)) {
                r.depth_tex = 1.0f - depth;
                r.obj = -2 - (int32_t)s;
                r.t = t;
            }
        }
    }

    // -- mesh pass (lib.rs:1174-1184) -----------------------------------------
    MeshHit best;
    best.have = false; best.t = 0.0f; best.u = 0.0f; best.v = 0.0f; best.ndotd = 0.0f; best.idx = 0u;
    if (p.n_tris) {
        const float bx0 = (float)blk_x0;
        const TileRect blk_rect = {bx0, ty0, bx0 + 32.0f, ty0 + 8.0f};
        const TileRect tile_rect = {tx0, ty0, tx0 + 8.0f, ty0 + 8.0f};
        // source of candidate faces: the whole scene, or this workgroup's screen bin
        uint32_t n_src = p.n_tris;
        const uint32_t *__restrict__ src = nullptr;
        if (CULL && p.bins.enabled) {
            const uint32_t bin = ((tile_y0 - p.row_begin) / kBinH) * p.bins.bins_x + blk_x0 / kBinW;
            const uint32_t off = p.bins.offsets[bin];
            if (off != kBinNoList) {   // (kBinNoList: this frame's lists did not fit; walk the whole scene)
                n_src = p.bins.counts[bin];
                src = p.bins.lists + off;
            }
        }
        n_src = __builtin_amdgcn_readfirstlane(n_src);
        // workgroup-uniform: the 32x8 block lies outside the screen rectangle of the whole mesh
        if (CULL && ((bx0 + 32.0f < p.mesh_rect[0]) || (bx0 > p.mesh_rect[2]) || (ty0 + 8.0f < p.mesh_rect[1]) || (ty0 > p.mesh_rect[3])))
            n_src = 0u;
        for (uint32_t base = 0; base < n_src; base += 256u) {
            // level 1: 256 faces vs the block rectangle, order-preserving compaction into LDS
            const uint32_t e0 = base + threadIdx.x;
            bool keep = e0 < n_src;
            const uint32_t j = (keep && src) ? src[e0] : e0;
            if (CULL && keep) keep = !rect_culls(ftris[j], blk_rect);
            const unsigned long long m = __ballot(keep);
            if (lane == 0) sh.wave_cnt[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t off = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) {
                const uint32_t c = sh.wave_cnt[w];
                off += (w < wave) ? c : 0u;
                total += c;
            }
            total = __builtin_amdgcn_readfirstlane(total);
            if (COUNT) dbg_listed += total;
            if (keep) sh.cand[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = j;
            __syncthreads();
            // level 2: list entries vs this wave's tile rectangle (skipped for short lists,
            // where it costs more than the exact tests it saves), then the exact test
            const bool wave_cull = CULL && total > p.wave_cull_min;
            for (uint32_t cbase = 0; cbase < total; cbase += 64u) {
                const uint32_t e = cbase + lane;
                bool keep2 = e < total;
                const uint32_t my_idx = keep2 ? sh.cand[e] : 0u;
                if (wave_cull && keep2) keep2 = !rect_culls(ftris[my_idx], tile_rect);
                unsigned long long m2 = __ballot(keep2);
                while (m2) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(m2);
                    m2 &= m2 - 1ull;
                    // wave-uniform face index: the record comes in through scalar loads
                    const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)my_idx, (int)b);
                    intersect_and_select(tris[idx], idx, O, D, best);
                    if (COUNT) dbg_tested++;
                }
            }
            if (base + 256u < n_src) __syncthreads();  // cand / wave_cnt are rewritten by the next batch
        }
    }
    r.mesh = best;
    if (best.have) {
        const float current_depth = 1.0f - r.depth_tex;  // compute.wgsl:210
        const float depth = to_non_linear_depth_auto(best.t);
        if (!(depth >= current_depth)) {
            r.depth_tex = 1.0f - depth;
            r.obj = (int32_t)best.idx;
            r.t = best.t;
        }
    }
}

// Local shading E of the winning surface (colour path) -> rgb (alpha is 2.0 on a hit), and its albedo.
// (obj, t: the winner and its distance; u, v, ndotd: MeshHit fields of a mesh winner — by value, a
// reference to the hit record keeps part of it in memory.)
// NMAP: the render asked for normal-mapped shading (RWR_FLAG_NORMAL_MAP) — a template parameter so that kernels which
// are instantiated without it keep their registers.
template <bool NMAP = false>
RWR_DEV Shaded shade_winner(const FrameParams &p, int32_t obj, float t, float u, float v, float ndotd,
                            const ShadeRec *__restrict__ shade, const float4 *__restrict__ tex, f3 O, f3 D)
{
    if (obj >= 0) {
        const ShadeRec &S = shade[obj];
        // normal-mapped shading (extension): the face's tangent frame and its material's map
        const TangentRec *G = NMAP ? &p.tangents[obj] : nullptr;
        const MaterialRec *Mn = NMAP ? &p.materials[S.material] : nullptr;
        if (p.n_materials > 1u) {  // wave-uniform: per-face material (extension)
            const MaterialRec &M = p.materials[S.material];
            return shade_mesh(S, u, v, ndotd, D, M.ambient, M.specular, M.tex, M.tex_w * 16u, M.wmax, M.hmax, G, Mn);
        }
        return shade_mesh(S, u, v, ndotd, D, p.ambient, p.specular, tex, p.tex_w * 16u, p.tex_wmax, p.tex_hmax, G, Mn);
    }
    const uint32_t k = (uint32_t)(-2 - obj);
    const f3 P = along(O, t, D);
    const f3 n = cnormalize(sub3(P, ld3(p.spheres[k].center)));
    Shaded s;
    s.albedo = mk3(1.0f, 0.0f, 0.0f);
    s.colour = shade_sphere(n, D);
    return s;
}

}  // namespace rwr
