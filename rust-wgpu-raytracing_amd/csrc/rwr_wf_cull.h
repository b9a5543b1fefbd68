// Wavefront integrator: which faces a wave's 32x4-pixel tile can see, and whether it can see anything at all.
// Shared by the primary stage (kernels_wf_primary.hip: k_wf_primary culls with it) and by the tile classification that
// precedes it on frames that show little (k_wf_classify), so that both reach the same verdict from the same arithmetic.
#pragma once
#include "rwr_cull.h"
#include "rwr_internal.h"

namespace rwr {

struct WfWaveCull {
    uint32_t n_src;                 // candidate faces of the tile's source (screen bin or whole scene), wave-uniform
    const uint32_t *src;            // the bin's list, or null: faces 0 .. n_src
    bool cached;                    // n_src <= 128: the survivors of the rectangle test are the two ballots below
    unsigned long long cm0, cm1;    // surviving faces among the first / second 64 of the source
    uint32_t cf0, cf1;              // per lane: the face its bit of cm0 / cm1 stands for
};

// blk_x0, blk_y0: the 64x8-pixel block's first pixel (blk_y0 includes row_begin); wave: 0..3 inside the block
template <bool CULL, typename Bins = BinGrid>   // (Bins: BinGrid, possibly in the kernel-argument address space)
RWR_DEV WfWaveCull wf_wave_cull(const FrameTri *__restrict__ ftris, uint32_t n_tris, uint32_t row_begin, uint32_t bins_enabled,
                                int32_t mesh_x0, int32_t mesh_y0, int32_t mesh_x1, int32_t mesh_y1, const Bins &bins,
                                uint32_t blk_x0, uint32_t blk_y0, uint32_t wave, uint32_t lane)
{
    WfWaveCull c;
    const uint32_t tile_x0 = blk_x0 + (wave & 1u) * 32u, tile_y0 = blk_y0 + (wave >> 1) * 4u;
    // source of candidate faces: the whole scene, or this tile's screen bin (shared by all samples of the frame)
    c.n_src = n_tris;
    c.src = nullptr;
    if (CULL && bins_enabled) {
        const uint32_t bin = ((tile_y0 - row_begin) / kBinH) * bins.bins_x + blk_x0 / kBinW;
        const uint32_t off = bins.offsets[bin];
        if (off != kBinNoList) {   // (kBinNoList: this frame's lists did not fit; walk the whole scene)
            c.n_src = bins.counts[bin];
            c.src = bins.lists + off;
        }
    }
    c.n_src = __builtin_amdgcn_readfirstlane(c.n_src);
    if (CULL) {  // the tile lies outside the screen rectangle of the whole mesh
        const int32_t wu = __builtin_amdgcn_readfirstlane((int32_t)wave);
        const int32_t sx0 = (int32_t)blk_x0 + (wu & 1) * 32, sy0 = (int32_t)blk_y0 + (wu >> 1) * 4;
        if (sx0 + 32 < mesh_x0 || sx0 > mesh_x1 || sy0 + 4 < mesh_y0 || sy0 > mesh_y1) c.n_src = 0u;
    }
    const float tx0 = (float)tile_x0, ty0 = (float)tile_y0;
    const TileRect tile_rect = {tx0, ty0, tx0 + 32.0f, ty0 + 4.0f};
    // Jitter keeps a sample inside its pixel, so the tile's candidate faces are the same for every sample: up to 128
    // source faces are culled once (two ballots, the face of each bit in a VGPR); longer lists are re-culled per sample.
    c.cached = c.n_src <= 128u;
    c.cm0 = 0ull; c.cm1 = 0ull;
    c.cf0 = lane; c.cf1 = 64u + lane;
    if (c.cached) {
        bool keep = lane < c.n_src;
        c.cf0 = (keep && c.src) ? c.src[lane] : lane;
        if (CULL && keep) keep = !rect_culls(ftris[c.cf0], tile_rect);
        c.cm0 = __ballot(keep);
        keep = 64u + lane < c.n_src;
        c.cf1 = (keep && c.src) ? c.src[64u + lane] : 64u + lane;
        if (CULL && keep) keep = !rect_culls(ftris[c.cf1], tile_rect);
        c.cm1 = __ballot(keep);
    }
    return c;
}

// A tile no face and no sphere can be seen through (conservative bounds: nothing any jittered ray of its pixels could
// hit) has nothing to trace in any sample: its pixels keep the clear values.  On a frame that shows a small mesh
// that is most tiles.  Wave-uniform.
template <bool CULL, typename P = FrameParams>
RWR_DEV bool wf_wave_empty(const WfWaveCull &c, const P &p, float tx0, float ty0)
{
    bool empty_tile = CULL && (c.n_src == 0u || (c.cached && (c.cm0 | c.cm1) == 0ull));
    if (empty_tile)
        for (uint32_t s = 0; s < p.n_spheres; s++)
            if (!((tx0 + 32.0f < p.sphere_rect[s][0]) || (tx0 > p.sphere_rect[s][2]) || (ty0 + 4.0f < p.sphere_rect[s][1]) ||
                  (ty0 > p.sphere_rect[s][3])))
                empty_tile = false;
    return empty_tile;
}

}  // namespace rwr
