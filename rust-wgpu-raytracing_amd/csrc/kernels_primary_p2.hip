// Fused frame kernel, two pixels per lane (see rwr_device_p2.h for why): ONE launch replaces the
// reference's three clears, two sphere passes, two depth copies and the mesh pass
// (/root/reference/src/lib.rs:1024-1184).  Same results as kernels_primary.hip's k_primary, which stays
// as the one-pixel-per-lane form (RWR_FLAG_ONE_PIXEL_PER_LANE and the wavefront integrator's first stage).
//   wave64 = 32x4 pixel tile, lane l owns pixels (2*(l&15), l>>4) and (2*(l&15)+1, l>>4);
//   workgroup (4 waves, 2x2 tiles) = 64x8 pixels = one screen-bin column (kBinW).
//   Each wave culls the faces for its own tile, 64 at a time, one per lane, from the per-frame
//   records of k_frame_setup, and runs the exact test on the survivors in ascending face order with
//   the face record in scalar registers.  No LDS, no barrier.
#include <hip/hip_ext.h>

#include "rwr_frame_setup.h"
#include "rwr_primary.h"
#include "rwr_shade_p2.h"

namespace rwr {

// Tile of a wave: 32x4 pixels (default) or 16x8.  With 32x4 every row of a tile is one whole 128-byte line
// of the RGBA8 and the R32F target, which the streaming stores then write without a partial-line pass
// through the L2 (WRITE_SIZE = 8 B/pixel exactly; 16x8 tiles measured 10 % more); the frame time is the same.
#ifndef RWR_P2_TILE_32x4
#define RWR_P2_TILE_32x4 1
#endif
#ifndef RWR_P2_OCC
#define RWR_P2_OCC 7  // 72 VGPRs: the shading step needs 66; at 8 waves (64) it spills and is slower (measured)
#endif
// NMAP: normal-mapped shading (extension, RWR_FLAG_NORMAL_MAP) — its own instantiation, so that the reference's frame keeps
// its registers.
// FUSED: ONE launch per frame (the reference submits a frame once, lib.rs:1226).  The grid's first rows are workgroups that
// make the frame's records and tables — the work of k_frame_setup, rwr_frame_setup.h — and count themselves off in
// FusedSetup::flag; every other workgroup waits for that count before it touches a record (workgroups are dispatched in
// order of their flattened index, so the record makers are running before a waiting one exists; the wait is BOUNDED all the
// same: a wave whose wait runs out flags the frame as incomplete and leaves, it never hangs).  Saves the host one of its two
// launches per frame and the device the boundary between them.
template <bool AUX, bool CULL, bool NMAP, bool FUSED = false>
__global__ void __launch_bounds__(256, (AUX || NMAP) ? 4 : RWR_P2_OCC)
// (the first eleven arguments repeat FrameParams fields: they are what a wave needs first, and the Makefile
// has their 14 dwords preloaded into SGPRs)
k_primary_p2(const FrameTri *__restrict__ ftris, const float4 *__restrict__ ray_colp, const float4 *__restrict__ ray_row,
             uint32_t n_tris, uint32_t row_begin, uint32_t bins_enabled, uint32_t row_pitch,
             int32_t mesh_x0, int32_t mesh_y0, int32_t mesh_x1, int32_t mesh_y1,
             const FrameParams p, const TriRecord *__restrict__ tris, const ShadeRec *__restrict__ shade,
             const float4 *__restrict__ tex, const Targets tg, const FusedSetup fs)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t by = blockIdx.y;
    const_ptr<TriRecord> tris_c = nullptr;
    const_ptr<ShadeRec> shade_c = nullptr;
    const_ptr<float> tnum_c = nullptr;
    if (FUSED) {
        if (blockIdx.y < fs.extra_rows) {   // a record-making workgroup (or a spare one of the last extra row)
            const uint32_t f = blockIdx.y * gridDim.x + blockIdx.x;
            if (f < fs.n_blocks) {
                frame_setup_block(f, fs.n_blocks, fs.cc, p.cam, p.width, p.height, fs.cull, tris, n_tris, fs.nb_tris, fs.out);
                __threadfence();      // the records are visible to the device ...
                __syncthreads();
                if (threadIdx.x == 0u) atomicAdd(fs.flag, 1u);   // ... before the block counts itself off
            }
            return;
        }
        by -= fs.extra_rows;
        // One wave per workgroup watches the count (thousands of waves polling one word swamp its L2 channel: measured, a
        // frame took 125 us), the others wait for it at a barrier.
        __shared__ uint32_t s_ready;
        if (wave == 0u) {
            uint32_t spins = 0;
            bool ok = true;
            while (__hip_atomic_load(fs.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - fs.flag_base < fs.n_blocks) {   // (wave-uniform)
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1u << 15)) { ok = false; break; }   // ~15 ms: never in practice — and never a hang
            }
            if (lane == 0u) {
                s_ready = ok ? 1u : 0u;
                if (!ok) atomicOr(fs.flag + 1, 1u);
            }
        }
        __syncthreads();
        if (s_ready == 0u) return;   // the frame is flagged incomplete (rwr_synchronize / rwr_readback report it)
        // No cache invalidation here on purpose: the kernel's launch invalidated this CU's caches, and no wave reads a line of
        // the records before this point (they are allocations of their own), so no stale line can be resident; what is needed
        // is that the loads below are ISSUED after the count was seen, which the pointers re-made here guarantee.
        // everything the record makers wrote is read through pointers made HERE, behind the wait
        asm volatile("" : "+s"(ftris), "+s"(ray_colp), "+s"(ray_row));
        const float *tn = p.tnum;
        asm volatile("" : "+s"(tn));
        tnum_c = to_const_space(tn);
        tris_c = to_const_space(tris);
        shade_c = to_const_space(shade);
    }
    const uint32_t blk_x0 = blockIdx.x * 64u;
#if RWR_P2_TILE_32x4
    const uint32_t tile_x0 = blk_x0 + (wave & 1u) * 32u;
    const uint32_t tile_y0 = row_begin + by * row_pitch + (wave >> 1) * 4u;
    const uint32_t px0 = tile_x0 + 2u * (lane & 15u), py = tile_y0 + (lane >> 4);
    constexpr float kTileWf = 32.0f, kTileHf = 4.0f;
#else
    const uint32_t tile_x0 = blk_x0 + wave * 16u;
    const uint32_t tile_y0 = p.row_begin + by * row_pitch;
    const uint32_t px0 = tile_x0 + 2u * (lane & 7u), py = tile_y0 + (lane >> 3);
    constexpr float kTileWf = 16.0f, kTileHf = 8.0f;
#endif

    // -- candidate faces of this wave's tile: the first 64 culling records are requested before anything
    // else, so that their latency hides behind the ray generation ------------------------------
    uint32_t n_src = n_tris;
    const uint32_t *__restrict__ src = nullptr;
    if (CULL && bins_enabled) {
        const uint32_t bin = ((tile_y0 - row_begin) / kBinH) * p.bins.bins_x + blk_x0 / kBinW;
        const uint32_t off = p.bins.offsets[bin];
        if (off != kBinNoList) {   // (kBinNoList: this frame's lists did not fit; walk the whole scene)
            n_src = p.bins.counts[bin];
            src = p.bins.lists + off;
        }
    }
    n_src = __builtin_amdgcn_readfirstlane(n_src);
    if (CULL) {  // the tile lies outside the screen rectangle of the whole mesh: scalar integer compares
        const int32_t wu = __builtin_amdgcn_readfirstlane((int32_t)wave);
#if RWR_P2_TILE_32x4
        const int32_t sx0 = (int32_t)blk_x0 + (wu & 1) * 32, sy0 = (int32_t)(row_begin + by * row_pitch) + (wu >> 1) * 4;
        const int32_t sx1 = sx0 + 32, sy1 = sy0 + 4;
#else
        const int32_t sx0 = (int32_t)blk_x0 + wu * 16, sy0 = (int32_t)(row_begin + by * row_pitch);
        const int32_t sx1 = sx0 + 16, sy1 = sy0 + 8;
#endif
        if (sx1 < mesh_x0 || sx0 > mesh_x1 || sy1 < mesh_y0 || sy0 > mesh_y1) n_src = 0u;
    }
    bool valid = lane < n_src;
    uint32_t face = (valid && src) ? src[lane] : lane;
    FrameTri cur;
    if (CULL && valid) cur = ftris[face];

    const f3 O = ld3(p.cam.origin);
    const v3 D = pixel_pair_ray_dir_tab(p.cam, ray_colp, ray_row, px0, py);

    // framebuffer state of the two pixels, as the reference's cleared textures hold it
    f2 depth_tex = splat(0.0f), win_t = splat(0.0f);
    i2 obj = i2{-1, -1};

    // -- analytic sphere passes, in order (lib.rs:1106-1173) -----------------
    const float tx0 = (float)tile_x0, ty0 = (float)tile_y0;
    for (uint32_t s = 0; s < p.n_spheres; s++) {
        if (CULL && ((tx0 + kTileWf < p.sphere_rect[s][0]) || (tx0 > p.sphere_rect[s][2]) ||
                     (ty0 + kTileHf < p.sphere_rect[s][1]) || (ty0 > p.sphere_rect[s][3])))
            continue;
        f2 t = splat(0.0f);
        const i2 hit = sphere_ray_intersect_t(ld3(p.spheres[s].center), p.spheres[s].radius, O, D, t);
        if (any2(hit)) {
            const f2 current_depth = 1.0f - depth_tex;  // sphere/compute.wgsl:130
            const f2 depth = to_non_linear_depth(t);
            const i2 win = hit & ~(depth >= current_depth);
            depth_tex = win ? (1.0f - depth) : depth_tex;
            obj = win ? i2{-2 - (int)s, -2 - (int)s} : obj;
            win_t = win ? t : win_t;
        }
    }

    // -- mesh pass (lib.rs:1174-1184) -----------------------------------------
    MeshHit2 best;
    best.have = i2{0, 0};
    best.t = best.u = best.v = best.ndotd = splat(0.0f);
    best.idx = u2{0u, 0u};
    uint32_t dbg_listed = 0, dbg_tested = 0;
    uint32_t n_tested = 0;  // wave-uniform
    ShadeRec last_shade = {};  // of the face tested last (scalar registers)
    {
        // Each wave culls for its own tile, 64 faces at a time, one per lane (rwr_cull.h), and walks
        // the survivors in ascending face order: no LDS, no barrier.  The next 64 records are requested
        // before the exact tests of the current ones.
        const TileRect tile_rect = {tx0, ty0, tx0 + kTileWf, ty0 + kTileHf};
        for (uint32_t base = 0; base < n_src; base += 64u) {
            bool keep = valid;
            if (CULL && keep) keep = !rect_culls(cur, tile_rect);
            unsigned long long m = __ballot(keep);
            const uint32_t my_face = face;
            const uint32_t e = base + 64u + lane;
            valid = e < n_src;
            face = (valid && src) ? src[e] : e;
            if (CULL && valid) cur = ftris[face];
            if (AUX) dbg_listed += (uint32_t)__popcll(m);
            while (m) {
                const uint32_t b = (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                // wave-uniform face index: the record comes in through scalar loads
                const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)my_face, (int)b);
                if (FUSED) {
                    intersect_and_select(load_tri_record(tris_c + idx), tnum_c[idx], idx, O, D, best);
                    last_shade = load_shade_record(shade_c + idx);
                } else {
                    intersect_and_select(tris[idx], p.tnum[idx], idx, O, D, best);
                    last_shade = shade[idx];
                }
                n_tested++;
                if (AUX) dbg_tested++;
            }
        }
    }
    if (any2(best.have)) {
        const f2 current_depth = 1.0f - depth_tex;  // compute.wgsl:210
        // same bits either way; the short form covers every distance a scene produces (rwr_device.h)
        const bool fast = !__any(any2(best.have & ~depth_fast_domain(best.t)));
        const f2 depth = fast ? to_non_linear_depth_fast(best.t) : to_non_linear_depth(best.t);
        const i2 win = best.have & ~(depth >= current_depth);
        depth_tex = win ? (1.0f - depth) : depth_tex;
        obj = win ? i2{(int)best.idx.x, (int)best.idx.y} : obj;
        win_t = win ? best.t : win_t;
    }

    // depth plane first (nothing below needs it): 8-byte aligned pair whenever the width is even
    const bool in_frame = py < p.row_end && px0 < p.width;
    const uint32_t o = py * p.width + px0;  // < 2^30 pixels (rwr_resize)
    const bool both = px0 + 1u < p.width;
    const bool pair_store = both && (o & 1u) == 0u;
    if (in_frame) {
        if (pair_store) {
            // streaming stores: the targets are written once and never read here; they must not push the
            // texture and the face records out of the L2 (3.7 % of the frame rate)
            __builtin_nontemporal_store(depth_tex, reinterpret_cast<f2 *>(tg.depth + o));
        } else {
            tg.depth[o] = depth_tex.x;
            if (both) tg.depth[o + 1] = depth_tex.y;
        }
    }

    // -- shade the winners and store --------------------------------------------
    // Sphere winners first (few tiles; one pixel at a time), straight to their packed form, so that the
    // ray direction is dead before the mesh shading starts.  Untouched pixels keep the clear value.
    uint32_t rgba[2] = {0u, 0u};
    float4 cf[2];
    if (AUX) cf[0] = cf[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (any2(obj < -1)) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int o = k ? obj.y : obj.x;
            if (o < -1) {
                const f3 Dk = lane3(D, k);
                const f3 center = ld3(p.spheres[-2 - o].center);
                float t = 0.0f;  // the winner's distance again (same function, same bits) rather than a live register pair
                (void)sphere_ray_intersect_t(center, p.spheres[-2 - o].radius, O, Dk, t);
                const f3 c = shade_sphere(cnormalize(sub3(along(O, t, Dk), center)), Dk);
                rgba[k] = pack_rgba8(c.x, c.y, c.z, 2.0f);
                if (AUX) cf[k] = make_float4(c.x, c.y, c.z, 2.0f);
            }
        }
    }
    if (__any(any2(obj >= 0))) {  // wave-uniform; lanes without a mesh winner shade face 0 and drop the result
        f2 cr, cg, cb;
        if (NMAP) shade_mesh_pair<true, false, true>(p, shade, tex, obj, last_shade, best, D, cr, cg, cb);   // (per-face material path)
        else if (p.n_materials > 1u) shade_mesh_pair<true, false>(p, shade, tex, obj, last_shade, best, D, cr, cg, cb);
        else if (n_tested == 1u) shade_mesh_pair<false, true>(p, shade, tex, obj, last_shade, best, D, cr, cg, cb);
        else shade_mesh_pair<false, false>(p, shade, tex, obj, last_shade, best, D, cr, cg, cb);
        // rgba8unorm conversion of both pixels (rwr_device.h); alpha 2.0 -> 255
        const f2 sr = cr * 255.0f, sg = cg * 255.0f, sb = cb * 255.0f;
        const uint32_t m0 = pack_rgba8_scaled(sr.x, sg.x, sb.x, 0xff000000u);
        const uint32_t m1 = pack_rgba8_scaled(sr.y, sg.y, sb.y, 0xff000000u);
        rgba[0] = obj.x >= 0 ? m0 : rgba[0];
        rgba[1] = obj.y >= 0 ? m1 : rgba[1];
        if (AUX) {
            if (obj.x >= 0) cf[0] = make_float4(cr.x, cg.x, cb.x, 2.0f);
            if (obj.y >= 0) cf[1] = make_float4(cr.y, cg.y, cb.y, 2.0f);
        }
    }

    if (in_frame) {
        if (pair_store) {
            __builtin_nontemporal_store(u2{rgba[0], rgba[1]}, reinterpret_cast<u2 *>(reinterpret_cast<uint32_t *>(tg.color) + o));
        } else {
            reinterpret_cast<uint32_t *>(tg.color)[o] = rgba[0];
            if (both) reinterpret_cast<uint32_t *>(tg.color)[o + 1] = rgba[1];
        }
        if (AUX) {
            const bool dbg = (p.flags & RWR_FLAG_DEBUG_COUNTS) != 0;
            reinterpret_cast<float4 *>(tg.color_f32)[o] = cf[0];
            tg.obj_id[o] = dbg ? (int32_t)dbg_listed : obj.x;
            tg.hit_t[o] = dbg ? (float)dbg_tested : win_t.x;
            if (both) {
                reinterpret_cast<float4 *>(tg.color_f32)[o + 1] = cf[1];
                tg.obj_id[o + 1] = dbg ? (int32_t)dbg_listed : obj.y;
                tg.hit_t[o + 1] = dbg ? (float)dbg_tested : win_t.y;
            }
        }
    }
}

uint32_t primary_p2_fused_rows(const FrameParams &fp, uint32_t n_blocks)
{
    const uint32_t gx = (fp.width + 63u) / 64u;
    return gx ? (n_blocks + gx - 1u) / gx : 0u;
}

hipError_t launch_primary_p2(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                             const FrameTri *ftris, const float4 *tex, const Targets &tg, hipEvent_t ev_start,
                             hipEvent_t ev_stop, const FusedSetup *fused)
{
    if (fp.row_end <= fp.row_begin || fp.width == 0) return hipSuccess;
    const dim3 grid((fp.width + 63u) / 64u, band_strips(fp) + (fused ? fused->extra_rows : 0u));
    const dim3 block(256);
    const bool aux = (fp.flags & RWR_FLAG_AUX_OUTPUTS) != 0;
    const bool do_cull = (fp.flags & RWR_FLAG_NO_CULL) == 0;
    // ev_start / ev_stop (may be null): timestamps of this dispatch itself (hipExtLaunchKernelGGL), i.e. the
    // kernel's own duration as a profiler reports it, without the gap to the preceding kernel
    const bool nmap = (fp.flags & RWR_FLAG_NORMAL_MAP) != 0 && fp.tangents != nullptr;
    const FusedSetup fs = fused ? *fused : FusedSetup{};
#define RWR_P2_LAUNCH(A, C, N, F) hipExtLaunchKernelGGL((k_primary_p2<A, C, N, F>), grid, block, 0, s, ev_start, ev_stop, 0, ftris, fp.ray_colp, fp.ray_row, \
    fp.n_tris, fp.row_begin, fp.bins.enabled, fp.row_pitch, fp.mesh_px[0], fp.mesh_px[1], fp.mesh_px[2], fp.mesh_px[3], fp, tris, shade, tex, tg, fs)
    if (fused) {   // (the plain reference frame only: context.cpp)
        RWR_P2_LAUNCH(false, true, false, true);
    } else if (nmap) {
        if (aux && do_cull) RWR_P2_LAUNCH(true, true, true, false);
        else if (aux) RWR_P2_LAUNCH(true, false, true, false);
        else if (do_cull) RWR_P2_LAUNCH(false, true, true, false);
        else RWR_P2_LAUNCH(false, false, true, false);
    } else {
        if (aux && do_cull) RWR_P2_LAUNCH(true, true, false, false);
        else if (aux) RWR_P2_LAUNCH(true, false, false, false);
        else if (do_cull) RWR_P2_LAUNCH(false, true, false, false);
        else RWR_P2_LAUNCH(false, false, false, false);
    }
#undef RWR_P2_LAUNCH
    return hipGetLastError();
}

hipError_t preload_kernels_primary_p2()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>((&k_primary_p2<false, true, false>)));
}

}  // namespace rwr
