// Wavefront integrator, first stage (extension; BASELINE.json configs 2-4 — the reference traces one centre
// ray per pixel and has no samples or bounces; specification: DESIGN.md "Extended integrator",
// restated brute force in oracle/rt_oracle.c or_render_path).
//
//   k_wf_primary   ONE launch traces a whole GROUP of samples (up to 32) of every pixel.  Same mapping as the
//                  fused frame kernel (kernels_primary_p2.hip): wave64 = 32x4-pixel tile, two horizontally
//                  adjacent pixels per lane (v_pk_* arithmetic), per-wave culling against the per-frame
//                  FrameTri records, face records in scalar registers, no LDS, no barrier.  Per sample: jittered
//                  ray (counter-based RNG keyed by GLOBAL pixel, sample, seed), spheres + mesh with the
//                  reference's depth compositing, local shading E(h0); the group's E(h0) are summed in
//                  registers as fixed point and added to the frame's integer planes ONCE per group (not per
//                  sample).  A pixel that hit a surface builds
//                  its cosine-distributed bounce ray and stores it at its FIXED slot of the ray queue in HBM
//                  (coalesced 16-byte stores, rwr_internal.h WfBuffers); the wave publishes the ballot of the
//                  lanes that emitted.  Compaction — ballot + prefix popcount — happens in the consumer
//                  (kernels_wf_bounce.hip), folded into the sort of the tile's ray pool by direction.
//                  Rays that left the scene cost nothing downstream.
//   k_wf_classify  frames that show little: lists the tiles anything can be seen through, so that k_wf_primary<LIST>
//                  takes (listed tile, share of its samples) items instead of grid coordinates (see there).
//
// Everything that decides what a sample sees (ray, hit tests, selection, depth, bounce ray) is written operation
// for operation like the oracle, so sample-0 planes are bit-exact and every bounce ray is the oracle's.
#include <hip/hip_ext.h>

#include <algorithm>

#include "rwr_primary.h"
#include "rwr_shade_p2.h"
#include "rwr_wf_cull.h"

namespace rwr {

// rng_hash (rwr_device.h) in two steps: everything up to the dimension is shared by the uniforms of one
// (pixel, sample).  rng_dim(rng_base(pixel, sample, seed), dim) == rng_hash(pixel, sample, dim, seed).
RWR_DEV uint32_t rng_base(uint32_t pixel, uint32_t sample, uint32_t seed)
{
    uint32_t h = seed ^ 0x9E3779B9u;
    h = rng_mix(h ^ pixel);
    return rng_mix(h ^ (sample * 0x85EBCA6Bu));
}
RWR_DEV float rng_dim(uint32_t base, uint32_t dim)
{
    return (float)(rng_mix(base ^ (dim * 0xC2B2AE35u)) >> 8) * (1.0f / 16777216.0f);
}

// n / d for a wave-uniform positive integer-valued d (the frame's width or height) and 0 <= n < 2^16: the
// compiler's IEEE expansion of the quotient with the reciprocal refined once per wave and the instructions that
// are the identity on such operands left out (rwr_device.h div_shared_rcp; zero numerators give +0 either way).
RWR_DEV float refined_rcp(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}

// pixelToRay (compute.wgsl:150-164) for the pixel-space points (fx.x, fy.x) and (fx.y, fy.y).
template <typename Cam>   // rwr_camera_inv_uniform, possibly in the kernel-argument address space
RWR_DEV v3 pixel_pair_ray_dir_at(const Cam &cam, f2 fx, f2 fy, float width, float height, float rw, float rh)
{
    const f2 x_nds = div_shared_rcp(2.0f * fx, splat(width), splat(rw)) - 1.0f;
    const f2 y_nds = div_shared_rcp(2.0f * fy, splat(height), splat(rh)) - 1.0f;
    const auto &p = cam.proj_inv;
    const f2 vx = p[0][0] * x_nds + p[1][0] * y_nds + p[2][0] * 1.0f + p[3][0] * 1.0f;
    const f2 vy = p[0][1] * x_nds + p[1][1] * y_nds + p[2][1] * 1.0f + p[3][1] * 1.0f;
    const f2 vz = p[0][2] * x_nds + p[1][2] * y_nds + p[2][2] * 1.0f + p[3][2] * 1.0f;
    const f2 vw = splat(0.0f);
    const auto &m = cam.viewmodel_inv;
    v3 w;
    w.x = m[0][0] * vx + m[1][0] * vy + m[2][0] * vz + m[3][0] * vw;
    w.y = m[0][1] * vx + m[1][1] * vy + m[2][1] * vz + m[3][1] * vw;
    w.z = m[0][2] * vx + m[1][2] * vy + m[2][2] * vz + m[3][2] * vw;
    if (__all(normalize_fast_domain(w))) return normalize3_fast(w);
    return normalize3(w);
}

// bounce_direction (rwr_device.h) for a pixel pair; lanes / elements whose `want` is 0 still compute (cheaply
// wrong values are never stored).  base: rng_base of each pixel for this sample.
RWR_DEV v3 bounce_direction_pair(v3 n, u2 base, i2 want)
{
    // Rejection-sample the unit disk (up to 8 tries per pixel, a = b = 0 when all fail).  78.5 % of the tries
    // succeed: the first try runs for both pixels at once; for the rest a lane takes ONE try of ONE pixel per
    // iteration (first its left pixel's, then its right pixel's), so a wave iterates max over lanes of the tries
    // both pixels still need, not 2 x max over pixels.
    f2 a = splat(0.0f), b = splat(0.0f);
    i2 need = want;
    {
        const f2 ua = 2.0f * f2{rng_dim(base.x, 2u), rng_dim(base.y, 2u)} - 1.0f;
        const f2 ub = 2.0f * f2{rng_dim(base.x, 3u), rng_dim(base.y, 3u)} - 1.0f;
        const i2 ok = need & ((ua * ua + ub * ub) <= 1.0f);
        a = ok ? ua : a;
        b = ok ? ub : b;
        need &= ~ok;
    }
    bool busy = any2(need);
    uint32_t sel = need.x ? 0u : 1u, k = 1u;
    while (__any(busy)) {
        if (busy) {
            const uint32_t bs = sel ? base.y : base.x;
            const float ua = 2.0f * rng_dim(bs, 2u + 2u * k) - 1.0f;
            const float ub = 2.0f * rng_dim(bs, 3u + 2u * k) - 1.0f;
            const bool ok = (ua * ua + ub * ub) <= 1.0f;
            if (ok) {
                if (sel) { a.y = ua; b.y = ub; } else { a.x = ua; b.x = ub; }
            }
            if (ok || k == 7u) {          // this pixel is done: on to the lane's other pixel, or out
                if (sel == 0u && need.y) { sel = 1u; k = 1u; }
                else busy = false;
            } else {
                k++;
            }
        }
    }
    const f2 rad = 1.0f - a * a - b * b;
    const f2 clamped = f2{fmaxf(0.0f, rad.x), fmaxf(0.0f, rad.y)};
    // sqrt_fast returns sqrtf's bits on [2^-100, 2^100] (rwr_device.h); 0 and the rare tiny radicand take the long form
    const bool sq_fast = __all(clamped.x >= 0x1p-100f && clamped.y >= 0x1p-100f);
    const f2 dz = sq_fast ? sqrt_fast(clamped) : sqrt2(clamped);
    const f2 sign = f2{copysignf(1.0f, n.z.x), copysignf(1.0f, n.z.y)};
    const f2 aa = -1.0f / (sign + n.z);
    const f2 bb = n.x * n.y * aa;
    const v3 b1 = v3{1.0f + sign * n.x * n.x * aa, sign * bb, -sign * n.x};
    const v3 b2 = v3{bb, sign + n.y * n.y * aa, -n.y};
    v3 d = v3{a * b1.x + b * b2.x + dz * n.x, a * b1.y + b * b2.y + dz * n.y, a * b1.z + b * b2.z + dz * n.z};
    // elements nobody wants get a harmless in-domain vector, so that they do not push the wave onto the long form
    d.x = want ? d.x : splat(1.0f); d.y = want ? d.y : splat(1.0f); d.z = want ? d.z : splat(1.0f);
    if (__all(normalize_fast_domain(d))) return normalize3_fast(d);
    return normalize3(d);
}

#ifndef RWR_WF_OCC
#define RWR_WF_OCC 4
#endif
// The kernel's ONE argument.  The sample loop is long and uses some 130 wave-uniform words — camera matrices, spheres and their
// rectangles, a dozen pointers, the face record under test — against 102 scalar registers; the compiler hoists every
// kernel-argument load out of the loop and then spills (round 2: 86-135 scalar spills through v_writelane / v_readlane, 11-26
// vector registers in scratch).  So the loop reads what a phase needs through a pointer to the kernel-argument segment that is
// RE-DERIVED at the phase (wf_args_again: an empty asm the compiler cannot see through), i.e. by scalar loads from the constant
// cache where the values are used; they are dead again before the next phase.
struct WfPrimaryArgs {
    const FrameTri *ftris;
    uint32_t n_tris, row_begin, bins_enabled;
    int32_t mesh_x0, mesh_y0, mesh_x1, mesh_y1;
    uint32_t sample_begin, sample_count, z_split;
    FrameParams p;
    const TriRecord *tris;
    const ShadeRec *shade;
    const float4 *tex;
    Targets tg;
    WfBuffers wf;
};
template <typename T> using kernarg = const __attribute__((address_space(4))) T;
RWR_DEV kernarg<WfPrimaryArgs> *wf_args_again(kernarg<WfPrimaryArgs> *q)
{
    asm volatile("" : "+s"(q));
    return q;
}
RWR_DEV f3 ld3(kernarg<float> *q) { return mk3(q[0], q[1], q[2]); }

template <bool AUX, bool CULL, bool NMAP, bool LIST = false>
// (LIST — frames that show little — keeps an item loop's state on top of everything else and needs 142 registers: three waves
// per SIMD like the AUX / NMAP forms; such a frame does not fill the chip anyway)
__global__ void __launch_bounds__(256, (AUX || NMAP || LIST) ? 3 : RWR_WF_OCC)
k_wf_primary(const WfPrimaryArgs a)
{
    kernarg<WfPrimaryArgs> *const ka = (kernarg<WfPrimaryArgs> *)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t sample_begin = a.sample_begin, sample_count = a.sample_count, z_split = a.z_split;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // the bounce stage's counters for this queue start from zero (nothing reads them before this kernel has ended)
    if (a.wf.counters && (blockIdx.x | blockIdx.y | blockIdx.z) == 0u && threadIdx.x < 4u) a.wf.counters[threadIdx.x] = 0u;
    // which tile, which share of its samples: the launch grid's (x, y, z) — or, on a frame that shows little, item after item
    // of (live tile of k_wf_classify's list) x (share), so that no workgroup is spent on finding its tile empty
    uint32_t bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
    uint32_t item = blockIdx.x, n_items = 0u;
    if (LIST) n_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)*a.wf.live_count) * z_split;
    do {
    kernarg<WfPrimaryArgs> *const qi = wf_args_again(ka);   // this tile's set-up: culling inputs, the frame's size and band
    const uint32_t tiles_x = qi->wf.tiles_x;
    if (LIST) {
        if (item >= n_items) break;
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)qi->wf.live_list[item / z_split]);
        z = item % z_split;
        by = t / tiles_x;
        bx = t - by * tiles_x;
        item += gridDim.x;
    }
    const uint32_t row_begin = qi->row_begin, width = qi->p.width;
    const uint32_t blk_x0 = bx * kWfTileW;
    const uint32_t tile_x0 = blk_x0 + (wave & 1u) * 32u;
    const uint32_t tile_y0 = row_begin + by * qi->p.row_pitch + (wave >> 1) * 4u;
    const uint32_t px0 = tile_x0 + 2u * (lane & 15u), py = tile_y0 + (lane >> 4);
    constexpr float kTileWf = 32.0f, kTileHf = 4.0f;
    const bool in0 = py < qi->p.row_end && px0 < width, in1 = in0 && (px0 + 1u < width);
    const uint32_t pix0 = py * width + px0;  // GLOBAL pixel index: RNG key and accumulator slot (< 2^30, rwr_resize)
    const uint32_t tile = by * tiles_x + bx;

    // candidate faces of this wave's tile: rwr_wf_cull.h (shared with k_wf_classify)
    const FrameTri *__restrict__ ftris = qi->ftris;
    const WfWaveCull wc = wf_wave_cull<CULL>(ftris, qi->n_tris, row_begin, qi->bins_enabled, qi->mesh_x0, qi->mesh_y0, qi->mesh_x1, qi->mesh_y1,
                                             qi->p.bins, blk_x0, row_begin + by * qi->p.row_pitch, wave, lane);
    const uint32_t n_src = wc.n_src;
    const uint32_t *__restrict__ src = wc.src;
    const float tx0 = (float)tile_x0, ty0 = (float)tile_y0;
    const TileRect tile_rect = {tx0, ty0, tx0 + kTileWf, ty0 + kTileHf};
    const bool cached = wc.cached;
    const unsigned long long cm0 = wc.cm0, cm1 = wc.cm1;
    const uint32_t cf0 = wc.cf0, cf1 = wc.cf1;

    // the group's sums of E(h0) per pixel as 2^-22 fixed point in 32 bits (a sample's term is clamped to kWfE0Cap = 16 — part of
    // the integrator's definition, rwr_hip.h / oracle render_path_core — so a whole group of <= 64 fits; shifted into the planes'
    // 2^-26 units at the end: WfBuffers::fix), and its primary hits
    uint32_t fr0 = 0, fg0 = 0, fb0 = 0, fr1 = 0, fg1 = 0, fb1 = 0;
    uint32_t hits0 = 0, hits1 = 0;
    uint32_t emitted = 0;  // rays this wave emitted in this launch (wave-uniform)
    // z_split workgroups share a tile's samples (sample z, z + z_split, ...): on a frame that shows a small mesh a wave
    // would otherwise trace all the group's samples one after the other while most of the chip idles

    // A tile no face and no sphere can be seen through (conservative bounds: nothing any jittered ray of its pixels could
    // hit) has nothing to trace in any sample: its pixels keep the clear values.  On a frame that shows a small mesh
    // that is most tiles.
    const bool empty_tile = wf_wave_empty<CULL>(wc, qi->p, tx0, ty0);
    if (empty_tile && sample_begin == 0u && z == 0u && in0) {   // sample 0's planes
        qi->tg.depth[pix0] = 0.0f;
        if (AUX) { qi->tg.obj_id[pix0] = -1; qi->tg.hit_t[pix0] = 0.0f; }
        if (in1) {
            qi->tg.depth[pix0 + 1u] = 0.0f;
            if (AUX) { qi->tg.obj_id[pix0 + 1u] = -1; qi->tg.hit_t[pix0 + 1u] = 0.0f; }
        }
    }
    if (empty_tile && qi->p.bounces != 0u && lane == 0u && z == 0u)    // nothing emitted: the bounce stage sees empty ballots
        for (uint32_t sidx = 0; sidx < sample_count; sidx++) {
            unsigned long long *mk = qi->wf.masks + (size_t)(tile * qi->wf.group + sidx) * 8u + wave * 2u;
            mk[0] = 0ull; mk[1] = 0ull;
        }

    for (uint32_t sidx = z; sidx < (empty_tile ? 0u : sample_count); sidx += z_split) {
        const uint32_t sample = sample_begin + sidx;
        // -- the sample's ray: pixel centre at spp = 1, else two uniforms of the counter-based RNG ---------------
        kernarg<WfPrimaryArgs> *const qr = wf_args_again(ka);   // this phase's kernel arguments: seed, spp, the camera
        const u2 base = u2{rng_base(pix0, sample, qr->p.seed), rng_base(pix0 + 1u, sample, qr->p.seed)};
        f2 jx = splat(0.5f), jy = splat(0.5f);
        if (qr->p.spp > 1u) {
            jx = f2{rng_dim(base.x, 0u), rng_dim(base.y, 0u)};
            jy = f2{rng_dim(base.x, 1u), rng_dim(base.y, 1u)};
        }
        const f2 fx = f2{(float)px0, (float)(px0 + 1u)} + jx;
        const f2 fy = splat((float)py) + jy;
        const float fw = (float)qr->p.width, fh = (float)qr->p.height;   // (8 instructions a sample against 4-8 registers across the loop)
        const v3 D = pixel_pair_ray_dir_at(qr->p.cam, fx, fy, fw, fh, refined_rcp(fw), refined_rcp(fh));
        const f3 O = ld3(qr->p.cam.origin);

        f2 depth_tex = splat(0.0f), win_t = splat(0.0f);
        i2 obj = i2{-1, -1};
        // -- analytic sphere passes, in order (lib.rs:1106-1173) ------------------------------------------------
        kernarg<WfPrimaryArgs> *const qs = wf_args_again(ka);   // the spheres and their screen rectangles
        const uint32_t n_spheres = qs->p.n_spheres;
        for (uint32_t s = 0; s < n_spheres; s++) {
            if (CULL && ((tx0 + kTileWf < qs->p.sphere_rect[s][0]) || (tx0 > qs->p.sphere_rect[s][2]) ||
                         (ty0 + kTileHf < qs->p.sphere_rect[s][1]) || (ty0 > qs->p.sphere_rect[s][3])))
                continue;
            f2 t = splat(0.0f);
            const i2 hit = sphere_ray_intersect_t(ld3(qs->p.spheres[s].center), qs->p.spheres[s].radius, O, D, t);
            if (any2(hit)) {
                const f2 current_depth = 1.0f - depth_tex;  // sphere/compute.wgsl:130
                const f2 depth = to_non_linear_depth(t);
                const i2 win = hit & ~(depth >= current_depth);
                depth_tex = win ? (1.0f - depth) : depth_tex;
                obj = win ? i2{-2 - (int)s, -2 - (int)s} : obj;
                win_t = win ? t : win_t;
            }
        }
        // -- mesh pass (lib.rs:1174-1184): per-wave culling, survivors in ascending face order ------------------
        MeshHit2 best;
        best.have = i2{0, 0};
        best.t = best.u = best.v = best.ndotd = splat(0.0f);
        best.idx = u2{0u, 0u};
        uint32_t n_tested = 0;
        ShadeRec last_shade = {};
        // survivors in ascending face order; wave-uniform face index: the record comes in through scalar loads.
        // Samples jitter the ray, never the origin: the plane numerator is the frame's per-face table.
        kernarg<WfPrimaryArgs> *const qm = wf_args_again(ka);   // the face records, the shading records, the numerators
        const const_ptr<TriRecord> tris_c = to_const_space(qm->tris);
        const const_ptr<ShadeRec> shade_c = to_const_space(qm->shade);
        const const_ptr<float> tnum_c = to_const_space(qm->p.tnum);
        auto walk = [&](unsigned long long m, uint32_t faces_v) {
            while (m) {
                const uint32_t b = (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)faces_v, (int)b);
                intersect_and_select(load_tri_record(tris_c + idx), tnum_c[idx], idx, O, D, best);
                n_tested++;
                last_shade = load_shade_record(shade_c + idx);
            }
        };
        if (cached) {
            walk(cm0, cf0);
            walk(cm1, cf1);
        } else {
            for (uint32_t base_f = 0; base_f < n_src; base_f += 64u) {
                const uint32_t e = base_f + lane;
                bool keep = e < n_src;
                const uint32_t my_face = (keep && src) ? src[e] : e;
                if (CULL && keep) keep = !rect_culls(ftris[my_face], tile_rect);
                walk(__ballot(keep), my_face);
            }
        }
        if (any2(best.have)) {
            const f2 current_depth = 1.0f - depth_tex;  // compute.wgsl:210
            const bool fast = !__any(any2(best.have & ~depth_fast_domain(best.t)));
            const f2 depth = fast ? to_non_linear_depth_fast(best.t) : to_non_linear_depth(best.t);
            const i2 win = best.have & ~(depth >= current_depth);
            depth_tex = win ? (1.0f - depth) : depth_tex;
            obj = win ? i2{(int)best.idx.x, (int)best.idx.y} : obj;
            win_t = win ? best.t : win_t;
        }
        if (sample == 0u && in0) {  // depth / aux planes report sample 0
            kernarg<WfPrimaryArgs> *const qt = wf_args_again(ka);   // the targets (once per frame and pixel: addresses made here, not kept)
            float *const depth_plane = qt->tg.depth;
            depth_plane[pix0] = depth_tex.x;
            if (AUX) { qt->tg.obj_id[pix0] = obj.x; qt->tg.hit_t[pix0] = win_t.x; }
            if (in1) {
                depth_plane[pix0 + 1u] = depth_tex.y;
                if (AUX) { qt->tg.obj_id[pix0 + 1u] = obj.y; qt->tg.hit_t[pix0 + 1u] = win_t.y; }
            }
        }

        // -- local shading E(h0) and the surface's albedo ----------------------------------------------------------
        f2 cr = splat(0.0f), cg = splat(0.0f), cb = splat(0.0f);
        f2 tr = splat(0.0f), tgc = splat(0.0f), tb = splat(0.0f);
        if (__any(any2(obj >= 0))) {
            f2 mr, mg, mb, xr, xg, xb;
            kernarg<WfPrimaryArgs> *const qh = wf_args_again(ka);   // material constants, the texture, the shading records
            const ShadeRec *shade = qh->shade;
            const float4 *tex = qh->tex;
            if (NMAP) shade_mesh_pair<true, false, true>(qh->p, shade, tex, obj, last_shade, best, D, mr, mg, mb, xr, xg, xb);
            else if (qh->p.n_materials > 1u) shade_mesh_pair<true, false>(qh->p, shade, tex, obj, last_shade, best, D, mr, mg, mb, xr, xg, xb);
            else if (n_tested == 1u) shade_mesh_pair<false, true>(qh->p, shade, tex, obj, last_shade, best, D, mr, mg, mb, xr, xg, xb);
            else shade_mesh_pair<false, false>(qh->p, shade, tex, obj, last_shade, best, D, mr, mg, mb, xr, xg, xb);
            const i2 is_mesh = obj >= 0;
            cr = is_mesh ? mr : cr; cg = is_mesh ? mg : cg; cb = is_mesh ? mb : cb;
            tr = is_mesh ? xr : tr; tgc = is_mesh ? xg : tgc; tb = is_mesh ? xb : tb;
        }
        v3 n;  // the surface normal as the reference's HitRecord holds it (exact: it steers the bounce)
        n.x = n.y = splat(0.0f); n.z = splat(1.0f);
        if (__any(any2(obj < -1))) {  // sphere winners (few tiles): one pixel at a time
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int o = k ? obj.y : obj.x;
                if (o < -1) {
                    const f3 Dk = lane3(D, k);
                    const f3 center = ld3(wf_args_again(ka)->p.spheres[-2 - o].center);
                    const f3 P = along(O, k ? win_t.y : win_t.x, Dk);
                    const f3 c = shade_sphere(cnormalize(sub3(P, center)), Dk);
                    const f3 ne = normalize3(sub3(P, center));
                    if (k) { cr.y = c.x; cg.y = c.y; cb.y = c.z; tr.y = 1.0f; tgc.y = 0.0f; tb.y = 0.0f; n.x.y = ne.x; n.y.y = ne.y; n.z.y = ne.z; }
                    else { cr.x = c.x; cg.x = c.y; cb.x = c.z; tr.x = 1.0f; tgc.x = 0.0f; tb.x = 0.0f; n.x.x = ne.x; n.y.x = ne.y; n.z.x = ne.z; }
                }
            }
        }
        const i2 hit = obj != -1;
        {   // (cr, cg, cb are 0 where nothing was hit; float -> u32 conversion sends NaN / negatives to 0; alpha is 1 + 1 on a
            // written pixel, :231-234)
            constexpr float kScale22 = kWfFixedScale / 16.0f;
            constexpr uint32_t kCap = (kWfE0Cap << 22) - 1u;
            static_assert((unsigned long long)kWfMaxGroup * (kWfE0Cap << 22) <= (1ull << 32), "a launch group's sum of clamped terms fits 32 bits");
            const f2 sr = cr * kScale22, sg = cg * kScale22, sb = cb * kScale22;
            fr0 += min((uint32_t)sr.x, kCap); fg0 += min((uint32_t)sg.x, kCap); fb0 += min((uint32_t)sb.x, kCap);
            fr1 += min((uint32_t)sr.y, kCap); fg1 += min((uint32_t)sg.y, kCap); fb1 += min((uint32_t)sb.y, kCap);
            hits0 += hit.x ? 1u : 0u; hits1 += hit.y ? 1u : 0u;
        }

        // -- bounce ray of every pixel that hit something ----------------------------------------------------------
        kernarg<WfPrimaryArgs> *const qb = wf_args_again(ka);   // the ray queue
        if (qb->p.bounces != 0u) {
            const uint32_t wf_group = qb->wf.group;
            const TriRecord *tris = qb->tris;
            const i2 emit = hit & i2{in0 ? -1 : 0, in1 ? -1 : 0};
            const unsigned long long m0 = __ballot(emit.x != 0), m1 = __ballot(emit.y != 0);
            const uint32_t slot_base = (tile * wf_group + sidx) * kWfTilePixels + wave * 128u;
            if (lane == 0u) {
                unsigned long long *mk = qb->wf.masks + (size_t)(tile * wf_group + sidx) * 8u + wave * 2u;
                mk[0] = m0; mk[1] = m1;
            }
            emitted += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
            if (m0 | m1) {
                // mesh winners: +-normalize(N), prebaked with the shader's own operations (TriRecord::nhat)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int o = k ? obj.y : obj.x;
                    if (o >= 0) {
                        f3 nh = ld3(tris[o].nhat);
                        if ((k ? best.ndotd.y : best.ndotd.x) > 0.0f) nh = neg3(nh);   // compute.wgsl:140-142
                        if (k) { n.x.y = nh.x; n.y.y = nh.y; n.z.y = nh.z; } else { n.x.x = nh.x; n.y.x = nh.y; n.z.x = nh.z; }
                    }
                }
                const v3 P = along(splat3(O), win_t, D);
                const v3 O1 = v3{P.x + n.x * 1e-4f, P.y + n.y * 1e-4f, P.z + n.z * 1e-4f};
                const v3 D1 = bounce_direction_pair(n, base, emit);
                kernarg<WfPrimaryArgs> *const qq = wf_args_again(ka);   // where the rays go
                float4 *const rays = qq->wf.rays;
                uint16_t *const bins = qq->wf.bins;
                if (emit.x) {
                    const uint32_t slot = slot_base + lane;
                    rays[2u * slot] = make_float4(O1.x.x, O1.y.x, O1.z.x, wf_pack_unorm16x2(tr.x, tgc.x));
                    rays[2u * slot + 1u] = make_float4(D1.x.x, D1.y.x, D1.z.x, wf_pack_unorm16x2(tb.x, 0.0f));
                    bins[slot] = (uint16_t)wf_direction_bin(lane3(D1, 0));
                }
                if (emit.y) {
                    const uint32_t slot = slot_base + 64u + lane;
                    rays[2u * slot] = make_float4(O1.x.y, O1.y.y, O1.z.y, wf_pack_unorm16x2(tr.y, tgc.y));
                    rays[2u * slot + 1u] = make_float4(D1.x.y, D1.y.y, D1.z.y, wf_pack_unorm16x2(tb.y, 0.0f));
                    bins[slot] = (uint16_t)wf_direction_bin(lane3(D1, 1));
                }
            }
        }
    }

    // -- the frame's fixed-point sums: once per group and pixel (plain read-modify-write when this workgroup owns the tile's
    // samples alone, integer atomics when it shares them: the same bits either way) ------------------------------------------
    kernarg<WfPrimaryArgs> *const qf = wf_args_again(ka);   // the frame's planes
    if (in0 && !empty_tile) {
        unsigned long long *const fix = qf->wf.fix;
        const size_t plane = (size_t)qf->p.width * qf->p.height;
        const unsigned long long fa0 = (unsigned long long)hits0 * (unsigned long long)(2.0f * kWfFixedScale);
        const unsigned long long fa1 = (unsigned long long)hits1 * (unsigned long long)(2.0f * kWfFixedScale);
        const unsigned long long r0 = (unsigned long long)fr0 << 4, g0 = (unsigned long long)fg0 << 4, b0 = (unsigned long long)fb0 << 4;
        const unsigned long long r1 = (unsigned long long)fr1 << 4, g1 = (unsigned long long)fg1 << 4, b1 = (unsigned long long)fb1 << 4;
        if (!LIST && z_split == 1u && !qf->wf.shared_planes) {
            fix[pix0] += r0; fix[plane + pix0] += g0; fix[2u * plane + pix0] += b0; fix[3u * plane + pix0] += fa0;
            if (in1) { fix[pix0 + 1u] += r1; fix[plane + pix0 + 1u] += g1; fix[2u * plane + pix0 + 1u] += b1; fix[3u * plane + pix0 + 1u] += fa1; }
        } else {
            if (hits0) { atomicAdd(&fix[pix0], r0); atomicAdd(&fix[plane + pix0], g0); atomicAdd(&fix[2u * plane + pix0], b0); atomicAdd(&fix[3u * plane + pix0], fa0); }
            if (in1 && hits1) { atomicAdd(&fix[pix0 + 1u], r1); atomicAdd(&fix[plane + pix0 + 1u], g1); atomicAdd(&fix[2u * plane + pix0 + 1u], b1); atomicAdd(&fix[3u * plane + pix0 + 1u], fa1); }
        }
    }
    if (lane == 0u && emitted) atomicAdd(qf->wf.wave_total + tile * 4u + wave, emitted);   // (zeroed when the frame starts)
    } while (LIST);
}

hipError_t launch_wf_primary(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                             const FrameTri *ftris, const float4 *tex, const Targets &tg, const WfBuffers &wf,
                             uint32_t sample_begin, uint32_t sample_count, uint32_t z_split)
{
    if (fp.row_end <= fp.row_begin || fp.width == 0 || sample_count == 0) return hipSuccess;
    z_split = std::max(1u, std::min(z_split, sample_count));
    const dim3 grid((fp.width + kWfTileW - 1u) / kWfTileW, band_strips(fp), z_split);
    const dim3 block(256);
    const bool aux = (fp.flags & RWR_FLAG_AUX_OUTPUTS) != 0, do_cull = (fp.flags & RWR_FLAG_NO_CULL) == 0;
    const WfPrimaryArgs args{ftris, fp.n_tris, fp.row_begin, fp.bins.enabled, fp.mesh_px[0], fp.mesh_px[1], fp.mesh_px[2], fp.mesh_px[3],
                             sample_begin, sample_count, z_split, fp, tris, shade, tex, tg, wf};
#define RWR_WF_ARGS args
    const bool nmap = (fp.flags & RWR_FLAG_NORMAL_MAP) != 0;
#define RWR_WF_LAUNCH(A, C, N) hipLaunchKernelGGL((k_wf_primary<A, C, N>), grid, block, 0, s, RWR_WF_ARGS)
    if (wf.live_list && do_cull) {   // item after item of (live tile) x (share of its samples)
        const dim3 lgrid(std::min(grid.x * grid.y * z_split, 4096u));
#define RWR_WF_LAUNCH_LIST(A, N) hipLaunchKernelGGL((k_wf_primary<A, true, N, true>), lgrid, block, 0, s, RWR_WF_ARGS)
        if (nmap) { if (aux) RWR_WF_LAUNCH_LIST(true, true); else RWR_WF_LAUNCH_LIST(false, true); }
        else { if (aux) RWR_WF_LAUNCH_LIST(true, false); else RWR_WF_LAUNCH_LIST(false, false); }
#undef RWR_WF_LAUNCH_LIST
    } else if (nmap) {
        if (aux && do_cull) RWR_WF_LAUNCH(true, true, true);
        else if (aux) RWR_WF_LAUNCH(true, false, true);
        else if (do_cull) RWR_WF_LAUNCH(false, true, true);
        else RWR_WF_LAUNCH(false, false, true);
    } else {
        if (aux && do_cull) RWR_WF_LAUNCH(true, true, false);
        else if (aux) RWR_WF_LAUNCH(true, false, false);
        else if (do_cull) RWR_WF_LAUNCH(false, true, false);
        else RWR_WF_LAUNCH(false, false, false);
    }
#undef RWR_WF_LAUNCH
#undef RWR_WF_ARGS
    return hipGetLastError();
}

// Frames that show little (a small mesh on an empty screen): which tiles can anything be seen through?  One workgroup per
// 64x8-pixel tile, its four waves reaching the verdict the primary stage's waves would reach (rwr_wf_cull.h).  Tiles with a
// live wave are listed (any order); a wave that sees nothing gives its pixels the clear values of sample 0's planes here,
// and the resolve step skips its pixels' sums (nothing is ever added to them).
template <bool AUX>
__global__ void __launch_bounds__(256)
k_wf_classify(const FrameTri *__restrict__ ftris, uint32_t n_tris, uint32_t row_begin, uint32_t bins_enabled, int32_t mesh_x0,
              int32_t mesh_y0, int32_t mesh_x1, int32_t mesh_y1, const FrameParams p, const Targets tg, uint32_t tiles_x,
              uint32_t *__restrict__ live_list, uint32_t *__restrict__ live_count, uint32_t *__restrict__ tile_live)
{
    __shared__ uint32_t s_live;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (threadIdx.x == 0u) s_live = 0u;
    __syncthreads();
    const uint32_t blk_x0 = blockIdx.x * kWfTileW, blk_y0 = row_begin + blockIdx.y * p.row_pitch;
    const uint32_t tile_x0 = blk_x0 + (wave & 1u) * 32u, tile_y0 = blk_y0 + (wave >> 1) * 4u;
    const WfWaveCull wc = wf_wave_cull<true>(ftris, n_tris, row_begin, bins_enabled, mesh_x0, mesh_y0, mesh_x1, mesh_y1, p.bins, blk_x0,
                                             blk_y0, wave, lane);
    if (wf_wave_empty<true>(wc, p, (float)tile_x0, (float)tile_y0)) {
        const uint32_t px0 = tile_x0 + 2u * (lane & 15u), py = tile_y0 + (lane >> 4);
        if (py < p.row_end && px0 < p.width) {
            const uint32_t pix0 = py * p.width + px0;
            tg.depth[pix0] = 0.0f;
            if (AUX) { tg.obj_id[pix0] = -1; tg.hit_t[pix0] = 0.0f; }
            if (px0 + 1u < p.width) {
                tg.depth[pix0 + 1u] = 0.0f;
                if (AUX) { tg.obj_id[pix0 + 1u] = -1; tg.hit_t[pix0 + 1u] = 0.0f; }
            }
        }
    } else if (lane == 0u) {
        atomicOr(&s_live, 1u << wave);
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t tile = blockIdx.y * tiles_x + blockIdx.x, live = s_live;
        tile_live[tile] = live;
        if (live) live_list[atomicAdd(live_count, 1u)] = tile;
    }
}

hipError_t launch_wf_classify(hipStream_t s, const FrameParams &fp, const FrameTri *ftris, const Targets &tg, uint32_t tiles_x,
                              uint32_t *live_list, uint32_t *live_count, uint32_t *tile_live)
{
    if (fp.row_end <= fp.row_begin || fp.width == 0) return hipSuccess;
    const dim3 grid(tiles_x, band_strips(fp));
#define RWR_WF_CLASSIFY(A) hipLaunchKernelGGL((k_wf_classify<A>), grid, dim3(256), 0, s, ftris, fp.n_tris, fp.row_begin, fp.bins.enabled, \
        fp.mesh_px[0], fp.mesh_px[1], fp.mesh_px[2], fp.mesh_px[3], fp, tg, tiles_x, live_list, live_count, tile_live)
    if (fp.flags & RWR_FLAG_AUX_OUTPUTS) RWR_WF_CLASSIFY(true); else RWR_WF_CLASSIFY(false);
#undef RWR_WF_CLASSIFY
    return hipGetLastError();
}

hipError_t preload_kernels_wf_primary()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>((&k_wf_primary<false, true, false>)));
}

}  // namespace rwr
