// Wavefront integrator (extension; BASELINE.json configs 3-5 — the reference traces
// one centre ray per pixel and has no bounce).  Per sample pass:
//
//   k_wf_primary  one lane per pixel: jittered primary ray through the same fused
//                 visibility code as the frame kernel (rwr_primary.h), local shading
//                 E(h0) added to the RGBA32F accumulator; pixels that hit a surface
//                 build their cosine-distributed bounce ray and APPEND it to the ray
//                 queue in HBM.  Compaction is wave-ballot + popcount prefix, then a
//                 4-entry prefix over the workgroup's waves in LDS; every workgroup
//                 owns the 256-slot segment [wg*256, wg*256 + count) of the queue and
//                 publishes `count`.  No global atomics: one counter word shared by
//                 32 400 waves saturates at ~88 atomics/us on MI355X (measured: the
//                 pass took 377 us with it, 14x the fused frame kernel).  Rays that
//                 left the scene cost nothing downstream.
//   k_wf_bounce   workgroup b takes segment b: one lane per queued ray (coalesced
//                 16-byte SoA loads), empty segments exit at once: analytic
//                 spheres + per-lane BVH traversal with LDS-staged nodelets
//                 (rwr_bvh.h), shading of the second hit, accumulator += albedo * E(h1).
//   k_wf_resolve  accumulator / spp -> RGBA8 (+ float plane).
//
// Pixels and the RNG are keyed by GLOBAL pixel index, so any row-band split over GPUs
// produces the same bits.
#include "rwr_bvh.h"
#include "rwr_primary.h"

namespace rwr {

template <bool AUX, bool CULL>
__global__ void __launch_bounds__(256, 8)
k_wf_primary(const FrameParams p, const TriRecord *__restrict__ tris, const ShadeRec *__restrict__ shade,
             const FrameTri *__restrict__ ftris, const float4 *__restrict__ tex,
             const Targets tg, const WfBuffers wf)
{
    __shared__ PrimaryShared s_prim;

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t blk_x0 = blockIdx.x * 32u;
    const uint32_t tile_x0 = blk_x0 + wave * 8u;
    const uint32_t tile_y0 = p.row_begin + blockIdx.y * 8u;
    const uint32_t px = tile_x0 + (lane & 7u), py = tile_y0 + (lane >> 3);
    const bool in_range = (px < p.width) && (py < p.row_end);
    const uint32_t pixel = py * p.width + px;  // GLOBAL pixel index: RNG key and accumulator slot

    float jx = 0.5f, jy = 0.5f;
    if (p.spp > 1u) {
        jx = rng_uniform(pixel, p.sample, 0u, p.seed);
        jy = rng_uniform(pixel, p.sample, 1u, p.seed);
    }
    const f3 O = ld3(p.cam.origin);
    const f3 D = pixel_to_ray_dir(p.cam, px, py, jx, jy, p.width, p.height);

    PrimaryHit r;
    uint32_t dl = 0, dt = 0;
    primary_visibility<CULL, false>(p, tris, ftris, s_prim, blk_x0, tile_x0, tile_y0, O, D, r, dl, dt);

    const bool hit = r.obj != -1;
    float4 e0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    f3 albedo = mk3(0.0f, 0.0f, 0.0f);
    if (hit) {
        const Shaded sw = shade_winner(p, r.obj, r.t, r.mesh.u, r.mesh.v, r.mesh.ndotd, shade, tex, O, D);
        albedo = sw.albedo;
        e0 = make_float4(sw.colour.x, sw.colour.y, sw.colour.z, 2.0f);
    }
    if (in_range) {
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (p.sample != 0u) acc = wf.accum[pixel];
        acc.x += e0.x; acc.y += e0.y; acc.z += e0.z; acc.w += e0.w;
        wf.accum[pixel] = acc;
        if (p.sample == 0u) {  // depth / aux planes report sample 0
            tg.depth[pixel] = r.depth_tex;
            if (AUX) {
                tg.obj_id[pixel] = r.obj;
                tg.hit_t[pixel] = r.t;
            }
        }
    }

    // -- bounce ray generation + wavefront compaction ---------------------------
    const bool emit = hit && in_range && p.bounces != 0u;
    f3 O1 = mk3(0, 0, 0), D1 = mk3(0, 0, 1);
    if (emit) {
        // the surface normal as the reference's HitRecord holds it (exact: it steers the bounce)
        f3 n;
        const f3 P = along(O, r.t, D);
        if (r.obj >= 0) {
            f3 N = ld3(tris[r.obj].N);
            if (r.mesh.ndotd > 0.0f) N = neg3(N);
            n = normalize3(N);
        } else {
            n = normalize3(sub3(P, ld3(p.spheres[-2 - r.obj].center)));
        }
        O1 = mk3(P.x + n.x * 1e-4f, P.y + n.y * 1e-4f, P.z + n.z * 1e-4f);
        D1 = bounce_direction(n, pixel, p.sample, p.seed);
    }
    // compaction: wave ballot + prefix, then a prefix over the 4 waves of the workgroup
    const unsigned long long m = __ballot(emit);
    __syncthreads();  // s_prim.wave_cnt is free again (all waves are past the mesh loop)
    if (lane == 0) s_prim.wave_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t c = s_prim.wave_cnt[w];
        off += (w < wave) ? c : 0u;
        total += c;
    }
    const uint32_t wg = blockIdx.y * gridDim.x + blockIdx.x;  // this workgroup's queue segment
    if (threadIdx.x == 0) {
        wf.seg_count[wg] = total;
        wf.seg_total[wg] = (p.sample == 0u ? 0u : wf.seg_total[wg]) + total;  // bounce rays of the whole frame
    }
    if (emit) {
        const uint32_t slot = wg * 256u + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        wf.q0[slot] = make_float4(O1.x, O1.y, O1.z, __uint_as_float(pixel));
        wf.q1[slot] = make_float4(D1.x, D1.y, D1.z, albedo.x);
        wf.q2[slot] = make_float2(albedo.y, albedo.z);
    }
}

template <bool NODES_IN_LDS>
__global__ void __launch_bounds__(256)
k_wf_bounce(const FrameParams p, const TriRecord *__restrict__ tris, const ShadeRec *__restrict__ shade,
            const BvhDevice bvh, const float4 *__restrict__ tex, const WfBuffers wf)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const uint32_t count = wf.seg_count[blockIdx.x];  // rays in this workgroup's queue segment
    if (count == 0u) return;                          // uniform

    // LDS carve: [nodelets][stack]
    BvhNode4 *s_nodes = reinterpret_cast<BvhNode4 *>(s_dyn);
    const uint32_t node_bytes = NODES_IN_LDS ? bvh.n_nodes * (uint32_t)sizeof(BvhNode4) : 0u;
    uint32_t *s_stack = reinterpret_cast<uint32_t *>(s_dyn + node_bytes);
    if (NODES_IN_LDS) {
        const float4 *src = reinterpret_cast<const float4 *>(bvh.nodes);
        float4 *dst = reinterpret_cast<float4 *>(s_nodes);
        for (uint32_t i = threadIdx.x; i < bvh.n_nodes * 8u; i += 256u) dst[i] = src[i];
    }
    __syncthreads();

    if (threadIdx.x >= count) return;  // no barrier below
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const float4 a = wf.q0[i], b = wf.q1[i];
    const float2 c = wf.q2[i];
    const f3 O = mk3(a.x, a.y, a.z), D = mk3(b.x, b.y, b.z);
    const uint32_t pixel = __float_as_uint(a.w);
    const f3 thr = mk3(b.w, c.x, c.y);

    // nearest over spheres (in order), then the mesh; strict '<' keeps the earlier candidate on ties
    bool have = false;
    float best_t = 0.0f;
    int32_t obj = -1;
    for (uint32_t s = 0; s < p.n_spheres; s++) {
        float t;
        if (sphere_ray_intersect_t(ld3(p.spheres[s].center), p.spheres[s].radius, O, D, t)) {
            if (!have || t < best_t) { have = true; best_t = t; obj = -2 - (int32_t)s; }
        }
    }
    MeshHit mh;
    mh.have = false; mh.t = 0.0f; mh.u = 0.0f; mh.v = 0.0f; mh.ndotd = 0.0f; mh.idx = 0u;
    if (p.n_tris) {
        if (NODES_IN_LDS) bvh_nearest(s_nodes, bvh.leaf_faces, tris, p.n_tris, s_stack, O, D, mh);
        else bvh_nearest(bvh.nodes, bvh.leaf_faces, tris, p.n_tris, s_stack, O, D, mh);
        if (mh.have && (!have || mh.t < best_t)) { have = true; best_t = mh.t; obj = (int32_t)mh.idx; }
    }
    if (!have) return;

    PrimaryHit r;
    r.depth_tex = 0.0f; r.obj = obj; r.t = best_t; r.mesh = mh;
    const f3 e1 = shade_winner(p, r.obj, r.t, r.mesh.u, r.mesh.v, r.mesh.ndotd, shade, tex, O, D).colour;
    float4 acc = wf.accum[pixel];
    acc.x += thr.x * e1.x; acc.y += thr.y * e1.y; acc.z += thr.z * e1.z;
    wf.accum[pixel] = acc;
}

// RWR_FLAG_USE_BVH: the reference frame with the mesh pass done by per-lane BVH traversal
// instead of candidate lists — for views where many small faces fall into one tile (a distant
// or finely tessellated mesh), where a wave would otherwise walk every face of its block.
// Same spheres, same exact hit test, ties by face index: results are bit-identical to k_primary.
// A separate kernel on purpose: inlined into (or called from) k_primary the traversal's register
// needs slowed EVERY frame 2.5x.
template <bool AUX, bool NODES_IN_LDS>
__global__ void __launch_bounds__(256)
k_primary_bvh(const FrameParams p, const TriRecord *__restrict__ tris, const ShadeRec *__restrict__ shade,
              const BvhDevice bvh, const float4 *__restrict__ tex, const Targets tg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    BvhNode4 *s_nodes = reinterpret_cast<BvhNode4 *>(s_dyn);
    const uint32_t node_bytes = NODES_IN_LDS ? bvh.n_nodes * (uint32_t)sizeof(BvhNode4) : 0u;
    uint32_t *s_stack = reinterpret_cast<uint32_t *>(s_dyn + node_bytes);
    if (NODES_IN_LDS) {
        const float4 *src = reinterpret_cast<const float4 *>(bvh.nodes);
        float4 *dst = reinterpret_cast<float4 *>(s_nodes);
        for (uint32_t i = threadIdx.x; i < bvh.n_nodes * 8u; i += 256u) dst[i] = src[i];
        __syncthreads();
    }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t px = blockIdx.x * 32u + wave * 8u + (lane & 7u), py = p.row_begin + blockIdx.y * 8u + (lane >> 3);
    const bool in_range = (px < p.width) && (py < p.row_end);
    const f3 O = ld3(p.cam.origin);
    const f3 D = pixel_to_ray_dir(p.cam, px, py, 0.5f, 0.5f, p.width, p.height);

    PrimaryHit r;
    r.depth_tex = 0.0f; r.obj = -1; r.t = 0.0f;
    for (uint32_t s = 0; s < p.n_spheres; s++) {  // sphere passes in order (lib.rs:1106-1173)
        float t;
        if (sphere_ray_intersect_t(ld3(p.spheres[s].center), p.spheres[s].radius, O, D, t)) {
            const float current_depth = 1.0f - r.depth_tex;
            const float depth = to_non_linear_depth(t);
            if (!(depth >= current_depth)) { r.depth_tex = 1.0f - depth; r.obj = -2 - (int32_t)s; r.t = t; }
        }
    }
    MeshHit best;
    best.have = false; best.t = 0.0f; best.u = 0.0f; best.v = 0.0f; best.ndotd = 0.0f; best.idx = 0u;
    // wave-uniform: the wave's 8x8 tile lies outside the screen rectangle of the whole mesh (FrameParams::mesh_rect)
    const float tx = (float)(blockIdx.x * 32u + wave * 8u), ty = (float)(p.row_begin + blockIdx.y * 8u);
    const bool outside = !(p.flags & RWR_FLAG_NO_CULL) && ((tx + 8.0f < p.mesh_rect[0]) || (tx > p.mesh_rect[2]) ||
                                                          (ty + 8.0f < p.mesh_rect[1]) || (ty > p.mesh_rect[3]));
    if (p.n_tris && !outside) {
        if (NODES_IN_LDS) bvh_nearest(s_nodes, bvh.leaf_faces, tris, p.n_tris, s_stack, O, D, best);
        else bvh_nearest(bvh.nodes, bvh.leaf_faces, tris, p.n_tris, s_stack, O, D, best);
    }
    r.mesh = best;
    if (best.have) {  // mesh pass depth test (compute.wgsl:210-215)
        const float current_depth = 1.0f - r.depth_tex;
        const float depth = to_non_linear_depth_auto(best.t);
        if (!(depth >= current_depth)) { r.depth_tex = 1.0f - depth; r.obj = (int32_t)best.idx; r.t = best.t; }
    }
    float cr = 0.0f, cg = 0.0f, cb = 0.0f, ca = 0.0f;
    if (r.obj != -1) {
        const f3 c = shade_winner(p, r.obj, r.t, r.mesh.u, r.mesh.v, r.mesh.ndotd, shade, tex, O, D).colour;
        cr = c.x; cg = c.y; cb = c.z; ca = 2.0f;
    }
    if (in_range) {
        const size_t o = (size_t)py * p.width + px;
        reinterpret_cast<uint32_t *>(tg.color)[o] = pack_rgba8(cr, cg, cb, ca);
        tg.depth[o] = r.depth_tex;
        if (AUX) {
            reinterpret_cast<float4 *>(tg.color_f32)[o] = make_float4(cr, cg, cb, ca);
            tg.obj_id[o] = r.obj;
            tg.hit_t[o] = r.t;
        }
    }
}

hipError_t launch_primary_bvh(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                              const BvhDevice &bvh, const float4 *tex, const Targets &tg)
{
    if (fp.row_end <= fp.row_begin || fp.width == 0) return hipSuccess;
    const dim3 grid((fp.width + 31u) / 32u, (fp.row_end - fp.row_begin + 7u) / 8u);
    const size_t fixed = (size_t)bvh.stack_depth * 256u * 4u;
    const size_t node_bytes = (size_t)bvh.n_nodes * sizeof(BvhNode4);
    const bool aux = (fp.flags & RWR_FLAG_AUX_OUTPUTS) != 0;
    if (node_bytes + fixed <= 64u * 1024u) {
        if (aux) hipLaunchKernelGGL((k_primary_bvh<true, true>), grid, dim3(256), node_bytes + fixed, s, fp, tris, shade, bvh, tex, tg);
        else hipLaunchKernelGGL((k_primary_bvh<false, true>), grid, dim3(256), node_bytes + fixed, s, fp, tris, shade, bvh, tex, tg);
    } else {
        if (aux) hipLaunchKernelGGL((k_primary_bvh<true, false>), grid, dim3(256), fixed, s, fp, tris, shade, bvh, tex, tg);
        else hipLaunchKernelGGL((k_primary_bvh<false, false>), grid, dim3(256), fixed, s, fp, tris, shade, bvh, tex, tg);
    }
    return hipGetLastError();
}

template <bool AUX>
__global__ void __launch_bounds__(256)
k_wf_resolve(const FrameParams p, const Targets tg, const WfBuffers wf)
{
    const uint32_t n = p.width * (p.row_end - p.row_begin);
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t pixel = p.row_begin * p.width + i;
    const float4 acc = wf.accum[pixel];
    const float fs = (float)p.spp;
    const float r = acc.x / fs, g = acc.y / fs, b = acc.z / fs, a = acc.w / fs;
    reinterpret_cast<uint32_t *>(tg.color)[pixel] = pack_rgba8(r, g, b, a);
    if (AUX) reinterpret_cast<float4 *>(tg.color_f32)[pixel] = make_float4(r, g, b, a);
}

hipError_t launch_wf_primary(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                             const FrameTri *ftris, const float4 *tex, const Targets &tg,
                             const WfBuffers &wf)
{
    if (fp.row_end <= fp.row_begin || fp.width == 0) return hipSuccess;
    const dim3 grid((fp.width + 31u) / 32u, (fp.row_end - fp.row_begin + 7u) / 8u);
    const bool aux = (fp.flags & RWR_FLAG_AUX_OUTPUTS) != 0, do_cull = (fp.flags & RWR_FLAG_NO_CULL) == 0;
    if (aux && do_cull) hipLaunchKernelGGL((k_wf_primary<true, true>), grid, dim3(256), 0, s, fp, tris, shade, ftris, tex, tg, wf);
    else if (aux) hipLaunchKernelGGL((k_wf_primary<true, false>), grid, dim3(256), 0, s, fp, tris, shade, ftris, tex, tg, wf);
    else if (do_cull) hipLaunchKernelGGL((k_wf_primary<false, true>), grid, dim3(256), 0, s, fp, tris, shade, ftris, tex, tg, wf);
    else hipLaunchKernelGGL((k_wf_primary<false, false>), grid, dim3(256), 0, s, fp, tris, shade, ftris, tex, tg, wf);
    return hipGetLastError();
}

hipError_t launch_wf_bounce(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                            const BvhDevice &bvh, const float4 *tex, const WfBuffers &wf,
                            uint32_t n_segments)
{
    if (n_segments == 0) return hipSuccess;
    const dim3 grid(n_segments);
    const size_t fixed = (size_t)bvh.stack_depth * 256u * 4u;
    const size_t node_bytes = (size_t)bvh.n_nodes * sizeof(BvhNode4);
    // nodelets go to LDS when they leave room for >= 2 workgroups per CU (160 KiB LDS)
    // nodelets go to LDS when they leave room for >= 2 workgroups per CU (160 KiB LDS).
    // (A persistent one-wave-per-segment form with dynamic ray hand-out was built and measured
    // 35 % SLOWER — 26.2 vs 19.7 ms at cfg3: a quarter of the waves, so less latency hiding and a
    // longer tail, for no gain in lane utilisation once finished rays are retired in groups.)
    if (node_bytes + fixed <= 64u * 1024u) {
        hipLaunchKernelGGL((k_wf_bounce<true>), grid, dim3(256), node_bytes + fixed, s, fp, tris, shade, bvh, tex, wf);
    } else {
        hipLaunchKernelGGL((k_wf_bounce<false>), grid, dim3(256), fixed, s, fp, tris, shade, bvh, tex, wf);
    }
    return hipGetLastError();
}

hipError_t launch_wf_resolve(hipStream_t s, const FrameParams &fp, const Targets &tg, const WfBuffers &wf)
{
    if (fp.row_end <= fp.row_begin || fp.width == 0) return hipSuccess;
    const uint32_t n = fp.width * (fp.row_end - fp.row_begin);
    const dim3 grid((n + 255u) / 256u);
    if (fp.flags & RWR_FLAG_AUX_OUTPUTS) hipLaunchKernelGGL((k_wf_resolve<true>), grid, dim3(256), 0, s, fp, tg, wf);
    else hipLaunchKernelGGL((k_wf_resolve<false>), grid, dim3(256), 0, s, fp, tg, wf);
    return hipGetLastError();
}

hipError_t preload_kernels_wavefront()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>((&k_wf_primary<false, true>)));
}

}  // namespace rwr
