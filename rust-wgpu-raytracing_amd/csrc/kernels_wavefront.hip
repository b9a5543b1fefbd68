// The reference frame by per-ray BVH traversal (k_primary_bvh, RWR_FLAG_USE_BVH / dense views) and the last step
// of the wavefront integrator (k_wf_resolve: accumulator / spp -> RGBA8).  The integrator's two stages live in
// kernels_wf_primary.hip and kernels_wf_bounce.hip.
#include "rwr_bvh.h"
#include "rwr_primary.h"

namespace rwr {

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
    const uint32_t px = blockIdx.x * 32u + wave * 8u + (lane & 7u), py = p.row_begin + blockIdx.y * p.row_pitch + (lane >> 3);
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
    const float tx = (float)(blockIdx.x * 32u + wave * 8u), ty = (float)(p.row_begin + blockIdx.y * p.row_pitch);
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
        const f3 c = ((p.flags & RWR_FLAG_NORMAL_MAP) ? shade_winner<true>(p, r.obj, r.t, r.mesh.u, r.mesh.v, r.mesh.ndotd, shade, tex, O, D) : shade_winner<false>(p, r.obj, r.t, r.mesh.u, r.mesh.v, r.mesh.ndotd, shade, tex, O, D)).colour;
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
    const dim3 grid((fp.width + 31u) / 32u, band_strips(fp));
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
k_wf_resolve(const FrameParams p, const Targets tg, WfBuffers wf, const uint32_t *__restrict__ live_counters, uint32_t *__restrict__ host_live)
{
    // the last launch group's live pool counts {packets, per-lane} for the host (it sizes the NEXT frame's schedule by them, reads
    // them a frame late and never waits): stored straight to pinned memory here instead of by a copy command on the stream
    if (host_live && (blockIdx.x | blockIdx.y) == 0u && threadIdx.x < 2u) host_live[threadIdx.x] = live_counters[threadIdx.x];
    // a workgroup = 64 x 4 pixels (half a tile of the band's strip blockIdx.y / 2): one row of 64 per wave, whole 512-byte pieces
    // of every plane
    const uint32_t strip = blockIdx.y >> 1, in_strip = (blockIdx.y & 1u) * 4u + (threadIdx.x >> 6);
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y = p.row_begin + strip * p.row_pitch + in_strip;
    if (x >= p.width || y >= p.row_end) return;
    const uint32_t pixel = y * p.width + x;
    if (wf.tile_live) {   // a piece of a tile nothing can be seen through (k_wf_classify): its sums were never touched
        const uint32_t live = wf.tile_live[strip * wf.tiles_x + blockIdx.x];
        const uint32_t piece = (blockIdx.y & 1u) * 2u + ((threadIdx.x >> 5) & 1u);
        if (!((live >> piece) & 1u)) {
            reinterpret_cast<uint32_t *>(tg.color)[pixel] = 0u;
            if (AUX) reinterpret_cast<float4 *>(tg.color_f32)[pixel] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            return;
        }
    }
    // the frame's fixed-point sums (2^-26 units; kernels_wf_primary.hip, kernels_wf_bounce.hip), read and left zeroed for
    // the next frame
    const size_t plane = (size_t)p.width * p.height;
    const float unit = 1.0f / kWfFixedScale, fs = (float)p.spp;
    const float r = (float)wf.fix[pixel] * unit / fs, g = (float)wf.fix[plane + pixel] * unit / fs;
    const float b = (float)wf.fix[2u * plane + pixel] * unit / fs, a = (float)wf.fix[3u * plane + pixel] * unit / fs;
    wf.fix[pixel] = 0ull; wf.fix[plane + pixel] = 0ull; wf.fix[2u * plane + pixel] = 0ull; wf.fix[3u * plane + pixel] = 0ull;
    reinterpret_cast<uint32_t *>(tg.color)[pixel] = pack_rgba8(r, g, b, a);
    if (AUX) reinterpret_cast<float4 *>(tg.color_f32)[pixel] = make_float4(r, g, b, a);
}

hipError_t launch_wf_resolve(hipStream_t s, const FrameParams &fp, const Targets &tg, const WfBuffers &wf, const uint32_t *live_counters,
                             uint32_t *host_live)
{
    if (fp.row_end <= fp.row_begin || fp.width == 0) return hipSuccess;
    const dim3 grid((fp.width + 63u) / 64u, 2u * band_strips(fp));
    if (fp.flags & RWR_FLAG_AUX_OUTPUTS) hipLaunchKernelGGL((k_wf_resolve<true>), grid, dim3(256), 0, s, fp, tg, wf, live_counters, host_live);
    else hipLaunchKernelGGL((k_wf_resolve<false>), grid, dim3(256), 0, s, fp, tg, wf, live_counters, host_live);
    return hipGetLastError();
}

hipError_t preload_kernels_wavefront()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>((&k_wf_resolve<false>)));
}

}  // namespace rwr
