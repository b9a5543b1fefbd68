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
//  * all primary rays share the camera origin, so faces are culled against ray
//    frusta, two levels deep: the workgroup's 256 threads test 256 faces at a
//    time against the 32x8 block frustum and compact the survivors (ballot +
//    prefix popcount, order-preserving) into an LDS candidate list; each wave
//    then tests the list entries, one per lane, against its own 8x8 tile
//    frustum, __ballots again and walks the set bits in ascending face order.
//    The face index is wave-uniform there, so its 128-byte TriRecord arrives
//    through scalar loads and sits in SGPRs while 64 rays are tested against it.
//    Ascending order + strict '<' reproduces the reference's lowest-index tie
//    rule; skipped faces are ones no ray of the tile can hit, so the result is
//    bit-identical to the brute-force loop (checked against RWR_FLAG_NO_CULL).
#include "rwr_frame_setup.h"
#include "rwr_primary.h"

namespace rwr {

// ---------------------------------------------------------------------------
// Scene prebake: ModelVertexSmall[] + ModelFaceSmall[] (+ rigid instances)
// -> TriRecord[] + ShadeRec[] + CullRec[].  One thread per (instance, face).
__global__ void __launch_bounds__(256)
k_prebake(const rwr_model_vertex_small *__restrict__ verts, const rwr_model_face_small *__restrict__ faces,
          const uint32_t *__restrict__ face_material, uint32_t n_faces, const rwr_instance_raw *__restrict__ instances,
          uint32_t n_instances, const MaterialRec *__restrict__ materials,
          TriRecord *__restrict__ tris, ShadeRec *__restrict__ shade, CullRec *__restrict__ cull, TangentRec *__restrict__ tangents)
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
    const f3 nh = normalize3(N);   // literal f32 (IEEE sqrt / divide in this translation unit)
    T.nhat[0] = nh.x; T.nhat[1] = nh.y; T.nhat[2] = nh.z; T.pad5 = 0.0f;
    tris[i] = T;
    // Shading record (colour path): the face-only part of compute.wgsl:217-234 in double, rounded once.
    // A degenerate face (N = 0) gets non-finite values and can never be hit (:94).
    ShadeRec S;
    const uint32_t mat = face_material ? face_material[f] : 0u;
    const double tw = (double)materials[mat].tex_w, th = (double)materials[mat].tex_h;
    const double den = (double)T.denom, inv_len = 1.0 / sqrt(den);
    const double nx = N.x * inv_len, ny = N.y * inv_len, nz = N.z * inv_len;
    const double inv_l = 1.0 / sqrt(27.0);  // -normalize(vec3(1, -1, -5)), :55
    S.n[0] = (float)nx; S.n[1] = (float)ny; S.n[2] = (float)nz;
    S.ndl0 = (float)((-nx + ny + 5.0 * nz) * inv_l);
    // barycentric = (u, v, den - u - v) / den on (v0, v1, v2) (:144-147); texel x = tw * tex.x - 0.5,
    // texel y = th * (1 - tex.y) - 0.5 (:224 and the sampler's half-texel offset)
    const double u0 = v0.tex_coords[0], u1 = v1.tex_coords[0], u2 = v2.tex_coords[0];
    const double w0 = v0.tex_coords[1], w1 = v1.tex_coords[1], w2 = v2.tex_coords[1];
    S.c0[0] = (float)(tw * u2 - 0.5);         S.c0[1] = (float)(th * (1.0 - w2) - 0.5);
    S.c1[0] = (float)(tw * (u0 - u2) / den);  S.c1[1] = (float)(-th * (w0 - w2) / den);
    S.c2[0] = (float)(tw * (u1 - u2) / den);  S.c2[1] = (float)(-th * (w1 - w2) / den);
    S.material = mat;
    S.pad = 0.0f;
    shade[i] = S;
    {
        // Tangent frame for normal-mapped shading (extension; oracle/rt_oracle.c normal_mapped): T = dP/du, B = -dP/dv' in
        // the sampling space (u, v' = 1 - v), Gram-Schmidt against the unit normal as wound, in double.
        const double d1[3] = {(double)p1.x - p0.x, (double)p1.y - p0.y, (double)p1.z - p0.z};
        const double d2[3] = {(double)p2.x - p0.x, (double)p2.y - p0.y, (double)p2.z - p0.z};
        double ng[3] = {d1[1] * d2[2] - d1[2] * d2[1], d1[2] * d2[0] - d1[0] * d2[2], d1[0] * d2[1] - d1[1] * d2[0]};
        const double nl = sqrt(ng[0] * ng[0] + ng[1] * ng[1] + ng[2] * ng[2]);
        ng[0] /= nl; ng[1] /= nl; ng[2] /= nl;
        const double du1 = u1 - u0, dv1 = (1.0 - w1) - (1.0 - w0), du2 = u2 - u0, dv2 = (1.0 - w2) - (1.0 - w0);
        const double det = du1 * dv2 - dv1 * du2;
        double th[3] = {0.0, 0.0, 0.0}, bh[3] = {0.0, 0.0, 0.0};
        if (det != 0.0 && isfinite(1.0 / det)) {
            const double r = 1.0 / det;
            double T[3], B[3], tt[3];
            for (int c = 0; c < 3; c++) { T[c] = (d1[c] * dv2 - d2[c] * dv1) * r; B[c] = (d2[c] * du1 - d1[c] * du2) * -r; }
            const double nt = ng[0] * T[0] + ng[1] * T[1] + ng[2] * T[2];
            for (int c = 0; c < 3; c++) tt[c] = T[c] - ng[c] * nt;
            const double tl = sqrt(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]);
            if (tl > 0.0 && isfinite(tl)) {
                for (int c = 0; c < 3; c++) th[c] = tt[c] / tl;
                bh[0] = ng[1] * th[2] - ng[2] * th[1]; bh[1] = ng[2] * th[0] - ng[0] * th[2]; bh[2] = ng[0] * th[1] - ng[1] * th[0];
                if (bh[0] * B[0] + bh[1] * B[1] + bh[2] * B[2] < 0.0) { bh[0] = -bh[0]; bh[1] = -bh[1]; bh[2] = -bh[2]; }
            }
        }
        TangentRec G;
        G.t[0] = (float)th[0]; G.t[1] = (float)th[1]; G.t[2] = (float)th[2]; G.pad0 = 0.0f;
        G.b[0] = (float)bh[0]; G.b[1] = (float)bh[1]; G.b[2] = (float)bh[2]; G.pad1 = 0.0f;
        tangents[i] = G;
    }
    CullRec R;
    R.p0[0] = p0.x; R.p0[1] = p0.y; R.p0[2] = p0.z;
    R.p1[0] = p1.x; R.p1[1] = p1.y; R.p1[2] = p1.z;
    R.p2[0] = p2.x; R.p2[1] = p2.y; R.p2[2] = p2.z;
    R.pad[0] = R.pad[1] = R.pad[2] = 0.0f;
    cull[i] = R;
}

hipError_t launch_prebake(hipStream_t s, const rwr_model_vertex_small *verts, const rwr_model_face_small *faces,
                          const uint32_t *face_material, uint32_t n_faces, const rwr_instance_raw *instances,
                          uint32_t n_instances, const MaterialRec *materials, TriRecord *tris, ShadeRec *shade, CullRec *cull,
                          TangentRec *tangents)
{
    const uint32_t total = n_faces * (n_instances ? n_instances : 1u);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prebake, dim3((total + 255) / 256), dim3(256), 0, s, verts, faces, face_material, n_faces,
                       instances, n_instances, materials, tris, shade, cull, tangents);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Per-frame setup, one launch: blocks [0, nb_tris) make one face each per thread (the culling record
// of rwr_cull.h and the ray-independent numerator of the plane distance), the blocks after them fill
// the ray tables (FrameParams::ray_colp / ray_row), one column pair or one row per thread.
__global__ void __launch_bounds__(256)
k_frame_setup(const CullConsts cc, const rwr_camera_inv_uniform cam, uint32_t width, uint32_t height,
              const CullRec *__restrict__ cull, const TriRecord *__restrict__ tris, uint32_t n_tris, uint32_t nb_tris,
              const FrameSetupOut out)
{
    frame_setup_block(blockIdx.x, gridDim.x, cc, cam, width, height, cull, tris, n_tris, nb_tris, out);   // rwr_frame_setup.h
}

hipError_t launch_frame_setup(hipStream_t s, const CullConsts &cc, const rwr_camera_inv_uniform &cam, uint32_t width,
                              uint32_t height, const CullRec *cull, const TriRecord *tris, uint32_t n_tris,
                              const FrameSetupOut &out)
{
    const uint32_t nb_tris = (n_tris + 255u) / 256u, nb_tab = (out.ray_pairs + out.ray_rows + 255u) / 256u;
    if (nb_tris + nb_tab == 0) return hipSuccess;
    hipLaunchKernelGGL(k_frame_setup, dim3(nb_tris + nb_tab), dim3(256), 0, s, cc, cam, width, height, cull, tris, n_tris,
                       nb_tris, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Per-frame screen binning, sized by a count pass: k_bin_faces<false> counts, per 64x32-pixel bin, the faces its
// rectangle cannot reject; k_bin_scan turns the counts into list offsets (exclusive scan) and reports the total;
// k_bin_faces<true> walks the faces again and writes every bin's ascending list at its offset.  One workgroup per bin;
// each of its four waves owns one contiguous quarter of the faces and counts (then fills) it by ballot + prefix popcount
// with no barrier and nothing shared — the count pass leaves one count per (bin, wave), so the fill pass knows where every
// wave's part of the list begins.  Lists stay in face order: the lowest-index tie rule survives.  When the total exceeds
// the buffer's capacity the scan marks every bin kBinNoList — the render kernels then walk the whole scene for this
// frame, the same pixels — and the context allocates more for the next frames.
template <bool FILL>
__global__ void __launch_bounds__(256)
k_bin_faces(const FrameTri *__restrict__ ftris, uint32_t n_tris, uint32_t row_begin, uint32_t *__restrict__ lists,
            uint32_t *__restrict__ counts4, const uint32_t *__restrict__ offsets, uint32_t bins_x, int32_t mesh_x0, int32_t mesh_y0,
            int32_t mesh_x1, int32_t mesh_y1)
{
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint32_t bin = blockIdx.y * bins_x + blockIdx.x;
    {   // a bin outside the screen rectangle of the whole mesh (FrameParams::mesh_px: the render kernels skip the mesh pass for every
        // tile of it and never look at its list) has nothing to walk: on a frame that shows a small mesh that is most bins
        const int32_t bx0 = (int32_t)(blockIdx.x * kBinW), by0 = (int32_t)(row_begin + blockIdx.y * kBinH);
        if (bx0 + (int32_t)kBinW < mesh_x0 || bx0 > mesh_x1 || by0 + (int32_t)kBinH < mesh_y0 || by0 > mesh_y1) {
            if (!FILL && lane == 0u) counts4[bin * 4u + wave] = 0u;
            return;
        }
    }
    const uint32_t quarter = ((n_tris + 255u) / 256u) * 64u;
    const uint32_t begin = wave * quarter, end = min(n_tris, begin + quarter);
    uint32_t *__restrict__ out = nullptr;
    if (FILL) {
        const uint32_t off = offsets[bin];
        if (off == kBinNoList) return;   // uniform
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; w++) before += counts4[bin * 4u + w];
        out = lists + off + before;
    }
    const float x0 = (float)(blockIdx.x * kBinW), y0 = (float)(row_begin + blockIdx.y * kBinH);
    const TileRect rect = {x0, y0, x0 + (float)kBinW, y0 + (float)kBinH};
    uint32_t written = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t base = begin; base < end; base += 128u) {   // two records per lane in flight before the first test
        const uint32_t j0 = base + lane, j1 = j0 + 64u;
        const bool in0 = j0 < end, in1 = j1 < end;
        FrameTri t0 = {}, t1 = {};
        if (in0) t0 = ftris[j0];
        if (in1) t1 = ftris[j1];
        const bool keep0 = in0 && !rect_culls(t0, rect), keep1 = in1 && !rect_culls(t1, rect);
        const unsigned long long m0 = __ballot(keep0), m1 = __ballot(keep1);
        const uint32_t c0 = (uint32_t)__popcll(m0);
        if (FILL && keep0) out[written + (uint32_t)__popcll(m0 & below)] = j0;  // within the counted size
        if (FILL && keep1) out[written + c0 + (uint32_t)__popcll(m1 & below)] = j1;
        written += c0 + (uint32_t)__popcll(m1);
    }
    if (!FILL && lane == 0u) counts4[bin * 4u + wave] = written;
}

// One workgroup: counts[b] = the bin's four wave counts, offsets[b] = sum of counts[0..b), *total = the sum; everything
// kBinNoList when it exceeds capacity.
__global__ void __launch_bounds__(1024)
k_bin_scan(const uint32_t *__restrict__ counts4, uint32_t *__restrict__ counts, uint32_t *__restrict__ offsets,
           uint32_t *__restrict__ total_out, uint32_t *__restrict__ total_host, uint32_t n_bins, uint32_t capacity)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    if (tid == 0) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < n_bins; base += 1024u) {
        const uint32_t b = base + tid;
        uint32_t c = 0u;
        if (b < n_bins) {   // a bin's count = its four waves' counts (k_bin_faces<false>)
            const uint4 c4 = reinterpret_cast<const uint4 *>(counts4)[b];
            c = c4.x + c4.y + c4.z + c4.w;
            counts[b] = c;
        }
        uint32_t incl = c;   // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if ((int)lane >= d) incl += up;
        }
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (uint32_t w = 0; w < wave; w++) before += s_wave[w];
        if (b < n_bins) offsets[b] = before + incl - c;
        __syncthreads();
        if (tid == 1023u) s_carry = before + incl;
        __syncthreads();
    }
    const uint32_t total = s_carry;
    if (tid == 0) {
        *total_out = total;
        if (total_host) *total_host = total;   // pinned host memory: the context reads it a frame late, never waits for it
    }
    if (total > capacity)   // uniform
        for (uint32_t b = tid; b < n_bins; b += 1024u) offsets[b] = kBinNoList;
}

hipError_t launch_bin_faces(hipStream_t s, const FrameTri *ftris, uint32_t n_tris, uint32_t row_begin, uint32_t *lists,
                            uint32_t *counts, uint32_t *offsets, uint32_t *total_out, uint32_t bins_x, uint32_t bins_y, uint32_t capacity,
                            const int32_t mesh_px[4], uint32_t *total_host)
{
    if (n_tris == 0 || bins_x == 0 || bins_y == 0) return hipSuccess;
    uint32_t *counts4 = counts;                      // the counts buffer: four per bin (one per wave, 16-byte groups) ...
    counts += 4u * (size_t)bins_x * bins_y;          // ... then the bins' own counts, which the render kernels read
    hipLaunchKernelGGL((k_bin_faces<false>), dim3(bins_x, bins_y), dim3(256), 0, s, ftris, n_tris, row_begin, lists, counts4, offsets, bins_x,
                       mesh_px[0], mesh_px[1], mesh_px[2], mesh_px[3]);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, counts4, counts, offsets, total_out, total_host, bins_x * bins_y, capacity);
    hipLaunchKernelGGL((k_bin_faces<true>), dim3(bins_x, bins_y), dim3(256), 0, s, ftris, n_tris, row_begin, lists, counts4, offsets, bins_x,
                       mesh_px[0], mesh_px[1], mesh_px[2], mesh_px[3]);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
template <bool AUX, bool CULL>
__global__ void __launch_bounds__(256, 8)  // 8 waves per SIMD: <= 64 VGPRs
k_primary(const FrameParams p, const TriRecord *__restrict__ tris, const ShadeRec *__restrict__ shade,
          const FrameTri *__restrict__ ftris, const float4 *__restrict__ tex,
          const Targets tg)
{
    __shared__ PrimaryShared s_prim;

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t blk_x0 = blockIdx.x * 32u;
    const uint32_t tile_x0 = blk_x0 + wave * 8u;
    const uint32_t tile_y0 = p.row_begin + blockIdx.y * p.row_pitch;
    const uint32_t px = tile_x0 + (lane & 7u), py = tile_y0 + (lane >> 3);
    const bool in_range = (px < p.width) && (py < p.row_end);

    const f3 O = ld3(p.cam.origin);
    const f3 D = pixel_to_ray_dir(p.cam, px, py, 0.5f, 0.5f, p.width, p.height);

    PrimaryHit r;
    uint32_t dbg_listed = 0, dbg_tested = 0;  // RWR_FLAG_DEBUG_COUNTS (aux builds only)
    primary_visibility<CULL, AUX>(p, tris, ftris, s_prim, blk_x0, tile_x0, tile_y0, O, D, r, dbg_listed, dbg_tested);

    // A pixel no pass wrote keeps the clear value (0,0,0,0); a written one gets
    // final_color with alpha 1 + 1 (compute.wgsl:231-234).
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
            const bool dbg = (p.flags & RWR_FLAG_DEBUG_COUNTS) != 0;
            tg.obj_id[o] = dbg ? (int32_t)dbg_listed : r.obj;
            tg.hit_t[o] = dbg ? (float)dbg_tested : r.t;
        }
    }
}

hipError_t launch_primary(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                          const FrameTri *ftris, const float4 *tex, const Targets &tg)
{
    if (fp.row_end <= fp.row_begin || fp.width == 0) return hipSuccess;
    const dim3 grid((fp.width + 31u) / 32u, band_strips(fp));
    const dim3 block(256);
    const bool aux = (fp.flags & RWR_FLAG_AUX_OUTPUTS) != 0;
    const bool do_cull = (fp.flags & RWR_FLAG_NO_CULL) == 0;
    if (aux && do_cull) hipLaunchKernelGGL((k_primary<true, true>), grid, block, 0, s, fp, tris, shade, ftris, tex, tg);
    else if (aux) hipLaunchKernelGGL((k_primary<true, false>), grid, block, 0, s, fp, tris, shade, ftris, tex, tg);
    else if (do_cull) hipLaunchKernelGGL((k_primary<false, true>), grid, block, 0, s, fp, tris, shade, ftris, tex, tg);
    else hipLaunchKernelGGL((k_primary<false, false>), grid, block, 0, s, fp, tris, shade, ftris, tex, tg);
    return hipGetLastError();
}

hipError_t preload_kernels_primary()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_frame_setup));
}

hipError_t preload_kernels()
{
    hipError_t e = preload_kernels_primary();
    if (e == hipSuccess) e = preload_kernels_primary_p2();
    if (e == hipSuccess) e = preload_kernels_wavefront();
    if (e == hipSuccess) e = preload_kernels_wf_primary();
    if (e == hipSuccess) e = preload_kernels_wf_bounce();
    if (e == hipSuccess) e = preload_kernels_dist();
    return e;
}

}  // namespace rwr
