// Internal declarations shared by the HIP kernels and the C-ABI implementation.
// Not part of the public boundary (include/rwr_hip.h is).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rwr_hip.h"

namespace rwr {

// Internal debug flag (not in the public header): with RWR_FLAG_AUX_OUTPUTS, the
// obj_id plane receives the number of block-level candidate faces and the hit_t
// plane the number of faces the pixel's wave ran the exact test on.
constexpr uint32_t RWR_FLAG_DEBUG_COUNTS = 1u << 16;
// Internal: render frames with the one-pixel-per-lane kernel (k_primary) instead of the
// two-pixels-per-lane one (k_primary_p2); both must give identical results.
constexpr uint32_t RWR_FLAG_ONE_PIXEL_PER_LANE = 1u << 17;

// ---------------------------------------------------------------------------
// Device-side scene records (data layout in HBM, see DESIGN.md §"Data layout").
//
// TriRecord: everything triangleRayIntersect (compute.wgsl:82-148) derives from
// the triangle alone, hoisted out of the per-ray loop.  The values are produced
// by k_prebake with exactly the operations the shader writes (no fusion), so
// hoisting does not change a single bit of the per-ray result.
// 128 B = four s_load_dwordx8 when the face index is wave-uniform.
struct alignas(16) TriRecord {
    float p0[3]; float d;      // d = -dot(N, p0)                      compute.wgsl:99
    float p1[3]; float denom;  // denom = dot(N, N)                    compute.wgsl:88
    float p2[3]; float pad0;
    float N[3];  float pad1;   // cross(p1 - p0, p2 - p0)              compute.wgsl:84-87
    float e0[3]; float pad2;   // p1 - p0                              compute.wgsl:115
    float e1[3]; float pad3;   // p2 - p1                              compute.wgsl:123
    float e2[3]; float pad4;   // p0 - p2                              compute.wgsl:132
    float nhat[3]; float pad5; // normalize(N), literal f32 operations (compute.wgsl:142 before the flip of :140):
                               // the extended integrator's bounce rays start from it (normalize(-N) == -normalize(N) exactly)
};
static_assert(sizeof(TriRecord) == 128, "TriRecord is 128 B");

// Per-face shading record (colour path only; 48 B = three dwordx4), made by k_prebake.
// Everything of triangle_list/compute.wgsl:217-234 that depends on the face alone is folded in
// double precision and rounded once: the unit normal, its Lambert term against the fixed
// directional light (:55), and the affine map from the winner's un-normalised edge functions
// (u, v) (:126,135) to texel space,  (x, y) = c0 + u*c1 + v*c2  with  x = tex_w * dot(barycentric,
// tex_coords.x) - 0.5  (:144-147,218-225), y likewise with the 1 - v flip of :224.  (x, y) pairs are
// adjacent so that the map is two v_pk_fma_f32.
struct alignas(16) ShadeRec {
    float n[3];  float ndl0;        // N / |N| (as wound, not flipped), dot(n, -normalize(kLightDir))
    float c0[2], c1[2];
    float c2[2]; uint32_t material; // index into MaterialRec[]
    float pad;
};
static_assert(sizeof(ShadeRec) == 48, "ShadeRec is 48 B");

// Largest texture edge accepted (wgpu's own default limit is 8192): keeps tap byte offsets in 32 bits.
constexpr uint32_t kMaxTextureDim = 16384;

// One material (MaterialData's ambient / specular, triangle_list.rs:24-33) with its decoded diffuse
// texture.  The reference binds materials[0] only (triangle_list.rs:212); more than one is an extension.
struct alignas(16) MaterialRec {
    float ambient[3];  uint32_t tex_w;
    float specular[3]; uint32_t tex_h;
    const float4 *tex;
    float wmax, hmax;               // (float)(tex_w - 1), (float)(tex_h - 1): the ClampToEdge bounds
    const float4 *nmap;             // optional normal map (extension, RWR_FLAG_NORMAL_MAP): linear texels, nullptr = none
    uint32_t nmap_w, nmap_h;
};
static_assert(sizeof(MaterialRec) == 64, "MaterialRec is 64 B");

// Per-face tangent frame for normal-mapped shading (extension; 32 B, made by k_prebake in double and rounded once):
// t = normalize(dP/du - n (n . dP/du)), b = +-cross(n, t) towards -dP/dv' (v' = 1 - v: the sampling space of
// compute.wgsl:224; the map's green axis points up the image); zeros for degenerate texture coordinates.
struct alignas(16) TangentRec {
    float t[3]; float pad0;
    float b[3]; float pad1;
};
static_assert(sizeof(TangentRec) == 32, "TangentRec is 32 B");

// The three corners again, packed (48 B): what the conservative tile/block
// frustum tests read, one record per lane, coalesced.
struct alignas(16) CullRec {
    float p0[3], p1[3], p2[3];
    float pad[3];
};
static_assert(sizeof(CullRec) == 48, "CullRec is 48 B");

struct Targets {
    uint8_t *color;     // W*H*4 rgba8unorm            (screen_texture, lib.rs:503-515)
    float *depth;       // W*H r32float                (depth_texture_output, lib.rs:482-495)
    float *color_f32;   // W*H*4, aux
    int32_t *obj_id;    // W*H, aux
    float *hit_t;       // W*H, aux
};

// Per-frame constants of the conservative culling code (host-computed in double
// from the camera uniform, context.cpp).  The un-normalised world direction of the
// ray through pixel-space point (fx, fy) is affine:  dir = A + fx*Bx + fy*By
// (compute.wgsl:151-159), so every plane through the origin that contains a
// pixel column x = const or row y = const has the normal  Ux + x*Vx  resp.
// Uy + y*Vy, and a point q = t*dir(x, y) has  Vx.q = t*vxa,  Vy.q = t*vya.
struct CullConsts {
    float A[4], Bx[4], By[4];
    float Ux[4], Vx[4];    // [3]: L1 norm of the xyz part
    float Uy[4], Vy[4];
    float origin[4];
    float vxa, vya;        // Vx.A, Vy.A  (non-zero for a pinhole camera)
    float corner_margin;   // kCullRel * max |dir|_1 over the frame
    uint32_t enabled;      // 0: degenerate camera, cull nothing
};

// Per-frame, per-face culling record written by k_frame_setup (64 B, one
// dwordx4 x4 per lane): the face's conservative pixel-space bounding rectangle
// and, for each of its edges, the affine function  d(x, y) = ea + x*ex + y*ey
// of dot(edge plane normal, dir(x, y)), signed so that the face can only be hit
// where all three are >= -me.
struct alignas(16) FrameTri {
    float bx0, by0, bx1, by1;
    float ea[3]; float me0;
    float ex[3]; float me1;
    float ey[3]; float me2;
};
static_assert(sizeof(FrameTri) == 64, "FrameTri is 64 B");

// Per-frame screen bins (kernels_primary.hip k_bin_faces): for scenes with more faces than one
// 256-wide batch, every 64x32-pixel bin (= 2x4 workgroups of the render kernels) gets the
// ascending list of faces that can be seen through it, so a workgroup walks its bin's list
// instead of the whole scene.  Built once per frame, shared by all sample passes.
constexpr uint32_t kBinW = 64, kBinH = 32;
struct BinGrid {
    const uint32_t *lists;    // the bins' face lists, back to back (sized by a count pass: count -> exclusive scan -> fill)
    const uint32_t *counts;   // bins
    const uint32_t *offsets;  // bins: where a bin's list starts; kBinNoList = "walk the whole scene" (the lists did not
                              // fit this frame's capacity; the context grows it for the next frames)
    uint32_t bins_x, bins_y, capacity, enabled;
};
constexpr uint32_t kBinNoList = 0xffffffffu;
constexpr uint32_t kStripRows = 8;   // rows of a strip = of every render kernel's workgroup tile (RWR_STRIP_ROWS)

struct FrameParams {
    rwr_camera_inv_uniform cam;
    BinGrid bins;
    uint32_t width, height;       // full frame
    uint32_t row_begin, row_end;  // band rendered by this launch ...
    uint32_t row_pitch;           // ... in 8-row strips: strip k of the launch starts at row_begin + k * row_pitch (8: every row of the
                                  // band; 8 N: every N-th strip — the interleaved partition of a multi-GPU frame, rwr_render_strips)
    uint32_t n_spheres;
    uint32_t n_tris;
    uint32_t tex_w, tex_h;
    float tex_wmax, tex_hmax;     // (float)(tex_w - 1), (float)(tex_h - 1) of material 0
    uint32_t flags;
    uint32_t wave_cull_min;   // run the per-wave (8x8 tile) cull only for block lists longer than this
    // wavefront integrator (extension): sample being traced, samples per pixel, RNG seed, bounces
    uint32_t sample, spp, seed, bounces;
    rwr_sphere_buffer_data spheres[RWR_MAX_SPHERES];
    // conservative pixel-space bounds {x0, y0, x1, y1} of each sphere's silhouette
    // (host-computed per frame, context.cpp); a tile outside them skips the sphere
    float sphere_rect[RWR_MAX_SPHERES][4];
    // conservative pixel-space bounds of the whole mesh's bounding box (+-inf when the camera is in or near it):
    // a tile outside skips the mesh pass
    float mesh_rect[4];
    int32_t mesh_px[4];   // the same, rounded outwards to whole pixels (scalar compares in the frame kernel)
    float ambient[4];
    float specular[4];
    // Per-frame tables written by k_frame_setup (the frame kernel with centre rays reads them; nothing
    // here changes a bit of the ray: the entries are the shader's own operations, hoisted because they
    // depend on the column, the row or the face alone):
    //   ray_colp[x / 2] (x even) = two float4 {c(x).x, c(x+1).x, c(x).y, c(x+1).y}, {c(x).z, c(x+1).z, -, -}
    //   with c(x) = proj_inv[0].xyz * x_nds(x),  x_nds = 2 * (x + 0.5) / width - 1     (compute.wgsl:151-155)
    //   ray_row[y] = {proj_inv[1].xyz * y_nds(y), y_nds}
    //   tnum[face] = -(dot(N, origin) + d)                                              (compute.wgsl:99-102)
    const float4 *ray_colp;
    const float4 *ray_row;
    const float *tnum;
    // multi-material scenes (n_materials > 1): per-face material through ShadeRec::material
    const MaterialRec *materials;
    uint32_t n_materials;
    uint32_t pad_m;
    const TangentRec *tangents;   // per face (RWR_FLAG_NORMAL_MAP)
};
// 8-row strips a launch renders (the y extent of every render kernel's grid)
inline uint32_t band_strips(const FrameParams &fp) { return fp.row_end > fp.row_begin ? (fp.row_end - fp.row_begin + fp.row_pitch - 1u) / fp.row_pitch : 0u; }


// context.cpp: records the calling thread's error message, returns `code`.
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// kernels_primary.hip
hipError_t launch_prebake(hipStream_t s, const rwr_model_vertex_small *verts, const rwr_model_face_small *faces,
                          const uint32_t *face_material, uint32_t n_faces, const rwr_instance_raw *instances,
                          uint32_t n_instances, const MaterialRec *materials, TriRecord *tris, ShadeRec *shade, CullRec *cull,
                          TangentRec *tangents);
// Wavefront integrator state (kernels_wf_primary.hip / kernels_wf_bounce.hip).  A launch group traces `group`
// samples of every pixel.  A *tile* is the 64x8-pixel block of one workgroup of the primary stage (4 waves of
// 32x4 pixels, two pixels per lane); its *pool* is the bounce rays those samples emit.  Ray queue = SoA in HBM,
// 32 B per bounce ray, at FIXED slots
//     slot = (tile * group + sample_in_group) * 512 + wave * 128 + k * 64 + lane        (k: which pixel of the lane)
// rays[2 slot] = {O.xyz, throughput r | g << 16}, rays[2 slot + 1] = {D.xyz, throughput b}: ONE 32-byte record per ray, because
// the bounce stage reads rays in sorted order, i.e. at random slots, and every separate access costs a whole memory
// sector (three separate arrays measured 4.5x the bytes the rays have); the throughput (the first hit's albedo, in
// [0, 1]) travels as unorm16, which moves a colour by at most 8e-6 of the second hit's radiance.  masks[(tile * group + s) * 8 + wave * 2 + k] = ballot of the
// lanes that emitted.  The bounce stage compacts a pool by those ballots while sorting it by direction.
constexpr uint32_t kWfTileW = 64, kWfTileH = 8, kWfTilePixels = kWfTileW * kWfTileH;
#ifndef RWR_WF_MAX_GROUP
#define RWR_WF_MAX_GROUP 64
#endif
constexpr uint32_t kWfMaxGroup = RWR_WF_MAX_GROUP;   // samples per launch group at most (the sort's LDS: 4 B per ray of a pool = 128 KiB at 64;
                                                     // the primary stage's 32-bit sums: kWfMaxGroup terms of at most kWfE0Cap)
constexpr float kWfFixedScale = 67108864.0f;  // 2^26: a term < 64, thousands of them < 2^64
// The integrator's definition (rwr_hip.h rwr_render_params; oracle render_path_core): a sample's E(h0) is clamped to [0, 16] per
// channel, its albedo * E(h1) to [0, 64] (= what a u32 of 2^-26 units holds: the float -> u32 conversion saturates).  Fixed
// numbers, not tuning parameters.
constexpr uint32_t kWfE0Cap = 16;
#ifndef RWR_WF_CELL_BITS
#define RWR_WF_CELL_BITS 4
#endif
constexpr uint32_t kWfDirCellBits = RWR_WF_CELL_BITS;                 // cells per side of an octant of the octahedral map, log2
constexpr uint32_t kWfDirBins = 8u << (2u * kWfDirCellBits);   // 8 octants x 16x16 cells (8x8: +6 % at configs[2] — coarser packets; the sort keeps its
                                                               // counts and offsets in ONE LDS array, or the finer histogram costs it its occupancy)
struct WfBuffers {
    unsigned long long *fix;       // 4 planes of W*H: the frame's radiance sums as 2^-26 FIXED POINT — red, green, blue of
                                   // E(h0) + albedo(h0) * E(h1) over all samples, and alpha (2 per primary hit).  Integer sums are
                                   // the same bits in whatever order and by whichever workgroups they are added.
    float4 *rays;                  // two float4 per slot: {O.xyz, thr.r | thr.g << 16}, {D.xyz, thr.b} (throughput as unorm16)
    unsigned long long *masks;
    uint16_t *bins;                // per slot: the ray's direction bin (wf_direction_bin; written with the ray, read by the sort)
    uint16_t *sorted;              // per tile: group * 512 pool slots in direction order (scratch of the bounce stage)
    uint32_t *wave_total;          // per tile and wave: bounce rays emitted over all groups of the frame
    uint32_t group;                // samples per launch group this frame
    uint32_t tiles_x;
    unsigned long long *dbg;       // optional (RWR_WF_STATS=1): {packet pools, their rays, per-lane pools, their rays}
    uint32_t *counters;            // this queue's four counters of the bounce stage (kernels_wf_bounce.hip); null without a bounce
    uint32_t shared_planes;        // another launch group may be adding to `fix` at the same time: atomics only
    // frames that show little (k_wf_classify ran): the tiles something can be seen through, how many, and per tile which of its
    // four waves' 32x4-pixel pieces; null otherwise (every tile is looked at)
    const uint32_t *live_list;
    const uint32_t *live_count;
    const uint32_t *tile_live;
};
struct BvhNode4;
struct BvhDevice {
    const BvhNode4 *nodes;
    const uint32_t *leaf_faces;
    uint32_t n_nodes;
    uint32_t stack_depth;  // 3 * tree depth + 2
    float packet_extent;   // pools whose ray origins span less than this are traced as packets (kernels_wf_bounce.hip)
    uint32_t min_packet_pools;  // fewer packet pools than this in a launch group: the per-lane kernel traces them
    uint32_t lane_items;        // work items the per-lane kernel's launch should have when pools are few (pool_split; tunable)
    uint32_t packet_dense_rays; // a pool of at least this many rays is traced as packets however far apart its rays start (0: never)
    uint32_t wide_lane;         // the per-lane trace kernel may run as 1 024-thread workgroups sharing one LDS copy of the nodelets
};

struct FusedSetup;
hipError_t launch_primary_p2(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                             const FrameTri *ftris, const float4 *tex, const Targets &tg, hipEvent_t ev_start = nullptr,
                             hipEvent_t ev_stop = nullptr, const FusedSetup *fused = nullptr);
// grid rows the fused form puts in front of the frame's strips for `n_blocks` record-making workgroups
uint32_t primary_p2_fused_rows(const FrameParams &fp, uint32_t n_blocks);
hipError_t launch_wf_primary(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                             const FrameTri *ftris, const float4 *tex, const Targets &tg,
                             const WfBuffers &wf, uint32_t sample_begin, uint32_t sample_count, uint32_t z_split);
// once per frame, ahead of the primary stage, when the frame is expected to show little: fills live_list / live_count / tile_live
hipError_t launch_wf_classify(hipStream_t s, const FrameParams &fp, const FrameTri *ftris, const Targets &tg, uint32_t tiles_x,
                              uint32_t *live_list, uint32_t *live_count, uint32_t *tile_live);
hipError_t launch_wf_bounce(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                            const BvhDevice &bvh, const float4 *tex, const WfBuffers &wf,
                            uint32_t n_tiles, uint32_t sample_count, uint32_t packet_min_rays, void *pool_info, uint32_t *pool_list);
size_t wf_pool_info_bytes();
hipError_t launch_primary_bvh(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                              const BvhDevice &bvh, const float4 *tex, const Targets &tg);
// live_counters / host_live: the last launch group's counters (kernels_wf_bounce.hip) and the pinned words the host reads them from
hipError_t launch_wf_resolve(hipStream_t s, const FrameParams &fp, const Targets &tg, const WfBuffers &wf, const uint32_t *live_counters = nullptr,
                             uint32_t *host_live = nullptr);

// per-frame records and tables (kernels_primary.hip): FrameTri + tnum per face, ray tables per column pair / row
struct FrameSetupOut {
    FrameTri *ftris;   // n_tris
    float *tnum;       // n_tris
    float4 *ray_colp;  // 2 * ray_pairs
    float4 *ray_row;   // ray_rows
    uint32_t ray_pairs, ray_rows;
    // words the frame wants zeroed before its first kernel (the wavefront integrator's per-tile ray counts and live-tile count):
    // done here instead of by memset commands of their own on the stream (4-5 us each on a frame of a few hundred)
    uint32_t *zero_a, *zero_b;
    uint32_t n_zero_a, n_zero_b;
};
// The frame kernel's FUSED form (one launch per frame, kernels_primary_p2.hip): the grid's first rows are workgroups that make
// the frame's records and tables (the work of k_frame_setup), the others wait until all of them have finished.
struct FusedSetup {
    CullConsts cc;
    const CullRec *cull;
    FrameSetupOut out;
    uint32_t n_blocks;     // record-making workgroups: blocks of k_frame_setup's grid
    uint32_t nb_tris;      // ... of which make face records
    uint32_t extra_rows;   // grid rows in front of the frame's own strips: ceil(n_blocks / grid.x)
    uint32_t flag_base;    // flag[0] before this frame (it counts finished record blocks: + n_blocks per frame, modulo 2^32)
    uint32_t *flag;        // [0] finished record blocks, [1] != 0: a wait ran out (the frame is incomplete; rwr_synchronize says so)
};

// kernels_dormant.hip: frames with single-triangle passes or orthographic rays (brute force, one pixel per lane)
struct SingleTriangles {  // the single-triangle passes of a frame (kept out of FrameParams: only this kernel reads them)
    uint32_t n;
    uint32_t pad[3];
    rwr_triangle_buffer_data t[RWR_MAX_TRIANGLES];
};
hipError_t launch_primary_dormant(hipStream_t s, const FrameParams &fp, const SingleTriangles &st, const TriRecord *tris,
                                  const ShadeRec *shade, const float4 *tex, const Targets &tg);

// kernels_*.hip: resolve one kernel of each translation unit (HIP loads a code object on first use)
hipError_t preload_kernels();
hipError_t preload_kernels_primary();
hipError_t preload_kernels_primary_p2();
hipError_t preload_kernels_wavefront();
hipError_t preload_kernels_wf_primary();
hipError_t preload_kernels_wf_bounce();
hipError_t preload_kernels_dist();

// kernels_dist.hip: the interleaved partition's gather (layout: rwr_strips.h) — every rank packs its strips into one message,
// the root deals the received messages out into the frame; one launch each.  The _host forms move host memory the same way.
struct StripLayout;
hipError_t launch_strips_pack(hipStream_t s, const StripLayout &L, uint32_t rank, uint32_t row_bytes, const uint8_t *frame, uint8_t *message);
hipError_t launch_strips_deal(hipStream_t s, const StripLayout &L, uint32_t row_bytes, const uint8_t *recv, uint8_t *frame);
void strips_pack_host(const StripLayout &L, uint32_t rank, size_t row_bytes, const uint8_t *frame, uint8_t *message);
void strips_deal_host(const StripLayout &L, size_t row_bytes, const uint8_t *recv, uint8_t *frame);

// kernels_selftest.hip: out[0..3] += depth inputs compared, mismatches, normalize inputs compared, mismatches
hipError_t launch_selftest_exact_math(hipStream_t s, unsigned long long *d_out4, uint32_t normalize_count, uint32_t seed);

// kernels_selftest.hip: d_out[wave] = {shader cycles, 100 MHz ticks} around iters * 8 v_fma_f32 (mode 0) / v_pk_fma_f32 (mode 1)
hipError_t launch_clock_probe(hipStream_t s, ulonglong2 *d_out, uint32_t ticks_100mhz);
hipError_t launch_measure_valu(hipStream_t s, int mode, ulonglong2 *d_out, uint32_t n_workgroups, uint32_t iters);

hipError_t launch_frame_setup(hipStream_t s, const CullConsts &cc, const rwr_camera_inv_uniform &cam, uint32_t width,
                              uint32_t height, const CullRec *cull, const TriRecord *tris, uint32_t n_tris,
                              const FrameSetupOut &out);
// count -> scan -> fill; *total_out (device) receives the entries the frame's lists need
hipError_t launch_bin_faces(hipStream_t s, const FrameTri *ftris, uint32_t n_tris, uint32_t row_begin, uint32_t *lists,
                            uint32_t *counts, uint32_t *offsets, uint32_t *total_out, uint32_t bins_x, uint32_t bins_y, uint32_t capacity,
                            const int32_t mesh_px[4], uint32_t *total_host);
hipError_t launch_primary(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                          const FrameTri *ftris, const float4 *tex, const Targets &tg);

}  // namespace rwr
