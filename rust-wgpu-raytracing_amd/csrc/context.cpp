// C-ABI implementation: context, scene upload, frame orchestration.
// Replaces the wgpu plumbing of State (/root/reference/src/lib.rs:260-1231) with
// one HIP stream and a handful of device buffers.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the library is bound at run time (rwr_dist_init), never at link time

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "bvh.hpp"
#include "rwr_internal.h"
#include "rwr_strips.h"

namespace rwr {

static thread_local std::string g_last_error;

int set_error(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define RWR_HIP_CHECK(expr)                                                                          \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return set_error(RWR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

template <typename T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t count = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= count && ptr) return hipSuccess;
        release();
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T));
        if (e == hipSuccess) count = n;
        else ptr = nullptr;
        return e;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
};

}  // namespace rwr

using namespace rwr;

// Everything one frame in flight owns: its stream, its targets and the per-frame records / tables.
// Frames alternate between slots (rwr_ctx_set_frames_in_flight), so the ramp-up of one frame's kernel
// fills the machine while the previous frame's last waves drain; a slot is reused in stream order.
struct FrameSlot {
    hipStream_t stream = nullptr;   // slot 0: the context's stream (may be the caller's); others: owned
    hipStream_t owned = nullptr;
    hipEvent_t done = nullptr;      // rwr_timer_end: joins the slot into the timing stream
    DeviceBuffer<uint8_t> d_color;
    DeviceBuffer<float> d_depth;
    DeviceBuffer<float> d_color_f32;
    DeviceBuffer<int32_t> d_obj_id;
    DeviceBuffer<float> d_hit_t;
    DeviceBuffer<FrameTri> d_ftris;
    DeviceBuffer<float> d_tnum;                  // per frame: plane-distance numerator per face
    DeviceBuffer<float4> d_ray_colp, d_ray_row;  // per frame: ray tables (FrameParams::ray_colp / ray_row)
    DeviceBuffer<uint32_t> d_bin_lists, d_bin_counts, d_bin_offsets, d_bin_total;   // per-frame screen bins (large scenes)
    uint32_t *h_bin_total = nullptr;   // pinned: entries the last binned frame of this slot needed (read a frame late, never waited for)
    bool aux_valid = false;
    // RWR_FRAME_GRAPH (A/B knob): the reference frame's two launches (k_frame_setup -> k_primary_p2) as a
    // hipGraph of this slot — replayed as it is while camera and parameters stay the same, updated in place when they change
    hipGraphExec_t frame_graph = nullptr;
    std::vector<unsigned char> frame_graph_key;
    // the frame kernel's fused form (one launch per frame): {finished record blocks, "a wait ran out"} on the device, the count
    // the host expects before the next frame, and the block count it is valid for
    DeviceBuffer<uint32_t> d_fused;
    uint32_t fused_count = 0, fused_blocks = 0;
    bool fused_used = false;
    void release_buffers()
    {
        if (frame_graph) { (void)hipGraphExecDestroy(frame_graph); frame_graph = nullptr; }
        frame_graph_key.clear();
        d_fused.release(); fused_count = 0; fused_blocks = 0; fused_used = false;
        d_color.release(); d_depth.release(); d_color_f32.release(); d_obj_id.release(); d_hit_t.release();
        d_ftris.release(); d_tnum.release(); d_ray_colp.release(); d_ray_row.release();
        d_bin_lists.release(); d_bin_counts.release(); d_bin_offsets.release(); d_bin_total.release();
        if (h_bin_total) { (void)hipHostFree(h_bin_total); h_bin_total = nullptr; }
    }
};
constexpr uint32_t kMaxFramesInFlight = 3;

struct rwr_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;   // == slots[0].stream
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    FrameSlot slots[kMaxFramesInFlight];
    uint32_t n_slots = 1;           // frames in flight
    uint32_t cur = 0;               // slot of the most recent frame

    // scene
    DeviceBuffer<rwr_model_vertex_small> d_verts;
    DeviceBuffer<rwr_model_face_small> d_faces;
    DeviceBuffer<rwr_instance_raw> d_instances;
    DeviceBuffer<TriRecord> d_tris;
    DeviceBuffer<ShadeRec> d_shade;
    DeviceBuffer<CullRec> d_cull;
    DeviceBuffer<TangentRec> d_tangent;                 // per-face tangent frames (normal-mapped shading)
    std::vector<DeviceBuffer<float4>> d_nmaps;          // one optional normal map per scene part (linear texels)
    uint32_t bin_min_faces = 256;                       // tunable: RWR_BIN_MIN_FACES
    uint32_t bin_min_capacity = 65536;                  // tunable: RWR_BIN_CAPACITY (entries the bin lists start with)
    bool force_one_pixel = false;                       // debug: RWR_ONE_PIXEL_PER_LANE=1
    uint32_t frame_graph_mode = 0;                      // A/B: RWR_FRAME_GRAPH=1 (hipGraph replay / update of the reference frame's launches)
    bool fused_setup = true;                            // one launch per small reference frame (k_primary_p2<FUSED>); RWR_FUSED_SETUP=0: two
    bool fused_setup_force = false;                     // RWR_FUSED_SETUP=1: wherever the fused form is possible
    // BVH over the (flattened) world-space faces, for bounce rays
    DeviceBuffer<BvhNode4> d_bvh_nodes;
    DeviceBuffer<uint32_t> d_bvh_leaf_faces;
    uint32_t bvh_n_nodes = 0, bvh_depth = 0;
    float bvh_leaf_extent = 0.0f;
    float wf_packet_extent = 0.5f;   // x mean leaf extent; tunable: RWR_WF_PACKET_EXTENT
    uint32_t wf_min_packet_pools = 64;    // tunable: RWR_WF_MIN_PACKET_POOLS (a quarter share of configs[4]'s frame has about 100 packet pools of 30 000 rays
                                          // and is 5 % faster with them as packets, an eighth share has 50 and is 10 % faster per lane: tools/share_probe.py)
    uint32_t wf_lane_items = 0;           // tunable: RWR_WF_LANE_ITEMS (0: chosen per frame, see the BvhDevice of the wavefront path)
    int32_t wf_wide_lane = -1;            // the per-lane trace kernel as 1 024-thread workgroups: -1 by itself (below), tunable: RWR_WF_WIDE_LANE=0/1
    uint32_t wf_packet_dense_rays = 16384;   // a pool of at least this many rays (32 samples of a full tile) is traced as packets
                                             // however far apart its rays start; tunable: RWR_WF_PACKET_RAYS (0: never).  Measured
                                             // (tools/packet_rays_sweep.sh): configs[4]'s frame, 64 samples per group, 2.11 -> 1.75
                                             // ms with any threshold from 2 000 to 24 000 (2.30 -> 2.18 one frame at a time at
                                             // 16 000); configs[3], 16 samples (pools of at most 8 192): 0.604 -> 0.71-0.77 ms
                                             // with thresholds up to 8 000, unchanged from 12 000
    float aabb_lo[3] = {0, 0, 0}, aabb_hi[3] = {0, 0, 0};   // of the (flattened) world-space faces
    float auto_bvh_face_px = 150.0f;   // tunable: RWR_AUTO_BVH_FACE_PX (0 = never pick the BVH kernel by itself)
    // wavefront integrator: tunables and what the host remembers of the last frame
    uint32_t *h_wf_live = nullptr;  // pinned: live pools of the last launch group {packets, per-lane}, read a frame late
    uint32_t wf_z_split = 0;        // tunable: RWR_WF_ZSPLIT (0 = from the previous frame's live pools)
    DeviceBuffer<unsigned long long> d_wf_dbg;   // RWR_WF_STATS=1: pool classification counters, printed at destroy
    uint32_t wf_group = 0;          // samples per launch group; tunable: RWR_WF_GROUP (1..64); 0: 32 for a context that renders one
                                    // frame at a time (a 64-spp frame's two groups overlap each other on two queues), 64 with frames in
                                    // flight (larger pools sort into tighter packets; the overlap comes from the other frame): measured
                                    // at configs[2], two slots: 8.55 -> 8.21 ms per frame; one slot: 8.81 -> 8.98
    float wf_packet_fill = 0.25f;   // pools filled at least this much are traced as packets; tunable: RWR_WF_PACKET_FILL (> 1: never)
    uint32_t last_segments = 0;     // tiles of the last wavefront frame
    uint32_t last_wf_state = 0;     // ... and whose accumulators and queues it used
    static constexpr uint32_t kWfMaxQueues = 4;
    uint32_t wf_queues = 2;         // tunable: RWR_WF_OVERLAP (1 puts every launch group on the frame's stream; measured at
                                    // configs[2] / [4]: two queues -6.5 % / -9.5 %, three and four less, a staggered start less)
    // The integrator's device state, one set per frame slot: a frame of the integrator then shares nothing with the frames in
    // the other slots (they overlap like reference frames do), and reuses its own set in stream order.
    struct WfState {
        DeviceBuffer<float4> d_rays;
        bool fix_clean = false;         // the fixed-point planes are all zero (k_wf_resolve leaves them so)
        DeviceBuffer<unsigned long long> d_masks;
        DeviceBuffer<uint16_t> d_sorted, d_bins;
        DeviceBuffer<uint32_t> d_wave_total;
        DeviceBuffer<unsigned long long> d_fix;   // the frame's fixed-point sums, 4 planes
        DeviceBuffer<uint8_t> d_pool_info;
        DeviceBuffer<uint32_t> d_live;            // device counters of the bounce stage, a set of four per ray queue
        DeviceBuffer<uint32_t> d_pool_list;       // live pools by class, 2 x tiles
        DeviceBuffer<uint32_t> d_tiles;           // frames that show little: live tile list, per-tile live pieces, the count (k_wf_classify)
        // Launch groups alternate between the frame's stream and these, each with its own part of the ray queue: the
        // latency-bound ends of one group (the sort, the last packets) run beside the other group's arithmetic.
        hipStream_t streams[kWfMaxQueues] = {};   // [0] unused: queue 0 runs on the frame's stream
        hipEvent_t fork = nullptr, join[kWfMaxQueues] = {};
        void release()
        {
            for (uint32_t q = 0; q < kWfMaxQueues; q++) {
                if (streams[q]) { (void)hipStreamSynchronize(streams[q]); (void)hipStreamDestroy(streams[q]); streams[q] = nullptr; }
                if (join[q]) { (void)hipEventDestroy(join[q]); join[q] = nullptr; }
            }
            if (fork) { (void)hipEventDestroy(fork); fork = nullptr; }
            d_rays.release(); d_masks.release(); d_sorted.release(); d_bins.release(); d_wave_total.release(); d_fix.release();
            d_pool_info.release(); d_live.release(); d_pool_list.release(); d_tiles.release();
            fix_clean = false;
        }
    } wf_state[kMaxFramesInFlight];
    uint32_t last_spp = 0;
    bool last_had_bounce = false;
    // one decoded texture per scene part (texels decoded to linear f32 at upload, Rgba8UnormSrgb semantics)
    std::vector<DeviceBuffer<float4>> d_texs;
    DeviceBuffer<uint32_t> d_face_mat;      // per face: index of its part's material
    DeviceBuffer<MaterialRec> d_materials;
    // host staging of the scene being assembled (rwr_scene_clear / add_mesh / commit)
    std::vector<rwr_model_vertex_small> st_verts;
    std::vector<rwr_model_face_small> st_faces;
    std::vector<uint32_t> st_face_mat;
    std::vector<MaterialRec> st_materials;
    uint32_t n_verts = 0, n_faces = 0, n_instances = 0, n_tris = 0;
    uint32_t tex_w = 0, tex_h = 0;
    rwr_material_data material{};
    bool have_mesh = false;
    bool tris_dirty = false;
    rwr_sphere_buffer_data spheres[RWR_MAX_SPHERES]{};
    uint32_t n_spheres = 0;
    rwr_triangle_buffer_data triangles[RWR_MAX_TRIANGLES]{};
    uint32_t n_triangles = 0;

    // targets
    rwr_screen screen{0, 0};

    uint64_t last_primary = 0, last_bounce = 0;
    // optional per-kernel timing (rwr_ctx_set_kernel_timing)
    uint32_t timing_every = 0;
    uint64_t timing_calls = 0;
    std::vector<hipEvent_t> timing_events;  // pairs
    uint32_t timing_pairs = 0;
    uint32_t wave_cull_min = 4;  // tunable: RWR_WAVE_CULL_MIN
    // multi-GPU frame (rwr_dist_*): one process per GPU, one RCCL communicator per context
    void *rccl_lib = nullptr;
    ncclComm_t comm = nullptr;
    int dist_rank = 0, dist_world = 0;
    // one gather set per frame slot: the gather of the frame in one slot shares nothing with the frame rendered next in another
    struct GatherSet {
        DeviceBuffer<uint8_t> d_gathered;       // root: the assembled RGBA8 frame
        DeviceBuffer<uint8_t> d_pack, d_recv;   // interleaved partition: this rank's message; root: every rank's, side by side (rwr_strips.h)
        hipEvent_t done = nullptr;              // the set's last gather has finished
        bool valid = false;                     // d_gathered holds (or will hold, once `done`) a whole frame
        void release()
        {
            d_gathered.release(); d_pack.release(); d_recv.release();
            if (done) { (void)hipEventDestroy(done); done = nullptr; }
            valid = false;
        }
    } gather[kMaxFramesInFlight];
    uint32_t last_gather = 0;               // the set rwr_dist_frame / rwr_dist_readback refer to
    hipEvent_t exchange_done = nullptr;     // orders the RCCL exchanges of consecutive frames (they run on different slots' streams)
    // shader-clock probe (rwr_clock_probe_start / _read): one spinning wave on its own stream
    hipStream_t probe_stream = nullptr;
    DeviceBuffer<ulonglong2> d_probe;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Waits for every frame in flight (scene changes, resizes, stream changes and teardown need an idle context).
hipError_t sync_all(rwr_context *ctx)
{
    hipError_t first = hipSuccess;
    for (uint32_t i = 0; i < kMaxFramesInFlight; i++) {
        if (!ctx->slots[i].stream) continue;
        const hipError_t e = hipStreamSynchronize(ctx->slots[i].stream);
        if (first == hipSuccess) first = e;
    }
    return first;
}

// Targets of slot `i` for the current screen (the reference's textures start zeroed and are cleared every frame).
hipError_t ensure_slot_targets(rwr_context *ctx, uint32_t i)
{
    FrameSlot &sl = ctx->slots[i];
    const size_t n = (size_t)ctx->screen.width * ctx->screen.height;
    if (n == 0) return hipSuccess;
    hipError_t e;
    if ((e = sl.d_color.ensure(n * 4)) != hipSuccess || (e = sl.d_depth.ensure(n)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(sl.d_color.ptr, 0, n * 4, sl.stream)) != hipSuccess) return e;
    return hipMemsetAsync(sl.d_depth.ptr, 0, n * sizeof(float), sl.stream);
}

// Per-frame buffers of every active slot for the current scene and screen, so that the first frame
// does not pay for allocations (the render path re-checks: both calls are no-ops once sized).
hipError_t ensure_frame_buffers(rwr_context *ctx)
{
    const uint32_t total = ctx->n_faces * (ctx->n_instances ? ctx->n_instances : 1u);
    for (uint32_t i = 0; i < ctx->n_slots; i++) {
        FrameSlot &sl = ctx->slots[i];
        hipError_t e;
        if (total && ((e = sl.d_ftris.ensure(total)) != hipSuccess || (e = sl.d_tnum.ensure(total)) != hipSuccess)) return e;
        if (ctx->screen.width) {
            if ((e = sl.d_ray_colp.ensure(2u * (size_t)(((ctx->screen.width + 63u) / 64u) * 32u))) != hipSuccess) return e;
            if ((e = sl.d_ray_row.ensure(ctx->screen.height + 8u)) != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

void build_srgb_lut(float *lut)
{
    // Rgba8UnormSrgb decode (texture.rs:122): the sRGB EOTF, evaluated in double.
    for (int i = 0; i < 256; i++) {
        const double c = (double)i / 255.0;
        const double l = (c <= 0.04045) ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
        lut[i] = (float)l;
    }
}

// Per-frame culling constants (rwr_internal.h CullConsts), evaluated in double.
void compute_cull_consts(const rwr_camera_inv_uniform &cam, uint32_t width, uint32_t height, CullConsts &cc)
{
    auto dir = [&](double fx, double fy, double out[3]) {
        const double xn = 2.0 * fx / (double)width - 1.0, yn = 2.0 * fy / (double)height - 1.0;
        double v[4];
        for (int r = 0; r < 4; r++)
            v[r] = cam.proj_inv[0][r] * xn + cam.proj_inv[1][r] * yn + cam.proj_inv[2][r] + cam.proj_inv[3][r];
        for (int r = 0; r < 3; r++)
            out[r] = cam.viewmodel_inv[0][r] * v[0] + cam.viewmodel_inv[1][r] * v[1] + cam.viewmodel_inv[2][r] * v[2];
    };
    auto cross = [](const double a[3], const double b[3], double o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    };
    double A[3], Px[3], Py[3], Bx[3], By[3];
    dir(0.0, 0.0, A);
    dir((double)width, 0.0, Px);
    dir(0.0, (double)height, Py);
    for (int k = 0; k < 3; k++) { Bx[k] = (Px[k] - A[k]) / (double)width; By[k] = (Py[k] - A[k]) / (double)height; }
    double Ux[3], Vx[3], Uy[3], Vy[3];
    cross(A, By, Ux); cross(Bx, By, Vx);
    cross(A, Bx, Uy); cross(By, Bx, Vy);
    const double detx = Ux[0] * Bx[0] + Ux[1] * Bx[1] + Ux[2] * Bx[2];  // n_x . Bx
    const double dety = Uy[0] * By[0] + Uy[1] * By[1] + Uy[2] * By[2];  // n_y . By
    const double sx = detx < 0.0 ? -1.0 : 1.0, sy = dety < 0.0 ? -1.0 : 1.0;
    std::memset(&cc, 0, sizeof cc);
    double l1[4] = {0, 0, 0, 0};
    for (int k = 0; k < 3; k++) {
        cc.A[k] = (float)A[k]; cc.Bx[k] = (float)Bx[k]; cc.By[k] = (float)By[k];
        cc.Ux[k] = (float)(sx * Ux[k]); cc.Vx[k] = (float)(sx * Vx[k]);
        cc.Uy[k] = (float)(sy * Uy[k]); cc.Vy[k] = (float)(sy * Vy[k]);
        l1[0] += std::fabs(Ux[k]); l1[1] += std::fabs(Vx[k]); l1[2] += std::fabs(Uy[k]); l1[3] += std::fabs(Vy[k]);
    }
    cc.Ux[3] = (float)l1[0]; cc.Vx[3] = (float)l1[1]; cc.Uy[3] = (float)l1[2]; cc.Vy[3] = (float)l1[3];
    for (int k = 0; k < 3; k++) cc.origin[k] = cam.origin[k];
    // the largest |dir|_1 (dir is affine, so |dir|_1 peaks at a screen corner)
    double corner[4][3];
    dir(0.0, 0.0, corner[0]); dir((double)width, 0.0, corner[1]);
    dir((double)width, (double)height, corner[2]); dir(0.0, (double)height, corner[3]);
    double max_dir_l1 = 0.0;
    for (int c = 0; c < 4; c++)
        max_dir_l1 = std::fmax(max_dir_l1, std::fabs(corner[c][0]) + std::fabs(corner[c][1]) + std::fabs(corner[c][2]));
    const double k_rel = 2e-5;  // kCullRel (rwr_cull.h)
    cc.corner_margin = (float)(k_rel * 1.001 * max_dir_l1);
    double vxa = 0.0, vya = 0.0;
    for (int k = 0; k < 3; k++) { vxa += sx * Vx[k] * A[k]; vya += sy * Vy[k] * A[k]; }
    cc.vxa = (float)vxa; cc.vya = (float)vya;
    // a pinhole camera has both determinants well away from 0; a singular or non-finite
    // uniform simply disables culling (the exact test then sees every face)
    const double bx1 = std::fabs(Bx[0]) + std::fabs(Bx[1]) + std::fabs(Bx[2]);
    const double by1 = std::fabs(By[0]) + std::fabs(By[1]) + std::fabs(By[2]);
    const bool ok = std::isfinite(detx) && std::isfinite(dety) && std::isfinite(max_dir_l1) &&
                    std::fabs(detx) > 1e-9 * l1[0] * bx1 && std::fabs(dety) > 1e-9 * l1[2] * by1;
    cc.enabled = (ok && cc.vxa != 0.0f && cc.vya != 0.0f) ? 1u : 0u;
}

// Screen rectangle {x0, y0, x1, y1} (pixel coordinates, un-clipped) of the bounding box of the whole mesh as
// this camera sees it; false when any box corner is behind (or beside) the camera plane — the camera is in or
// near the mesh — or when culling is off.  Conservative use needs the caller's margin.
bool mesh_screen_rect(const CullConsts &cc, const float lo[3], const float hi[3], double rect[4])
{
    if (!cc.enabled) return false;
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (int c = 0; c < 8; c++) {
        double q[3], vx = 0.0, vy = 0.0, ux = 0.0, uy = 0.0;
        for (int k = 0; k < 3; k++) {
            q[k] = (double)((c >> k & 1) ? hi[k] : lo[k]) - (double)cc.origin[k];
            vx += cc.Vx[k] * q[k]; vy += cc.Vy[k] * q[k]; ux += cc.Ux[k] * q[k]; uy += cc.Uy[k] * q[k];
        }
        // depth along the view direction of this corner (see compute_sphere_rects); must be clearly in front
        if (!(vx / cc.vxa > 1e-6) || !(vy / cc.vya > 1e-6)) return false;
        const double x = -ux / vx, y = -uy / vy;
        if (!std::isfinite(x) || !std::isfinite(y)) return false;
        x0 = std::fmin(x0, x); x1 = std::fmax(x1, x); y0 = std::fmin(y0, y); y1 = std::fmax(y1, y);
    }
    rect[0] = x0; rect[1] = y0; rect[2] = x1; rect[3] = y1;
    return true;
}

// Average projected area, in pixels, of a face of the mesh: the area of that rectangle (clipped to the
// frame) over half the face count; +inf when there is no rectangle.  The frame kernel walks, per 32x4-pixel
// tile, every face that may touch the tile, one after the other; when faces are much smaller than a tile (a
// distant or finely tessellated mesh) the per-ray BVH traversal of k_primary_bvh is faster
// (tools/dense_probe.py: cube.obj, 428 faces, from 8 units away and beyond — up to 2x) and gives the same
// frame bit for bit, so the context switches to it for binned scenes (more than 256 faces; a smaller mesh
// bounds the walk by itself).
double mean_face_pixels(bool have_rect, const double rect[4], uint32_t n_tris, uint32_t width, uint32_t height)
{
    if (!have_rect || n_tris == 0) return INFINITY;
    const double x0 = std::fmax(rect[0], 0.0), y0 = std::fmax(rect[1], 0.0);
    const double x1 = std::fmin(rect[2], (double)width), y1 = std::fmin(rect[3], (double)height);
    if (!(x1 > x0) || !(y1 > y0)) return INFINITY;   // off screen: nothing to trace either way
    return (x1 - x0) * (y1 - y0) / (0.5 * (double)n_tris);
}

// Conservative pixel-space bounds of each analytic sphere's silhouette, so that
// tiles which cannot see a sphere skip its intersection test (the skipped test
// would have returned "no hit").  The sphere touches pixel column x iff its
// centre q (relative to the ray origin) is within r of the plane with normal
// n(x) = Ux + x*Vx:  (n(x).q)^2 <= r^2 |n(x)|^2, a quadratic in x.  Anything
// unusual (origin inside the sphere, sphere straddling the camera plane,
// non-finite numbers) yields "whole screen".
void compute_sphere_rects(const CullConsts &cc, const rwr_sphere_buffer_data *spheres, uint32_t n, uint32_t width,
                          uint32_t height, float (*rects)[4])
{
    const float inf = HUGE_VALF;
    for (uint32_t s = 0; s < RWR_MAX_SPHERES; s++) { rects[s][0] = -inf; rects[s][1] = -inf; rects[s][2] = inf; rects[s][3] = inf; }
    if (!cc.enabled) return;
    auto interval = [](const float *U, const float *V, const double q[3], double r, double &lo, double &hi) -> bool {
        double uq = 0, vq = 0, uu = 0, uv = 0, vv = 0;
        for (int k = 0; k < 3; k++) { uq += U[k] * q[k]; vq += V[k] * q[k]; uu += (double)U[k] * U[k]; uv += (double)U[k] * V[k]; vv += (double)V[k] * V[k]; }
        const double a = vq * vq - r * r * vv, b = uq * vq - r * r * uv, c = uq * uq - r * r * uu;  // a x^2 + 2 b x + c <= 0
        const double disc = b * b - a * c;
        if (!(a > 0.0) || !(disc >= 0.0)) return false;
        const double sq = std::sqrt(disc);
        lo = (-b - sq) / a;
        hi = (-b + sq) / a;
        return std::isfinite(lo) && std::isfinite(hi);
    };
    for (uint32_t s = 0; s < n; s++) {
        const double r = std::fabs((double)spheres[s].radius);
        double q[3], qq = 0.0, vq = 0.0, vv = 0.0;
        for (int k = 0; k < 3; k++) {
            q[k] = (double)spheres[s].center[k] - (double)cc.origin[k];
            qq += q[k] * q[k]; vq += cc.Vx[k] * q[k]; vv += (double)cc.Vx[k] * cc.Vx[k];
        }
        if (!(qq > r * r * 1.0001)) continue;  // origin inside (or on) the sphere: every ray may hit
        // depth coordinate t of a point p: (Vx.p)/vxa; over the sphere it spans t_c -+ r|Vx|/|vxa|
        const double tc = vq / cc.vxa, tr = r * std::sqrt(vv) / std::fabs((double)cc.vxa);
        if (tc + tr < 0.0) { rects[s][0] = inf; rects[s][1] = inf; rects[s][2] = -inf; rects[s][3] = -inf; continue; }  // behind
        if (!(tc - tr > 0.0)) continue;  // straddles the camera plane
        double x0, x1, y0, y1;
        if (!interval(cc.Ux, cc.Vx, q, r, x0, x1) || !interval(cc.Uy, cc.Vy, q, r, y0, y1)) continue;
        rects[s][0] = (float)(x0 - 0.5 - 1e-4 * std::fabs(x0)); rects[s][2] = (float)(x1 + 0.5 + 1e-4 * std::fabs(x1));
        rects[s][1] = (float)(y0 - 0.5 - 1e-4 * std::fabs(y0)); rects[s][3] = (float)(y1 + 0.5 + 1e-4 * std::fabs(y1));
    }
    (void)width; (void)height;
}

int rebuild_tris(rwr_context *ctx)
{
    if (!ctx->tris_dirty) return RWR_OK;
    const uint32_t total = ctx->n_faces * (ctx->n_instances ? ctx->n_instances : 1u);
    RWR_HIP_CHECK(ctx->d_tris.ensure(total));
    RWR_HIP_CHECK(ctx->d_shade.ensure(total));
    RWR_HIP_CHECK(ctx->d_cull.ensure(total));
    RWR_HIP_CHECK(ctx->d_tangent.ensure(total));
    RWR_HIP_CHECK(launch_prebake(ctx->stream, ctx->d_verts.ptr, ctx->d_faces.ptr, ctx->d_face_mat.ptr, ctx->n_faces, ctx->d_instances.ptr,
                                 ctx->n_instances, ctx->d_materials.ptr, ctx->d_tris.ptr, ctx->d_shade.ptr, ctx->d_cull.ptr,
                                 ctx->d_tangent.ptr));
    // BVH for incoherent rays, built on the host from the device's own world-space corners
    // (so instancing arithmetic happens in exactly one place, k_prebake)
    std::vector<CullRec> host_cull(total);
    RWR_HIP_CHECK(hipMemcpyAsync(host_cull.data(), ctx->d_cull.ptr, (size_t)total * sizeof(CullRec), hipMemcpyDeviceToHost, ctx->stream));
    RWR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    std::vector<float> corners((size_t)total * 9);
    for (uint32_t i = 0; i < total; i++) {
        std::memcpy(&corners[9 * (size_t)i + 0], host_cull[i].p0, 12);
        std::memcpy(&corners[9 * (size_t)i + 3], host_cull[i].p1, 12);
        std::memcpy(&corners[9 * (size_t)i + 6], host_cull[i].p2, 12);
    }
    for (int k = 0; k < 3; k++) { ctx->aabb_lo[k] = INFINITY; ctx->aabb_hi[k] = -INFINITY; }
    for (size_t v = 0; v < (size_t)total * 3; v++)
        for (int k = 0; k < 3; k++) {
            ctx->aabb_lo[k] = std::fmin(ctx->aabb_lo[k], corners[3 * v + k]);
            ctx->aabb_hi[k] = std::fmax(ctx->aabb_hi[k], corners[3 * v + k]);
        }
    uint32_t max_leaf = kBvhMaxLeafDefault;
    if (const char *e = std::getenv("RWR_BVH_LEAF")) max_leaf = (uint32_t)std::strtoul(e, nullptr, 10);  // tuning knob
    const Bvh bvh = build_bvh(corners.data(), total, max_leaf);
    if (bvh.max_depth > kBvhMaxDepth)
        return set_error(RWR_ERR_UNSUPPORTED, "the scene's BVH is %u levels deep (limit %u): too many faces for the traversal stacks",
                         bvh.max_depth, kBvhMaxDepth);
    RWR_HIP_CHECK(ctx->d_bvh_nodes.ensure(bvh.nodes.size()));
    RWR_HIP_CHECK(ctx->d_bvh_leaf_faces.ensure(bvh.leaf_faces.size() ? bvh.leaf_faces.size() : 1));
    RWR_HIP_CHECK(hipMemcpy(ctx->d_bvh_nodes.ptr, bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode4), hipMemcpyHostToDevice));
    if (!bvh.leaf_faces.empty())
        RWR_HIP_CHECK(hipMemcpy(ctx->d_bvh_leaf_faces.ptr, bvh.leaf_faces.data(), bvh.leaf_faces.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    ctx->bvh_n_nodes = (uint32_t)bvh.nodes.size();
    ctx->bvh_depth = bvh.max_depth;
    ctx->bvh_leaf_extent = bvh.mean_leaf_extent;
    ctx->n_tris = total;
    ctx->tris_dirty = false;
    return RWR_OK;
}

}  // namespace

extern "C" {

const char *rwr_last_error_string(void) { return g_last_error.c_str(); }

int rwr_device_count(int *out_count)
{
    if (!out_count) return set_error(RWR_ERR_INVALID_ARGUMENT, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        return set_error(RWR_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *out_count = n;
    return RWR_OK;
}

int rwr_ctx_create(int device_id, rwr_context **out_ctx)
{
    if (!out_ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return set_error(RWR_ERR_HIP, "no HIP device available (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return set_error(RWR_ERR_INVALID_ARGUMENT, "device_id %d out of range [0,%d)", device_id, n);
    rwr_context *ctx = new (std::nothrow) rwr_context();
    if (!ctx) return set_error(RWR_ERR_HIP, "out of host memory");
    ctx->device = device_id;
    DeviceGuard g(device_id);
    if (!g.ok) {
        delete ctx;
        return set_error(RWR_ERR_HIP, "hipSetDevice(%d) failed", device_id);
    }
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_begin)) != hipSuccess || (e = hipEventCreate(&ctx->ev_end)) != hipSuccess) {
        rwr_ctx_destroy(ctx);
        return set_error(RWR_ERR_HIP, "stream/event creation failed: %s", hipGetErrorString(e));
    }
    ctx->stream = ctx->own_stream;
    ctx->slots[0].stream = ctx->stream;
    (void)preload_kernels();   // have the code objects on the device before the first frame asks for them
    if (const char *e2 = std::getenv("RWR_WAVE_CULL_MIN")) ctx->wave_cull_min = (uint32_t)std::strtoul(e2, nullptr, 10);
    if (const char *e4 = std::getenv("RWR_ONE_PIXEL_PER_LANE")) ctx->force_one_pixel = std::atoi(e4) != 0;
    if (const char *e14 = std::getenv("RWR_FRAME_GRAPH")) ctx->frame_graph_mode = (uint32_t)std::atoi(e14);
    if (const char *e16 = std::getenv("RWR_FUSED_SETUP")) { ctx->fused_setup = std::atoi(e16) != 0; ctx->fused_setup_force = ctx->fused_setup; }
    if (const char *e5 = std::getenv("RWR_AUTO_BVH_FACE_PX")) ctx->auto_bvh_face_px = (float)std::atof(e5);
    if (const char *e6 = std::getenv("RWR_WF_GROUP")) ctx->wf_group = std::min(kWfMaxGroup, std::max(1u, (uint32_t)std::strtoul(e6, nullptr, 10)));
    if (const char *e9 = std::getenv("RWR_WF_STATS")) {
        if (std::atoi(e9) && ctx->d_wf_dbg.ensure(4) == hipSuccess) (void)hipMemset(ctx->d_wf_dbg.ptr, 0, 32);
    }
    if (const char *e13 = std::getenv("RWR_WF_OVERLAP")) ctx->wf_queues = std::min(rwr_context::kWfMaxQueues, std::max(1u, (uint32_t)std::strtoul(e13, nullptr, 10)));
    if (const char *e12 = std::getenv("RWR_WF_ZSPLIT")) ctx->wf_z_split = (uint32_t)std::strtoul(e12, nullptr, 10);
    if (const char *e19 = std::getenv("RWR_WF_PACKET_RAYS")) ctx->wf_packet_dense_rays = (uint32_t)std::strtoul(e19, nullptr, 10);
    if (const char *e20 = std::getenv("RWR_WF_WIDE_LANE")) ctx->wf_wide_lane = std::atoi(e20) != 0 ? 1 : 0;
    if (const char *e15 = std::getenv("RWR_WF_LANE_ITEMS")) ctx->wf_lane_items = std::max(1u, (uint32_t)std::strtoul(e15, nullptr, 10));
    if (const char *e10 = std::getenv("RWR_WF_MIN_PACKET_POOLS")) ctx->wf_min_packet_pools = (uint32_t)std::strtoul(e10, nullptr, 10);
    if (const char *e8 = std::getenv("RWR_WF_PACKET_EXTENT")) ctx->wf_packet_extent = (float)std::atof(e8);
    if (const char *e7 = std::getenv("RWR_WF_PACKET_FILL")) ctx->wf_packet_fill = (float)std::atof(e7);
    if (const char *e11 = std::getenv("RWR_BIN_CAPACITY")) ctx->bin_min_capacity = (uint32_t)std::strtoul(e11, nullptr, 10);
    if (const char *e3 = std::getenv("RWR_BIN_MIN_FACES")) ctx->bin_min_faces = (uint32_t)std::strtoul(e3, nullptr, 10);
    *out_ctx = ctx;
    return RWR_OK;
}

void rwr_ctx_destroy(rwr_context *ctx)
{
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    (void)sync_all(ctx);
    if (ctx->d_wf_dbg.ptr) {
        unsigned long long h[4] = {0, 0, 0, 0};
        (void)hipMemcpy(h, ctx->d_wf_dbg.ptr, sizeof h, hipMemcpyDeviceToHost);
        std::fprintf(stderr, "rwr wavefront pools: packets %llu pools / %llu rays, per-lane %llu pools / %llu rays (leaf extent %g)\n",
                     h[0], h[1], h[2], h[3], (double)ctx->bvh_leaf_extent);
        ctx->d_wf_dbg.release();
    }
    ctx->d_verts.release(); ctx->d_faces.release(); ctx->d_instances.release();
    ctx->d_tris.release(); ctx->d_shade.release(); ctx->d_cull.release(); ctx->d_tangent.release();
    for (auto &t : ctx->d_nmaps) t.release();
    ctx->d_bvh_nodes.release(); ctx->d_bvh_leaf_faces.release();
    for (auto &w : ctx->wf_state) w.release();
    if (ctx->h_wf_live) { (void)hipHostFree(ctx->h_wf_live); ctx->h_wf_live = nullptr; } for (auto &t : ctx->d_texs) t.release();
    ctx->d_face_mat.release(); ctx->d_materials.release();
    for (FrameSlot &sl : ctx->slots) {
        sl.release_buffers();
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.owned) (void)hipStreamDestroy(sl.owned);
    }
    for (hipEvent_t e : ctx->timing_events) (void)hipEventDestroy(e);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    (void)rwr_dist_destroy(ctx);
    if (ctx->probe_stream) { (void)hipStreamSynchronize(ctx->probe_stream); (void)hipStreamDestroy(ctx->probe_stream); }
    ctx->d_probe.release();
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int rwr_ctx_device_info(rwr_context *ctx, char *name, size_t name_cap, int *cu_count, int *wave_size)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    hipDeviceProp_t prop;
    RWR_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_cap) {
        std::snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    return RWR_OK;
}

int rwr_ctx_set_stream(rwr_context *ctx, void *hip_stream)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(sync_all(ctx));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    ctx->slots[0].stream = ctx->stream;
    return RWR_OK;
}

void *rwr_ctx_get_stream(rwr_context *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }

int rwr_scene_clear(rwr_context *ctx)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(sync_all(ctx));
    ctx->st_verts.clear(); ctx->st_faces.clear(); ctx->st_face_mat.clear(); ctx->st_materials.clear();
    for (auto &t : ctx->d_texs) t.release();
    ctx->d_texs.clear();
    for (auto &t : ctx->d_nmaps) t.release();
    ctx->d_nmaps.clear();
    ctx->have_mesh = false;
    ctx->n_faces = ctx->n_verts = ctx->n_tris = 0;
    return RWR_OK;
}

int rwr_scene_add_mesh(rwr_context *ctx, const rwr_model_vertex_small *verts, uint32_t n_verts,
                       const rwr_model_face_small *faces, uint32_t n_faces, const rwr_material_data *material,
                       const uint8_t *rgba8_srgb, uint32_t tex_w, uint32_t tex_h)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n_faces == 0) return RWR_OK;  // nothing to add
    if (!verts || !faces || !material || !rgba8_srgb)
        return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL mesh array with n_faces = %u", n_faces);
    if (n_verts == 0 || tex_w == 0 || tex_h == 0)
        return set_error(RWR_ERR_INVALID_ARGUMENT, "empty vertex array or texture with n_faces = %u", n_faces);
    if (tex_w > kMaxTextureDim || tex_h > kMaxTextureDim)  // byte offsets of the taps are 32-bit (rwr_device.h)
        return set_error(RWR_ERR_INVALID_ARGUMENT, "texture %ux%u larger than %ux%u", tex_w, tex_h, kMaxTextureDim, kMaxTextureDim);
    // The shader indexes vertice_list unchecked (compute.wgsl:191-193); an
    // out-of-range index would be a GPU fault here, so it is rejected up front.
    for (uint32_t f = 0; f < n_faces; f++)
        for (int k = 0; k < 3; k++)
            if (faces[f].indices[k] >= n_verts)
                return set_error(RWR_ERR_INVALID_ARGUMENT, "face %u index %u out of range (n_verts %u)", f, faces[f].indices[k], n_verts);
    const uint64_t total_faces = (uint64_t)ctx->st_faces.size() + n_faces;
    if (total_faces * (ctx->n_instances ? ctx->n_instances : 1u) > 0x7fffffffull || (uint64_t)ctx->st_verts.size() + n_verts > 0xffffffffull)
        return set_error(RWR_ERR_INVALID_ARGUMENT, "too many faces or vertices");
    DeviceGuard g(ctx->device);
    // texture -> linear float4 on the device
    ctx->d_texs.emplace_back();
    DeviceBuffer<float4> &tex = ctx->d_texs.back();
    hipError_t e = tex.ensure((size_t)tex_w * tex_h);
    if (e == hipSuccess) {
        float lut[256];
        build_srgb_lut(lut);
        std::vector<float4> lin((size_t)tex_w * tex_h);
        for (size_t i = 0; i < lin.size(); i++)
            lin[i] = make_float4(lut[rgba8_srgb[4 * i]], lut[rgba8_srgb[4 * i + 1]], lut[rgba8_srgb[4 * i + 2]],
                                 (float)rgba8_srgb[4 * i + 3] / 255.0f);  // alpha is linear in sRGB formats
        e = hipMemcpy(tex.ptr, lin.data(), lin.size() * sizeof(float4), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        ctx->d_texs.back().release();
        ctx->d_texs.pop_back();
        return set_error(RWR_ERR_HIP, "texture upload failed: %s", hipGetErrorString(e));
    }
    const uint32_t vbase = (uint32_t)ctx->st_verts.size(), mid = (uint32_t)ctx->st_materials.size();
    ctx->st_verts.insert(ctx->st_verts.end(), verts, verts + n_verts);
    for (uint32_t f = 0; f < n_faces; f++) {
        rwr_model_face_small fc = faces[f];
        fc.indices[0] += vbase; fc.indices[1] += vbase; fc.indices[2] += vbase;
        ctx->st_faces.push_back(fc);
        ctx->st_face_mat.push_back(mid);
    }
    MaterialRec M{};
    for (int k = 0; k < 3; k++) { M.ambient[k] = material->ambient[k]; M.specular[k] = material->specular[k]; }
    M.tex_w = tex_w; M.tex_h = tex_h; M.tex = tex.ptr;
    M.wmax = (float)(tex_w - 1u); M.hmax = (float)(tex_h - 1u);
    M.nmap = nullptr; M.nmap_w = M.nmap_h = 0u;
    ctx->st_materials.push_back(M);
    ctx->d_nmaps.emplace_back();
    if (mid == 0) ctx->material = *material;
    return RWR_OK;
}

int rwr_scene_commit(rwr_context *ctx)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(sync_all(ctx));
    ctx->n_faces = (uint32_t)ctx->st_faces.size();
    ctx->n_verts = (uint32_t)ctx->st_verts.size();
    ctx->n_tris = 0;
    ctx->have_mesh = true;
    if (ctx->n_faces == 0) {
        ctx->tris_dirty = false;
        return RWR_OK;
    }
    RWR_HIP_CHECK(ctx->d_verts.ensure(ctx->n_verts));
    RWR_HIP_CHECK(ctx->d_faces.ensure(ctx->n_faces));
    RWR_HIP_CHECK(ctx->d_face_mat.ensure(ctx->n_faces));
    RWR_HIP_CHECK(ctx->d_materials.ensure(ctx->st_materials.size()));
    RWR_HIP_CHECK(hipMemcpy(ctx->d_verts.ptr, ctx->st_verts.data(), ctx->st_verts.size() * sizeof(rwr_model_vertex_small), hipMemcpyHostToDevice));
    RWR_HIP_CHECK(hipMemcpy(ctx->d_faces.ptr, ctx->st_faces.data(), ctx->st_faces.size() * sizeof(rwr_model_face_small), hipMemcpyHostToDevice));
    RWR_HIP_CHECK(hipMemcpy(ctx->d_face_mat.ptr, ctx->st_face_mat.data(), ctx->st_face_mat.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    RWR_HIP_CHECK(hipMemcpy(ctx->d_materials.ptr, ctx->st_materials.data(), ctx->st_materials.size() * sizeof(MaterialRec), hipMemcpyHostToDevice));
    ctx->tex_w = ctx->st_materials[0].tex_w;
    ctx->tex_h = ctx->st_materials[0].tex_h;
    ctx->tris_dirty = true;
    const int rc = rebuild_tris(ctx);
    if (rc != RWR_OK) return rc;
    RWR_HIP_CHECK(ensure_frame_buffers(ctx));
    return RWR_OK;
}

int rwr_scene_part_count(rwr_context *ctx, uint32_t *n_parts)
{
    if (!ctx || !n_parts) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    *n_parts = (uint32_t)ctx->st_materials.size();
    return RWR_OK;
}

int rwr_scene_set_normal_map(rwr_context *ctx, uint32_t part, const uint8_t *rgba8_linear, uint32_t tex_w, uint32_t tex_h)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (part >= ctx->st_materials.size()) return set_error(RWR_ERR_INVALID_ARGUMENT, "part %u: the scene has %zu parts", part, ctx->st_materials.size());
    if (rgba8_linear && (tex_w == 0 || tex_h == 0)) return set_error(RWR_ERR_INVALID_ARGUMENT, "empty normal map");
    if (tex_w > kMaxTextureDim || tex_h > kMaxTextureDim)
        return set_error(RWR_ERR_INVALID_ARGUMENT, "normal map %ux%u larger than %ux%u", tex_w, tex_h, kMaxTextureDim, kMaxTextureDim);
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(sync_all(ctx));
    DeviceBuffer<float4> &tex = ctx->d_nmaps[part];
    MaterialRec &M = ctx->st_materials[part];
    if (!rgba8_linear) {
        tex.release();
        M.nmap = nullptr; M.nmap_w = M.nmap_h = 0u;
    } else {
        tex.release();
        RWR_HIP_CHECK(tex.ensure((size_t)tex_w * tex_h));
        std::vector<float4> lin((size_t)tex_w * tex_h);
        for (size_t i = 0; i < lin.size(); i++)   // rgba8unorm, NOT sRGB: a normal map holds vectors
            lin[i] = make_float4((float)rgba8_linear[4 * i] / 255.0f, (float)rgba8_linear[4 * i + 1] / 255.0f,
                                 (float)rgba8_linear[4 * i + 2] / 255.0f, (float)rgba8_linear[4 * i + 3] / 255.0f);
        RWR_HIP_CHECK(hipMemcpy(tex.ptr, lin.data(), lin.size() * sizeof(float4), hipMemcpyHostToDevice));
        M.nmap = tex.ptr; M.nmap_w = tex_w; M.nmap_h = tex_h;
    }
    if (ctx->have_mesh && ctx->d_materials.ptr && ctx->d_materials.count >= ctx->st_materials.size())   // already committed: refresh the device copy
        RWR_HIP_CHECK(hipMemcpy(ctx->d_materials.ptr, ctx->st_materials.data(), ctx->st_materials.size() * sizeof(MaterialRec), hipMemcpyHostToDevice));
    return RWR_OK;
}

int rwr_scene_upload_mesh(rwr_context *ctx, const rwr_model_vertex_small *verts, uint32_t n_verts,
                          const rwr_model_face_small *faces, uint32_t n_faces, const rwr_material_data *material,
                          const uint8_t *rgba8_srgb, uint32_t tex_w, uint32_t tex_h)
{
    int rc = rwr_scene_clear(ctx);
    if (rc == RWR_OK) rc = rwr_scene_add_mesh(ctx, verts, n_verts, faces, n_faces, material, rgba8_srgb, tex_w, tex_h);
    if (rc == RWR_OK) rc = rwr_scene_commit(ctx);
    return rc;
}

int rwr_scene_set_spheres(rwr_context *ctx, const rwr_sphere_buffer_data *spheres, uint32_t n)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n > RWR_MAX_SPHERES) return set_error(RWR_ERR_INVALID_ARGUMENT, "at most %d spheres", RWR_MAX_SPHERES);
    if (n && !spheres) return set_error(RWR_ERR_INVALID_ARGUMENT, "spheres is NULL");
    for (uint32_t i = 0; i < n; i++) ctx->spheres[i] = spheres[i];
    ctx->n_spheres = n;
    return RWR_OK;
}

int rwr_scene_set_triangles(rwr_context *ctx, const rwr_triangle_buffer_data *triangles, uint32_t n)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n > RWR_MAX_TRIANGLES) return set_error(RWR_ERR_INVALID_ARGUMENT, "at most %d single triangles", RWR_MAX_TRIANGLES);
    if (n && !triangles) return set_error(RWR_ERR_INVALID_ARGUMENT, "triangles is NULL with n = %u", n);
    for (uint32_t i = 0; i < n; i++) ctx->triangles[i] = triangles[i];   // passed by value with every launch
    ctx->n_triangles = n;
    return RWR_OK;
}

int rwr_scene_set_instances(rwr_context *ctx, const rwr_instance_raw *instances, uint32_t n)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n && !instances) return set_error(RWR_ERR_INVALID_ARGUMENT, "instances is NULL");
    if ((uint64_t)ctx->n_faces * (n ? n : 1u) > 0x7fffffffull) return set_error(RWR_ERR_INVALID_ARGUMENT, "too many faces");
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(sync_all(ctx));
    if (n) {
        RWR_HIP_CHECK(ctx->d_instances.ensure(n));
        RWR_HIP_CHECK(hipMemcpy(ctx->d_instances.ptr, instances, (size_t)n * sizeof *instances, hipMemcpyHostToDevice));
    }
    ctx->n_instances = n;
    if (ctx->have_mesh && ctx->n_faces) {
        ctx->tris_dirty = true;
        return rebuild_tris(ctx);
    }
    return RWR_OK;
}

int rwr_resize(rwr_context *ctx, const rwr_screen *screen)
{
    if (!ctx || !screen) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    // resize ignores zero sizes (lib.rs:773); here that is an explicit error.
    if (screen->width == 0 || screen->height == 0) return set_error(RWR_ERR_INVALID_ARGUMENT, "zero-sized screen");
    if ((uint64_t)screen->width * screen->height > (1ull << 30)) return set_error(RWR_ERR_INVALID_ARGUMENT, "screen too large");
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(sync_all(ctx));
    ctx->screen = *screen;
    for (uint32_t i = 0; i < ctx->n_slots; i++) {
        RWR_HIP_CHECK(ensure_slot_targets(ctx, i));
        ctx->slots[i].aux_valid = false;
    }
    for (rwr_context::GatherSet &gs : ctx->gather) gs.valid = false;   // a frame gathered at the old size is gone
    RWR_HIP_CHECK(ensure_frame_buffers(ctx));
    return RWR_OK;
}

// One frame — rows [row_begin, row_end) in strips of 8 rows, strip k starting at row_begin + k * row_pitch (row_pitch 8: the
// whole band; 8 N: every N-th strip).
static int render_frame(rwr_context *ctx, const rwr_camera_inv_uniform *camera, const rwr_render_params *params,
                        uint32_t row_begin, uint32_t row_end, uint32_t row_pitch)
{
    if (!ctx || !camera) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (ctx->screen.width == 0) return set_error(RWR_ERR_NOT_READY, "rwr_resize has not been called");
    if (!ctx->have_mesh) return set_error(RWR_ERR_NOT_READY, "rwr_scene_upload_mesh has not been called");
    if (row_begin > row_end || row_end > ctx->screen.height)
        return set_error(RWR_ERR_INVALID_ARGUMENT, "row band [%u,%u) outside the %u-row frame", row_begin, row_end,
                         ctx->screen.height);
    rwr_render_params rp = {1, 0, 0, 0};
    if (params) rp = *params;
    if (rp.spp == 0) return set_error(RWR_ERR_INVALID_ARGUMENT, "spp must be >= 1");
    if (rp.max_bounces > 1) return set_error(RWR_ERR_UNSUPPORTED, "max_bounces > 1 is not supported");
    if ((rp.flags & RWR_FLAG_USE_BVH) && (rp.spp != 1 || rp.max_bounces != 0))
        return set_error(RWR_ERR_UNSUPPORTED, "RWR_FLAG_USE_BVH applies to the reference frame (spp 1, no bounce); bounce rays always use the BVH");
    if (rp.spp > 4096) return set_error(RWR_ERR_INVALID_ARGUMENT, "spp must be <= 4096");
    const bool wavefront = rp.spp != 1 || rp.max_bounces != 0;
    // the reference's dormant parts (single-triangle passes, orthographic rays) have their own plain kernel
    const bool dormant = ctx->n_triangles != 0 || (rp.flags & RWR_FLAG_ORTHO_RAYS) != 0;
    if (dormant && (wavefront || (rp.flags & RWR_FLAG_USE_BVH)))
        return set_error(RWR_ERR_UNSUPPORTED, "single-triangle passes and RWR_FLAG_ORTHO_RAYS apply to the reference frame (spp 1, no bounce, no RWR_FLAG_USE_BVH)");

    DeviceGuard g(ctx->device);
    const size_t n = (size_t)ctx->screen.width * ctx->screen.height;
    const bool aux = (rp.flags & RWR_FLAG_AUX_OUTPUTS) != 0;
    int rc = rebuild_tris(ctx);
    if (rc != RWR_OK) return rc;
    // Frame slot: the next one in turn.  A slot owns everything a frame writes — targets, per-frame records, and for the
    // wavefront integrator a whole set of accumulators and ray queues (WfState) — so frames in different slots share
    // nothing and a slot is reused in stream order.
    ctx->cur = (ctx->cur + 1u) % ctx->n_slots;
    FrameSlot &sl = ctx->slots[ctx->cur];
    const hipStream_t stream = sl.stream;
    if (aux) {
        // a band render leaves the rest of the planes untouched: they start zeroed, like the targets
        const bool fresh = sl.d_color_f32.count < n * 4 || sl.d_obj_id.count < n || sl.d_hit_t.count < n;
        RWR_HIP_CHECK(sl.d_color_f32.ensure(n * 4));
        RWR_HIP_CHECK(sl.d_obj_id.ensure(n));
        RWR_HIP_CHECK(sl.d_hit_t.ensure(n));
        if (fresh) {
            RWR_HIP_CHECK(hipMemsetAsync(sl.d_color_f32.ptr, 0, n * 4 * sizeof(float), stream));
            RWR_HIP_CHECK(hipMemsetAsync(sl.d_obj_id.ptr, 0, n * sizeof(int32_t), stream));
            RWR_HIP_CHECK(hipMemsetAsync(sl.d_hit_t.ptr, 0, n * sizeof(float), stream));
        }
    }
    const uint32_t total_tris = ctx->n_tris;
    RWR_HIP_CHECK(sl.d_ftris.ensure(total_tris));
    RWR_HIP_CHECK(sl.d_tnum.ensure(total_tris));

    FrameParams fp{};
    fp.cam = *camera;
    fp.wave_cull_min = ctx->wave_cull_min;
    fp.width = ctx->screen.width;
    fp.height = ctx->screen.height;
    fp.row_begin = row_begin;
    fp.row_end = row_end;
    fp.row_pitch = row_pitch;
    fp.n_spheres = ctx->n_spheres;
    for (uint32_t i = 0; i < ctx->n_spheres; i++) fp.spheres[i] = ctx->spheres[i];
    fp.n_tris = ctx->n_tris;
    fp.tex_w = ctx->tex_w;
    fp.tex_h = ctx->tex_h;
    fp.tex_wmax = ctx->tex_w ? (float)(ctx->tex_w - 1u) : 0.0f;
    fp.tex_hmax = ctx->tex_h ? (float)(ctx->tex_h - 1u) : 0.0f;
    fp.flags = rp.flags;
    for (int k = 0; k < 3; k++) {
        fp.ambient[k] = ctx->material.ambient[k];
        fp.specular[k] = ctx->material.specular[k];
    }
    fp.materials = ctx->d_materials.ptr;
    fp.n_materials = (uint32_t)ctx->st_materials.size();
    fp.tangents = ctx->d_tangent.ptr;
    const float4 *tex0 = ctx->d_texs.empty() ? nullptr : ctx->d_texs[0].ptr;
    Targets tg{sl.d_color.ptr, sl.d_depth.ptr, aux ? sl.d_color_f32.ptr : nullptr,
               aux ? sl.d_obj_id.ptr : nullptr, aux ? sl.d_hit_t.ptr : nullptr};
    CullConsts cc;
    compute_cull_consts(*camera, ctx->screen.width, ctx->screen.height, cc);
    compute_sphere_rects(cc, ctx->spheres, ctx->n_spheres, ctx->screen.width, ctx->screen.height, fp.sphere_rect);
    // the whole mesh's screen rectangle: tiles outside it skip the mesh pass altogether (same margins as the spheres')
    double mesh_rect[4] = {0, 0, 0, 0};
    const bool have_mesh_rect = ctx->n_tris != 0 && mesh_screen_rect(cc, ctx->aabb_lo, ctx->aabb_hi, mesh_rect);
    const float inf = std::numeric_limits<float>::infinity();
    fp.mesh_rect[0] = fp.mesh_rect[1] = -inf; fp.mesh_rect[2] = fp.mesh_rect[3] = inf;
    if (have_mesh_rect) {
        fp.mesh_rect[0] = (float)(mesh_rect[0] - 1.0 - 1e-4 * std::fabs(mesh_rect[0]));
        fp.mesh_rect[1] = (float)(mesh_rect[1] - 1.0 - 1e-4 * std::fabs(mesh_rect[1]));
        fp.mesh_rect[2] = (float)(mesh_rect[2] + 1.0 + 1e-4 * std::fabs(mesh_rect[2]));
        fp.mesh_rect[3] = (float)(mesh_rect[3] + 1.0 + 1e-4 * std::fabs(mesh_rect[3]));
    }
    for (int k = 0; k < 4; k++) {
        const double v = k < 2 ? std::floor((double)fp.mesh_rect[k]) : std::ceil((double)fp.mesh_rect[k]);
        fp.mesh_px[k] = (int32_t)std::fmax(-1e9, std::fmin(1e9, v));   // +-inf -> +-1e9
    }
    fp.spp = rp.spp;
    fp.seed = rp.seed;
    fp.bounces = rp.max_bounces;
    // Per-frame records and tables (k_frame_setup): they depend on the camera, so they are rebuilt
    // every frame, on the render stream just ahead of the render kernel.  (Running this small
    // kernel on a side stream, double-buffered so that it overlaps the previous frame, was
    // measured 4-10 us SLOWER per frame than the 3 us it hides: cross-stream event waits cost
    // more than the kernel.)
    FrameSetupOut so{};
    so.ray_pairs = ((ctx->screen.width + 63u) / 64u) * 32u;  // whole 64-pixel workgroup columns
    so.ray_rows = ctx->screen.height + 8u;                   // whole 8-row tiles below any band
    RWR_HIP_CHECK(sl.d_ray_colp.ensure(2u * (size_t)so.ray_pairs));
    RWR_HIP_CHECK(sl.d_ray_row.ensure(so.ray_rows));
    so.ftris = sl.d_ftris.ptr; so.tnum = sl.d_tnum.ptr;
    so.ray_colp = sl.d_ray_colp.ptr; so.ray_row = sl.d_ray_row.ptr;
    fp.ray_colp = so.ray_colp; fp.ray_row = so.ray_row; fp.tnum = so.tnum;
    if (rp.spp != 1 || rp.max_bounces != 0) {
        // the wavefront integrator's per-tile ray counts and live-tile count start every frame from zero: k_frame_setup zeroes them
        rwr_context::WfState &W0 = ctx->wf_state[ctx->n_slots > 1u ? ctx->cur : 0u];
        fp.row_begin = row_begin; fp.row_end = row_end; fp.row_pitch = row_pitch;
        const size_t n_tiles0 = (size_t)((ctx->screen.width + kWfTileW - 1u) / kWfTileW) * band_strips(fp);
        RWR_HIP_CHECK(W0.d_wave_total.ensure(n_tiles0 * 4u));
        RWR_HIP_CHECK(W0.d_tiles.ensure(2u * n_tiles0 + 1u));
        so.zero_a = W0.d_wave_total.ptr; so.n_zero_a = (uint32_t)(n_tiles0 * 4u);
        so.zero_b = W0.d_tiles.ptr + 2u * n_tiles0; so.n_zero_b = 1u;   // live_count (below)
    }
    // A/B (RWR_FRAME_GRAPH=1): the plain reference frame — records + frame kernel, nothing else on the stream — as one graph launch
    const bool as_graph = ctx->frame_graph_mode != 0u && !(rp.spp != 1 || rp.max_bounces != 0) && !aux && ctx->n_triangles == 0 &&
                          !(rp.flags & (RWR_FLAG_ORTHO_RAYS | RWR_FLAG_USE_BVH | RWR_FLAG_NO_CULL | RWR_FLAG_ONE_PIXEL_PER_LANE)) &&
                          !ctx->force_one_pixel && ctx->n_tris != 0 && ctx->n_tris <= ctx->bin_min_faces && !ctx->timing_every;
    if (as_graph) {
        std::vector<unsigned char> key(sizeof(FrameParams) + sizeof(CullConsts));
        std::memcpy(key.data(), &fp, sizeof fp);
        std::memcpy(key.data() + sizeof fp, &cc, sizeof cc);
        if (!sl.frame_graph || key != sl.frame_graph_key) {
            hipGraph_t g = nullptr;
            RWR_HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            hipError_t e = launch_frame_setup(stream, cc, *camera, ctx->screen.width, ctx->screen.height, ctx->d_cull.ptr, ctx->d_tris.ptr, ctx->n_tris, so);
            if (e == hipSuccess) e = launch_primary_p2(stream, fp, ctx->d_tris.ptr, ctx->d_shade.ptr, sl.d_ftris.ptr, tex0, tg);
            const hipError_t e2 = hipStreamEndCapture(stream, &g);
            RWR_HIP_CHECK(e);
            RWR_HIP_CHECK(e2);
            bool updated = false;
            if (sl.frame_graph) {
                hipGraphNode_t bad = nullptr;
                hipGraphExecUpdateResult res;
                updated = hipGraphExecUpdate(sl.frame_graph, g, &bad, &res) == hipSuccess;
                if (!updated) { (void)hipGetLastError(); (void)hipGraphExecDestroy(sl.frame_graph); sl.frame_graph = nullptr; }
            }
            if (!updated) {
                const hipError_t e3 = hipGraphInstantiate(&sl.frame_graph, g, nullptr, nullptr, 0);
                if (e3 != hipSuccess) { (void)hipGraphDestroy(g); RWR_HIP_CHECK(e3); }
            }
            (void)hipGraphDestroy(g);
            sl.frame_graph_key.swap(key);
        }
        RWR_HIP_CHECK(hipGraphLaunch(sl.frame_graph, stream));
        ctx->last_spp = 0;
        ctx->last_had_bounce = false;
        ctx->last_primary = 0;
        for (uint32_t y0 = row_begin; y0 < row_end; y0 += row_pitch) ctx->last_primary += (uint64_t)std::min(kStripRows, row_end - y0) * ctx->screen.width;
        ctx->last_bounce = 0;
        return RWR_OK;
    }
    // The plain reference frame — records + frame kernel, nothing else — as ONE launch: the frame kernel's first workgroups
    // make the records (kernels_primary_p2.hip, FUSED).
    // Where it pays (A/B in one box, tools/fused_ab.py, profiles/r03_fused_ab.txt): SMALL frames with frames in flight, which
    // are bound by the host's launches — one rank's share of a multi-GPU 1080p frame: 9.8 -> 6.9 us per frame (1/8), 10.4 -> 8.2
    // (1/4), the host's enqueue time 9.8 -> 5.2 us.  A whole 1080p frame is VALU-bound and LOSES (15.0 -> 17.4 us with two slots:
    // the next frame's waiting workgroups hold slots the running frame could use; 22.7 -> 23.8 us alone), so it keeps its two
    // launches.  RWR_FUSED_SETUP=1 forces the fused form wherever it is possible (tests), 0 switches it off.
    const uint32_t render_groups = ((ctx->screen.width + 63u) / 64u) * ((row_end - row_begin + row_pitch - 1u) / std::max(1u, row_pitch));
    const bool fused_pays = ctx->fused_setup_force || (ctx->n_slots > 1u && render_groups <= 1200u);
    const bool fused = ctx->fused_setup && fused_pays && !(rp.spp != 1 || rp.max_bounces != 0) && !aux && ctx->n_triangles == 0 &&
                       !(rp.flags & (RWR_FLAG_ORTHO_RAYS | RWR_FLAG_USE_BVH | RWR_FLAG_NO_CULL | RWR_FLAG_ONE_PIXEL_PER_LANE | RWR_FLAG_NORMAL_MAP)) &&
                       !ctx->force_one_pixel && ctx->n_tris != 0 && ctx->n_tris <= ctx->bin_min_faces && !ctx->timing_every &&
                       row_end > row_begin;
    if (fused) {
        FusedSetup fs{};
        fs.cc = cc;
        fs.cull = ctx->d_cull.ptr;
        fs.out = so;
        fs.nb_tris = (ctx->n_tris + 255u) / 256u;
        fs.n_blocks = fs.nb_tris + (so.ray_pairs + so.ray_rows + 255u) / 256u;
        fs.extra_rows = primary_p2_fused_rows(fp, fs.n_blocks);
        if (!sl.d_fused.ptr || sl.fused_blocks != fs.n_blocks) {   // first use, or another scene / frame size: the count starts over
            RWR_HIP_CHECK(hipStreamSynchronize(stream));
            RWR_HIP_CHECK(sl.d_fused.ensure(2));
            RWR_HIP_CHECK(hipMemsetAsync(sl.d_fused.ptr, 0, 2 * sizeof(uint32_t), stream));
            sl.fused_count = 0;
            sl.fused_blocks = fs.n_blocks;
        }
        fs.flag = sl.d_fused.ptr;
        fs.flag_base = sl.fused_count;
        sl.fused_count += fs.n_blocks;   // (modulo 2^32, like the device's count)
        sl.fused_used = true;
        RWR_HIP_CHECK(launch_primary_p2(stream, fp, ctx->d_tris.ptr, ctx->d_shade.ptr, sl.d_ftris.ptr, tex0, tg, nullptr, nullptr, &fs));
        sl.aux_valid = false;
        ctx->last_spp = 0;
        ctx->last_had_bounce = false;
        ctx->last_primary = 0;
        for (uint32_t y0 = row_begin; y0 < row_end; y0 += row_pitch) ctx->last_primary += (uint64_t)std::min(kStripRows, row_end - y0) * ctx->screen.width;
        ctx->last_bounce = 0;
        return RWR_OK;
    }
    RWR_HIP_CHECK(launch_frame_setup(stream, cc, *camera, ctx->screen.width, ctx->screen.height, ctx->d_cull.ptr,
                                     ctx->d_tris.ptr, ctx->n_tris, so));
    if (ctx->n_tris && !(rp.flags & RWR_FLAG_NO_CULL)) {
        const uint32_t bins_x = (ctx->screen.width + kBinW - 1) / kBinW, bins_y = (row_end - row_begin + kBinH - 1) / kBinH;
        if (ctx->n_tris > ctx->bin_min_faces) {
            // more faces than one 256-wide batch: bin them per 64x32-pixel screen region, once per frame.  The lists
            // are sized by a count pass on the device; the buffer keeps what the previous frames needed (read back a
            // frame late through pinned memory, never waited for) with headroom, and a frame whose lists do not fit
            // walks the whole scene instead — the same pixels — while the buffer grows for the next one.
            const size_t n_bins = (size_t)bins_x * bins_y;
            if (!sl.h_bin_total) {
                RWR_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&sl.h_bin_total), sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped));   // fine-grained: kernels store to it, the host reads it without a synchronisation
                *sl.h_bin_total = 0u;
            }
            const uint64_t needed = *sl.h_bin_total;
            uint64_t capacity = std::max<uint64_t>(sl.d_bin_lists.count, std::max<uint64_t>(ctx->bin_min_capacity, ctx->bin_min_capacity >= 65536u ? 8ull * ctx->n_tris : 0ull));
            if (needed > capacity || needed + needed / 4u > capacity) capacity = std::max<uint64_t>(capacity, needed + needed / 2u);
            capacity = std::min<uint64_t>(capacity, 0xfffffff0ull);
            if (capacity > sl.d_bin_lists.count) RWR_HIP_CHECK(hipStreamSynchronize(stream));   // the old buffer may still be read
            RWR_HIP_CHECK(sl.d_bin_lists.ensure((size_t)capacity));
            RWR_HIP_CHECK(sl.d_bin_counts.ensure(5u * (size_t)n_bins));   // four wave counts per bin, then the bins' own counts (launch_bin_faces)
            RWR_HIP_CHECK(sl.d_bin_offsets.ensure(n_bins));
            RWR_HIP_CHECK(sl.d_bin_total.ensure(1));
            RWR_HIP_CHECK(launch_bin_faces(stream, sl.d_ftris.ptr, ctx->n_tris, row_begin, sl.d_bin_lists.ptr, sl.d_bin_counts.ptr,
                                           sl.d_bin_offsets.ptr, sl.d_bin_total.ptr, bins_x, bins_y, (uint32_t)sl.d_bin_lists.count, fp.mesh_px,
                                           sl.h_bin_total));   // (the scan writes the total to the pinned word itself: no copy command)
            fp.bins = BinGrid{sl.d_bin_lists.ptr, sl.d_bin_counts.ptr + 4u * (size_t)n_bins, sl.d_bin_offsets.ptr, bins_x, bins_y, (uint32_t)sl.d_bin_lists.count, 1u};
        }
    }
    const bool time_this = ctx->timing_every && (ctx->timing_calls++ % ctx->timing_every == 0) && ctx->timing_pairs < 256;
    if (time_this) {
        if (ctx->timing_events.size() < 2u * (ctx->timing_pairs + 1u)) {
            hipEvent_t a, b;
            RWR_HIP_CHECK(hipEventCreate(&a));
            RWR_HIP_CHECK(hipEventCreate(&b));
            ctx->timing_events.push_back(a);
            ctx->timing_events.push_back(b);
        }
    }
    // the two-pixel frame kernel is timed by its own dispatch timestamps; everything else by stream events
    // faces much smaller than a tile: the per-ray BVH kernel is the faster way to the same frame
    const bool auto_bvh = !wavefront && !dormant && !(rp.flags & (RWR_FLAG_NO_CULL | RWR_FLAG_ONE_PIXEL_PER_LANE)) &&
                          !ctx->force_one_pixel && ctx->n_tris > ctx->bin_min_faces && ctx->auto_bvh_face_px > 0.0f &&
                          mean_face_pixels(have_mesh_rect, mesh_rect, ctx->n_tris, ctx->screen.width, ctx->screen.height) <
                              (double)ctx->auto_bvh_face_px;
    const bool dispatch_timed = time_this && !wavefront && !dormant && !auto_bvh && !(rp.flags & RWR_FLAG_USE_BVH) &&
                                !((rp.flags & RWR_FLAG_ONE_PIXEL_PER_LANE) || ctx->force_one_pixel);
    if (time_this && !dispatch_timed) RWR_HIP_CHECK(hipEventRecord(ctx->timing_events[2 * ctx->timing_pairs], stream));
    if (dormant) {
        SingleTriangles st{};
        st.n = ctx->n_triangles;
        for (uint32_t i = 0; i < ctx->n_triangles; i++) st.t[i] = ctx->triangles[i];
        RWR_HIP_CHECK(launch_primary_dormant(stream, fp, st, ctx->d_tris.ptr, ctx->d_shade.ptr, tex0, tg));
        ctx->last_spp = 0;
    } else if (!wavefront && ((rp.flags & RWR_FLAG_USE_BVH) || auto_bvh)) {
        const BvhDevice bvh_p{ctx->d_bvh_nodes.ptr, ctx->d_bvh_leaf_faces.ptr, ctx->bvh_n_nodes, 3u * ctx->bvh_depth + 2u, 0.0f, 0u, 0u, 0u, 0u};
        RWR_HIP_CHECK(launch_primary_bvh(stream, fp, ctx->d_tris.ptr, ctx->d_shade.ptr, bvh_p, tex0, tg));
        ctx->last_spp = 0;
    } else if (!wavefront) {
        if ((rp.flags & RWR_FLAG_ONE_PIXEL_PER_LANE) || ctx->force_one_pixel)
            RWR_HIP_CHECK(launch_primary(stream, fp, ctx->d_tris.ptr, ctx->d_shade.ptr, sl.d_ftris.ptr, tex0, tg));
        else
            RWR_HIP_CHECK(launch_primary_p2(stream, fp, ctx->d_tris.ptr, ctx->d_shade.ptr, sl.d_ftris.ptr, tex0, tg,
                                            dispatch_timed ? ctx->timing_events[2 * ctx->timing_pairs] : nullptr,
                                            dispatch_timed ? ctx->timing_events[2 * ctx->timing_pairs + 1] : nullptr));
        ctx->last_spp = 0;
    } else {
        // wavefront integrator: the samples are traced in launch groups; per group the primary stage (all of the
        // group's samples of every pixel, rays into the fixed-slot queue) then the bounce stage (one workgroup per
        // 64x8-pixel tile and its ray pool)
        rwr_context::WfState &W = ctx->wf_state[ctx->n_slots > 1u ? ctx->cur : 0u];   // this slot's accumulators and queues
        const uint32_t group = std::min(rp.spp, ctx->wf_group ? ctx->wf_group : (ctx->n_slots > 1u ? 64u : 32u));
        const uint32_t tiles_x = (ctx->screen.width + kWfTileW - 1u) / kWfTileW, tiles_y = band_strips(fp);
        const uint32_t n_tiles = tiles_x * tiles_y;
        RWR_HIP_CHECK(W.d_wave_total.ensure((size_t)n_tiles * 4u));   // (sized and zeroed with the frame's records: k_frame_setup)
        if (W.d_fix.count < 4u * n) W.fix_clean = false;
        RWR_HIP_CHECK(W.d_fix.ensure(4u * n));
        if (!W.fix_clean)   // first use, a new size, or a frame that did not reach its resolve
            RWR_HIP_CHECK(hipMemsetAsync(W.d_fix.ptr, 0, 4u * n * sizeof(unsigned long long), stream));
        W.fix_clean = false;
        // two launch groups in flight (each on its own stream, with its own half of the queue) when the frame has several
        const size_t n_queues = rp.max_bounces ? std::min<size_t>(ctx->wf_queues, (rp.spp + group - 1u) / group) : 1u;
        const bool overlap = n_queues > 1u;
        const size_t slots = (size_t)n_tiles * group * kWfTilePixels;
        if (rp.max_bounces) {
            // The ray queues are fixed-slot (space instead of atomics: 36 B per slot of every tile of the launch), the one large
            // allocation of the library — 19 GB for a 4K frame at 64 samples per group.  A frame whose queues cannot be held is
            // refused with its size, not left to a failed hipMalloc half-way through.
            if (W.d_rays.count < n_queues * 2u * slots) {
                const size_t need = n_queues * slots * (2u * sizeof(float4) + 2u * sizeof(uint16_t));
                const size_t held = W.d_rays.count * sizeof(float4) + (W.d_sorted.count + W.d_bins.count) * sizeof(uint16_t);
                size_t free_b = 0, total_b = 0;
                RWR_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
                if (need > free_b + held)
                    return set_error(RWR_ERR_UNSUPPORTED, "the frame's ray queues need %.1f GB (%zu tiles x %u samples per launch group x 512 slots x 36 B x %zu queues), "
                                     "%.1f GB are free: fewer frames in flight (each slot holds its own queues) or RWR_WF_GROUP < %u",
                                     need * 1e-9, (size_t)n_tiles, group, n_queues, (free_b + held) * 1e-9, group);
            }
            RWR_HIP_CHECK(W.d_rays.ensure(n_queues * 2u * slots));
            RWR_HIP_CHECK(W.d_sorted.ensure(n_queues * slots));
            RWR_HIP_CHECK(W.d_bins.ensure(n_queues * slots));
            RWR_HIP_CHECK(W.d_masks.ensure(n_queues * n_tiles * group * 8u));
            RWR_HIP_CHECK(W.d_pool_info.ensure(n_queues * n_tiles * wf_pool_info_bytes()));
            RWR_HIP_CHECK(W.d_pool_list.ensure(n_queues * 2u * (size_t)n_tiles));
            if (overlap && !W.fork) RWR_HIP_CHECK(hipEventCreateWithFlags(&W.fork, hipEventDisableTiming));
            for (size_t q = 0; q < n_queues; q++) {
                if (q && !W.streams[q]) RWR_HIP_CHECK(hipStreamCreateWithFlags(&W.streams[q], hipStreamNonBlocking));
                if (q && !W.join[q]) RWR_HIP_CHECK(hipEventCreateWithFlags(&W.join[q], hipEventDisableTiming));
            }
            if (!W.d_live.ptr) {
                RWR_HIP_CHECK(W.d_live.ensure(4u * rwr_context::kWfMaxQueues));
                RWR_HIP_CHECK(hipMemsetAsync(W.d_live.ptr, 0, 4u * rwr_context::kWfMaxQueues * sizeof(uint32_t), stream));
            }
            if (!ctx->h_wf_live) {
                RWR_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_wf_live), 2 * sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped));   // fine-grained: kernels store to it, the host reads it without a synchronisation
                ctx->h_wf_live[0] = ctx->h_wf_live[1] = 0u;
            }
        }
        // Did the frame before show LITTLE — fewer than 1 024 live tiles, and at most half of this frame's tiles (a small mesh
        // on an empty screen; not a small frame full of geometry, such as a row band of a multi-GPU frame: measured, that one
        // is best left alone)?  Its count arrives through pinned memory, a frame late, never waited for.  Then several
        // workgroups share a tile's samples in the primary stage (enough of them to fill the chip twice, up to one per
        // sample), and the tiles anything can be seen through are listed first (below).  Any choice gives the same frame: all
        // sums are integers.
        const uint32_t live = ctx->h_wf_live ? ctx->h_wf_live[0] + ctx->h_wf_live[1] : 0u;
        const bool shows_little = live != 0u && live < 1024u && 2u * live <= n_tiles;
        uint32_t z_split = ctx->wf_z_split;
        if (z_split == 0u) {
            z_split = 1u;
            if (shows_little)
                while (z_split < kWfMaxGroup && z_split * live < 2048u) z_split *= 2u;
        }
        // queue q: its half of every per-group buffer and its set of four counters (the primary stage zeroes the set it
        // is about to fill).  With one queue the frame's sums are read-modify-written by the tile's only workgroup; with
        // two, a group's primary stage runs beside the other group's trace kernels and everybody adds atomically.
        // A frame expected to show little (z_split > 1: the previous frame did) first lists the tiles anything can be seen
        // through; the primary stage, the sort and the resolve then touch those alone.  Same frame either way.
        uint32_t *live_list = nullptr, *live_count = nullptr, *tile_live = nullptr;
        if (z_split > 1u && (shows_little || ctx->wf_z_split != 0u) && !(rp.flags & RWR_FLAG_NO_CULL)) {   // (a forced split: the tests' way in)
            RWR_HIP_CHECK(W.d_tiles.ensure(2u * (size_t)n_tiles + 1u));
            live_list = W.d_tiles.ptr; tile_live = live_list + n_tiles; live_count = tile_live + n_tiles;   // (zeroed by k_frame_setup)
            RWR_HIP_CHECK(launch_wf_classify(stream, fp, sl.d_ftris.ptr, tg, tiles_x, live_list, live_count, tile_live));
        }
        WfBuffers wfq[rwr_context::kWfMaxQueues];
        for (size_t q = 0; q < n_queues; q++) {
            const size_t h = q;
            wfq[q] = WfBuffers{W.d_fix.ptr,
                               W.d_rays.ptr ? W.d_rays.ptr + h * 2u * slots : nullptr,
                               W.d_masks.ptr ? W.d_masks.ptr + h * n_tiles * group * 8u : nullptr,
                               W.d_bins.ptr ? W.d_bins.ptr + h * slots : nullptr,
                               W.d_sorted.ptr ? W.d_sorted.ptr + h * slots : nullptr,
                               W.d_wave_total.ptr, group, tiles_x, ctx->d_wf_dbg.ptr,
                               rp.max_bounces ? W.d_live.ptr + h * 4u : nullptr, overlap ? 1u : 0u, live_list, live_count, tile_live};
        }
        // The per-lane trace kernel as 1 024-thread workgroups that share ONE copy of the nodelets in LDS (kernels_wf_bounce.hip,
        // WIDE; only for a BVH too large for a copy per 256-thread workgroup and small enough for one per CU), one work item per
        // pool: when frames overlap and the frame before traced most of its pools per lane (a small mesh on an empty screen at few
        // samples: configs[3] 0.604 -> 0.54 ms; one frame at a time it loses, 0.73 -> 0.81, and configs[4]'s frame, whose large
        // pools are packets, loses 2 %: both keep the 256-thread kernel).  Same frame either way.
        const uint32_t prev_packets = ctx->h_wf_live ? ctx->h_wf_live[0] : 0u, prev_lane = ctx->h_wf_live ? ctx->h_wf_live[1] : 0u;
        const bool wide_lane = ctx->wf_wide_lane >= 0 ? ctx->wf_wide_lane != 0
                                                       : (ctx->n_slots > 1u && prev_lane >= 128u && prev_lane >= 4u * prev_packets);
        const BvhDevice bvh{ctx->d_bvh_nodes.ptr, ctx->d_bvh_leaf_faces.ptr, ctx->bvh_n_nodes, 3u * ctx->bvh_depth + 2u,
                            ctx->wf_packet_extent * ctx->bvh_leaf_extent, ctx->wf_min_packet_pools,
                            // work items of the per-lane trace kernel when pools are few: one 256-ray chunk each for a context that
                            // renders one frame at a time (the chip has nothing else to do: as many items as possible), about four
                            // chunks each when frames overlap (a pool's rays grow with the group's samples; measured at configs[3],
                            // 16 samples: 0.625 -> 0.607 ms with 4 096 items, but 0.74 -> 0.79 ms one frame at a time; configs[4]'s
                            // frame, 64 samples: 16 384 is best either way)
                            ctx->wf_lane_items ? ctx->wf_lane_items : (wide_lane ? 256u : ctx->n_slots > 1u ? std::min(16384u, 256u * group) : 16384u),
                            ctx->wf_packet_dense_rays, wide_lane ? 1u : 0u};
        if (overlap) {   // the other streams start behind this frame's setup (and so behind the previous frame's resolve)
            RWR_HIP_CHECK(hipEventRecord(W.fork, stream));
            for (size_t q = 1; q < n_queues; q++) RWR_HIP_CHECK(hipStreamWaitEvent(W.streams[q], W.fork, 0));
        }
        const uint32_t *last_counters = nullptr;
        for (uint32_t s0 = 0, g = 0; s0 < rp.spp; s0 += group, g++) {
            const uint32_t cnt = std::min(group, rp.spp - s0);
            const size_t q = g % n_queues;
            hipStream_t gs = q ? W.streams[q] : stream;
            RWR_HIP_CHECK(launch_wf_primary(gs, fp, ctx->d_tris.ptr, ctx->d_shade.ptr, sl.d_ftris.ptr, tex0, tg, wfq[q], s0, cnt, z_split));
            if (rp.max_bounces) {
                RWR_HIP_CHECK(launch_wf_bounce(gs, fp, ctx->d_tris.ptr, ctx->d_shade.ptr, bvh, tex0, wfq[q], n_tiles, cnt,
                                               (uint32_t)std::fmax(1.0f, std::ceil(ctx->wf_packet_fill * (float)(cnt * kWfTilePixels))),
                                               W.d_pool_info.ptr + q * n_tiles * wf_pool_info_bytes(), W.d_pool_list.ptr + q * 2u * (size_t)n_tiles));
                if (s0 + group >= rp.spp) last_counters = wfq[q].counters;   // the last group's live-pool counts, for the next frame's split
            }
        }
        for (size_t q = 1; q < n_queues; q++) {
            RWR_HIP_CHECK(hipEventRecord(W.join[q], W.streams[q]));
            RWR_HIP_CHECK(hipStreamWaitEvent(stream, W.join[q], 0));
        }
        // (the resolve also hands the last group's live pool counts to the host: a store to pinned memory, no copy command.  The
        // same store at the top of the per-lane trace kernel made THAT kernel twice as slow, 453 -> 840 us at configs[3], with the
        // pointer null and the instruction mix unchanged; here it costs nothing measurable.)
        RWR_HIP_CHECK(launch_wf_resolve(stream, fp, tg, wfq[0], last_counters, last_counters ? ctx->h_wf_live : nullptr));
        W.fix_clean = true;   // (the resolve zeroes what it reads; rows outside the band were never touched)
        ctx->last_spp = rp.spp;
        ctx->last_segments = n_tiles;
        ctx->last_wf_state = ctx->n_slots > 1u ? ctx->cur : 0u;
        ctx->last_had_bounce = rp.max_bounces != 0;
    }
    if (time_this) {
        if (!dispatch_timed) RWR_HIP_CHECK(hipEventRecord(ctx->timing_events[2 * ctx->timing_pairs + 1], stream));
        ctx->timing_pairs++;
    }
    sl.aux_valid = aux;
    uint64_t rows_rendered = 0;   // the strips' rows inside [row_begin, row_end)
    for (uint32_t y0 = row_begin; y0 < row_end; y0 += row_pitch) rows_rendered += std::min(kStripRows, row_end - y0);
    ctx->last_primary = (uint64_t)ctx->screen.width * rows_rendered * rp.spp;
    ctx->last_bounce = 0;  // filled in lazily by rwr_last_render_stats from the pass counters
    return RWR_OK;
}

int rwr_render(rwr_context *ctx, const rwr_camera_inv_uniform *camera, const rwr_render_params *params)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    return render_frame(ctx, camera, params, 0, ctx->screen.height, kStripRows);
}

int rwr_render_rows(rwr_context *ctx, const rwr_camera_inv_uniform *camera, const rwr_render_params *params,
                    uint32_t row_begin, uint32_t row_end)
{
    return render_frame(ctx, camera, params, row_begin, row_end, kStripRows);
}

int rwr_render_strips(rwr_context *ctx, const rwr_camera_inv_uniform *camera, const rwr_render_params *params,
                      uint32_t first_strip, uint32_t strip_stride)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (strip_stride == 0u || first_strip >= strip_stride || strip_stride > 0x0fffffffu)
        return set_error(RWR_ERR_INVALID_ARGUMENT, "strips %u, %u + %u, ...: the first strip must be below the stride", first_strip, first_strip, strip_stride);
    const uint32_t h = ctx->screen.height, first_row = first_strip * kStripRows;
    return render_frame(ctx, camera, params, std::min(first_row, h), h, strip_stride * kStripRows);
}

// A frame rendered by the fused frame kernel is complete unless one of its waves waited in vain for the records (the wait is
// bounded; it has never been seen to run out): slot `i` is idle when this is called.
static int check_fused_frame(rwr_context *ctx, uint32_t i)
{
    FrameSlot &sl = ctx->slots[i];
    if (!sl.fused_used || !sl.d_fused.ptr) return RWR_OK;
    uint32_t timed_out = 0;
    RWR_HIP_CHECK(hipMemcpy(&timed_out, sl.d_fused.ptr + 1, sizeof timed_out, hipMemcpyDeviceToHost));
    if (timed_out) {
        sl.fused_blocks = 0;   // the count is no longer what the host expects: start over with the next frame
        return set_error(RWR_ERR_HIP, "a frame is incomplete: workgroups of the fused frame kernel waited in vain for the frame's records "
                         "(RWR_FUSED_SETUP=0 renders with two launches per frame)");
    }
    return RWR_OK;
}

int rwr_synchronize(rwr_context *ctx)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(sync_all(ctx));
    for (uint32_t i = 0; i < ctx->n_slots; i++) {
        const int rc = check_fused_frame(ctx, i);
        if (rc != RWR_OK) return rc;
    }
    return RWR_OK;
}

int rwr_readback(rwr_context *ctx, uint8_t *rgba8, float *depth, float *rgba_f32, int32_t *obj_id, float *hit_t)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (ctx->screen.width == 0) return set_error(RWR_ERR_NOT_READY, "rwr_resize has not been called");
    if ((rgba_f32 || obj_id || hit_t) && !ctx->slots[ctx->cur].aux_valid)
        return set_error(RWR_ERR_NOT_READY, "aux planes requested but the last render did not set RWR_FLAG_AUX_OUTPUTS");
    DeviceGuard g(ctx->device);
    const size_t n = (size_t)ctx->screen.width * ctx->screen.height;
    RWR_HIP_CHECK(hipStreamSynchronize(ctx->slots[ctx->cur].stream));
    {
        const int rc = check_fused_frame(ctx, ctx->cur);
        if (rc != RWR_OK) return rc;
    }
    if (rgba8) RWR_HIP_CHECK(hipMemcpy(rgba8, ctx->slots[ctx->cur].d_color.ptr, n * 4, hipMemcpyDeviceToHost));
    if (depth) RWR_HIP_CHECK(hipMemcpy(depth, ctx->slots[ctx->cur].d_depth.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
    if (rgba_f32) RWR_HIP_CHECK(hipMemcpy(rgba_f32, ctx->slots[ctx->cur].d_color_f32.ptr, n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    if (obj_id) RWR_HIP_CHECK(hipMemcpy(obj_id, ctx->slots[ctx->cur].d_obj_id.ptr, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (hit_t) RWR_HIP_CHECK(hipMemcpy(hit_t, ctx->slots[ctx->cur].d_hit_t.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
    return RWR_OK;
}

int rwr_get_device_targets(rwr_context *ctx, void **d_rgba8, void **d_depth)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (ctx->screen.width == 0) return set_error(RWR_ERR_NOT_READY, "rwr_resize has not been called");
    if (ctx->cur != 0u) {
        // the frame rendered last ran on an internal stream: order the context's stream (rwr_ctx_get_stream) after it,
        // so that whatever the caller enqueues there next sees the finished targets
        DeviceGuard g(ctx->device);
        FrameSlot &sl = ctx->slots[ctx->cur];
        RWR_HIP_CHECK(hipEventRecord(sl.done, sl.stream));
        RWR_HIP_CHECK(hipStreamWaitEvent(ctx->stream, sl.done, 0));
    }
    if (d_rgba8) *d_rgba8 = ctx->slots[ctx->cur].d_color.ptr;
    if (d_depth) *d_depth = ctx->slots[ctx->cur].d_depth.ptr;
    return RWR_OK;
}

int rwr_ctx_set_frames_in_flight(rwr_context *ctx, uint32_t n)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n < 1u || n > kMaxFramesInFlight)
        return set_error(RWR_ERR_INVALID_ARGUMENT, "frames in flight must be 1..%u", kMaxFramesInFlight);
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(sync_all(ctx));
    for (uint32_t i = 1; i < n; i++) {
        FrameSlot &sl = ctx->slots[i];
        if (!sl.owned) RWR_HIP_CHECK(hipStreamCreateWithFlags(&sl.owned, hipStreamNonBlocking));
        if (!sl.done) RWR_HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        sl.stream = sl.owned;
    }
    for (uint32_t i = n; i < kMaxFramesInFlight; i++) {  // slots no longer used give their memory back
        if (i == 0) continue;
        ctx->slots[i].release_buffers();
        ctx->slots[i].aux_valid = false;
        ctx->wf_state[i].release();   // (gigabytes of ray queue when the slot rendered path-traced frames)
        ctx->gather[i].release();
    }
    if (ctx->last_gather >= n) ctx->last_gather = 0;
    if (ctx->last_wf_state >= n) { ctx->last_wf_state = 0; ctx->last_segments = 0; ctx->last_spp = 0; }
    // the most recent frame stays where it is if its slot survives, otherwise it is gone
    ctx->n_slots = n;
    if (ctx->cur >= n) ctx->cur = 0;
    for (uint32_t i = 0; i < n; i++)
        if (!ctx->slots[i].d_color.ptr) RWR_HIP_CHECK(ensure_slot_targets(ctx, i));
    RWR_HIP_CHECK(ensure_frame_buffers(ctx));
    RWR_HIP_CHECK(sync_all(ctx));
    return RWR_OK;
}

int rwr_timer_begin(rwr_context *ctx)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    DeviceGuard g(ctx->device);
    if (ctx->n_slots > 1u) RWR_HIP_CHECK(sync_all(ctx));  // the interval starts with nothing in flight
    RWR_HIP_CHECK(hipEventRecord(ctx->ev_begin, ctx->stream));
    return RWR_OK;
}

int rwr_timer_end(rwr_context *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    DeviceGuard g(ctx->device);
    for (uint32_t i = 1; i < ctx->n_slots; i++) {  // the interval ends when every frame in flight has ended
        RWR_HIP_CHECK(hipEventRecord(ctx->slots[i].done, ctx->slots[i].stream));
        RWR_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->slots[i].done, 0));
    }
    RWR_HIP_CHECK(hipEventRecord(ctx->ev_end, ctx->stream));
    RWR_HIP_CHECK(hipEventSynchronize(ctx->ev_end));
    RWR_HIP_CHECK(hipEventElapsedTime(elapsed_ms, ctx->ev_begin, ctx->ev_end));
    return RWR_OK;
}

int rwr_timer_stop(rwr_context *ctx)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    DeviceGuard g(ctx->device);
    for (uint32_t i = 1; i < ctx->n_slots; i++) {
        RWR_HIP_CHECK(hipEventRecord(ctx->slots[i].done, ctx->slots[i].stream));
        RWR_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->slots[i].done, 0));
    }
    RWR_HIP_CHECK(hipEventRecord(ctx->ev_end, ctx->stream));
    return RWR_OK;
}

int rwr_timer_elapsed(rwr_context *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(hipEventSynchronize(ctx->ev_end));
    RWR_HIP_CHECK(hipEventElapsedTime(elapsed_ms, ctx->ev_begin, ctx->ev_end));
    return RWR_OK;
}

int rwr_ctx_set_kernel_timing(rwr_context *ctx, uint32_t every_n)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    ctx->timing_every = every_n;
    ctx->timing_calls = 0;
    ctx->timing_pairs = 0;
    return RWR_OK;
}

int rwr_kernel_timing_stats(rwr_context *ctx, double *mean_us, uint32_t *count)
{
    if (!ctx || !mean_us || !count) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(sync_all(ctx));
    double sum = 0.0;
    for (uint32_t i = 0; i < ctx->timing_pairs; i++) {
        float ms = 0.0f;
        RWR_HIP_CHECK(hipEventElapsedTime(&ms, ctx->timing_events[2 * i], ctx->timing_events[2 * i + 1]));
        sum += ms * 1e3;
    }
    *count = ctx->timing_pairs;
    *mean_us = ctx->timing_pairs ? sum / ctx->timing_pairs : 0.0;
    return RWR_OK;
}

int rwr_selftest_exact_math(rwr_context *ctx, uint32_t normalize_count, uint32_t seed, uint64_t out4[4])
{
    if (!ctx || !out4) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    DeviceGuard g(ctx->device);
    struct Scoped {
        DeviceBuffer<unsigned long long> b;
        ~Scoped() { b.release(); }
    } scoped;
    DeviceBuffer<unsigned long long> &d_out = scoped.b;
    RWR_HIP_CHECK(d_out.ensure(4));
    RWR_HIP_CHECK(hipMemsetAsync(d_out.ptr, 0, 4 * sizeof(unsigned long long), ctx->stream));
    RWR_HIP_CHECK(launch_selftest_exact_math(ctx->stream, d_out.ptr, normalize_count, seed));
    unsigned long long h[4];
    RWR_HIP_CHECK(hipMemcpyAsync(h, d_out.ptr, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    RWR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; k++) out4[k] = h[k];
    return RWR_OK;
}

int rwr_measure_valu_clock(rwr_context *ctx, uint32_t waves_per_simd, double out4[4])
{
    if (!ctx || !out4) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (waves_per_simd < 1u || waves_per_simd > 8u) return set_error(RWR_ERR_INVALID_ARGUMENT, "waves_per_simd must be 1..8");
    DeviceGuard g(ctx->device);
    hipDeviceProp_t prop;
    RWR_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
    const uint32_t n_wg = (uint32_t)prop.multiProcessorCount * waves_per_simd, n_waves = n_wg * 4u, iters = 1u << 15;
    struct Scoped {
        DeviceBuffer<ulonglong2> b;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Scoped() { b.release(); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } scoped;
    RWR_HIP_CHECK(scoped.b.ensure(n_waves));
    RWR_HIP_CHECK(hipEventCreate(&scoped.e0));
    RWR_HIP_CHECK(hipEventCreate(&scoped.e1));
    std::vector<ulonglong2> h(n_waves);
    RWR_HIP_CHECK(sync_all(ctx));
    for (int mode = 0; mode < 2; mode++) {
        // an untimed launch first: the stamped one then starts on a busy, clocked-up chip
        RWR_HIP_CHECK(launch_measure_valu(ctx->stream, mode, scoped.b.ptr, n_wg, iters));
        RWR_HIP_CHECK(hipEventRecord(scoped.e0, ctx->stream));
        RWR_HIP_CHECK(launch_measure_valu(ctx->stream, mode, scoped.b.ptr, n_wg, iters));
        RWR_HIP_CHECK(hipEventRecord(scoped.e1, ctx->stream));
        RWR_HIP_CHECK(hipMemcpyAsync(h.data(), scoped.b.ptr, n_waves * sizeof(ulonglong2), hipMemcpyDeviceToHost, ctx->stream));
        RWR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        float ms = 0.0f;
        RWR_HIP_CHECK(hipEventElapsedTime(&ms, scoped.e0, scoped.e1));
        std::vector<double> mhz(n_waves);
        for (uint32_t i = 0; i < n_waves; i++) mhz[i] = h[i].y ? (double)h[i].x / (double)h[i].y * 100.0 : 0.0;
        std::nth_element(mhz.begin(), mhz.begin() + n_waves / 2, mhz.end());
        const double clock_mhz = mhz[n_waves / 2];
        // every SIMD issued (waves on it) * iters * 8 wave instructions during the launch (HIP events around it);
        // cycles = elapsed time x the in-kernel clock
        const double instr_per_simd = (double)n_waves / (4.0 * prop.multiProcessorCount) * iters * 8.0;
        const double per_instr = (double)ms * 1e-3 * clock_mhz * 1e6 / instr_per_simd;
        if (mode == 0) { out4[0] = clock_mhz; out4[1] = per_instr; }
        else { out4[2] = per_instr; out4[3] = clock_mhz; }
    }
    return RWR_OK;
}

int rwr_clock_probe_start(rwr_context *ctx, uint32_t micros)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (micros == 0u || micros > 100000u) return set_error(RWR_ERR_INVALID_ARGUMENT, "probe duration must be 1..100000 us");
    DeviceGuard g(ctx->device);
    if (!ctx->probe_stream) RWR_HIP_CHECK(hipStreamCreateWithFlags(&ctx->probe_stream, hipStreamNonBlocking));
    RWR_HIP_CHECK(ctx->d_probe.ensure(1));
    RWR_HIP_CHECK(hipStreamSynchronize(ctx->probe_stream));
    RWR_HIP_CHECK(launch_clock_probe(ctx->probe_stream, ctx->d_probe.ptr, micros * 100u));
    return RWR_OK;
}

int rwr_clock_probe_read(rwr_context *ctx, double *shader_mhz)
{
    if (!ctx || !shader_mhz) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!ctx->probe_stream) return set_error(RWR_ERR_NOT_READY, "rwr_clock_probe_start has not been called");
    DeviceGuard g(ctx->device);
    ulonglong2 h{0, 0};
    RWR_HIP_CHECK(hipMemcpyAsync(&h, ctx->d_probe.ptr, sizeof h, hipMemcpyDeviceToHost, ctx->probe_stream));
    RWR_HIP_CHECK(hipStreamSynchronize(ctx->probe_stream));
    *shader_mhz = h.y ? (double)h.x / (double)h.y * 100.0 : 0.0;
    return RWR_OK;
}

int rwr_last_render_stats(rwr_context *ctx, uint64_t *primary_rays, uint64_t *bounce_rays)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (ctx->last_spp && ctx->last_had_bounce) {
        DeviceGuard g(ctx->device);
        std::vector<uint32_t> counts((size_t)ctx->last_segments * 4u);   // per tile and wave of the primary stage
        RWR_HIP_CHECK(hipStreamSynchronize(ctx->slots[ctx->cur].stream));
        RWR_HIP_CHECK(hipMemcpy(counts.data(), ctx->wf_state[ctx->last_wf_state].d_wave_total.ptr, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        ctx->last_bounce = 0;
        for (uint32_t c : counts) ctx->last_bounce += c;
    }
    if (primary_rays) *primary_rays = ctx->last_primary;
    if (bounce_rays) *bounce_rays = ctx->last_bounce;
    return RWR_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Multi-GPU frames: one process (and one context) per GPU, the frame cut into contiguous row bands, ONE gather of
// the finished RGBA8 bands to the root per frame — RCCL point-to-point sends grouped into a single operation, over
// xGMI.  RCCL is bound at run time: a single-GPU host never needs it.
namespace {
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    void *lib = nullptr;
};
RcclApi g_rccl;

int load_rccl()
{
    if (g_rccl.lib) return RWR_OK;
    void *lib = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return set_error(RWR_ERR_UNSUPPORTED, "RCCL is not available: %s", dlerror());
#define RWR_SYM(field, sym)                                                                           \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, sym));                         \
    if (!g_rccl.field) return set_error(RWR_ERR_UNSUPPORTED, "RCCL symbol %s is missing", sym);
    RWR_SYM(GetUniqueId, "ncclGetUniqueId") RWR_SYM(CommInitRank, "ncclCommInitRank") RWR_SYM(CommDestroy, "ncclCommDestroy")
    RWR_SYM(GroupStart, "ncclGroupStart") RWR_SYM(GroupEnd, "ncclGroupEnd") RWR_SYM(Send, "ncclSend") RWR_SYM(Recv, "ncclRecv")
    RWR_SYM(AllReduce, "ncclAllReduce") RWR_SYM(GetErrorString, "ncclGetErrorString")
#undef RWR_SYM
    g_rccl.lib = lib;
    return RWR_OK;
}

#define RWR_NCCL_CHECK(expr)                                                                                     \
    do {                                                                                                         \
        ncclResult_t _r = (expr);                                                                                \
        if (_r != ncclSuccess)                                                                                   \
            return set_error(RWR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)
}  // namespace

int rwr_dist_band(uint32_t rank, uint32_t world, uint32_t height, uint32_t *row_begin, uint32_t *row_end)
{
    if (!row_begin || !row_end || world == 0u || rank >= world)
        return set_error(RWR_ERR_INVALID_ARGUMENT, "rank %u outside world %u", rank, world);
    *row_begin = (uint32_t)(((uint64_t)rank * height) / world);
    *row_end = (uint32_t)((((uint64_t)rank + 1u) * height) / world);
    return RWR_OK;
}

int rwr_dist_strip_layout(uint32_t rank, uint32_t world, uint32_t height, rwr_strip_layout *out)
{
    if (!out || world == 0u || rank >= world) return set_error(RWR_ERR_INVALID_ARGUMENT, "rank %u outside world %u", rank, world);
    const StripLayout L = StripLayout::make(height, world);
    out->n_strips = L.n_strips;
    out->strips = L.strips_of(rank);
    out->rows = L.rows_of(rank);
    out->recv_row = L.recv_row(rank);
    out->recv_rows_total = L.recv_rows_total();
    out->owns_tail = L.owns_tail(rank) ? 1u : 0u;
    return RWR_OK;
}

int rwr_dist_host_pack_strips(uint32_t rank, uint32_t world, uint32_t width, uint32_t height, const uint8_t *frame_rgba8, uint8_t *message)
{
    if (!frame_rgba8 || !message || world == 0u || rank >= world) return set_error(RWR_ERR_INVALID_ARGUMENT, "bad argument (rank %u, world %u)", rank, world);
    strips_pack_host(StripLayout::make(height, world), rank, (size_t)width * 4u, frame_rgba8, message);
    return RWR_OK;
}

int rwr_dist_host_deal_strips(uint32_t world, uint32_t width, uint32_t height, const uint8_t *recv, uint8_t *frame_rgba8)
{
    if (!frame_rgba8 || !recv || world == 0u) return set_error(RWR_ERR_INVALID_ARGUMENT, "bad argument (world %u)", world);
    strips_deal_host(StripLayout::make(height, world), (size_t)width * 4u, recv, frame_rgba8);
    return RWR_OK;
}

int rwr_dist_get_unique_id(uint8_t id[RWR_DIST_ID_BYTES])
{
    if (!id) return set_error(RWR_ERR_INVALID_ARGUMENT, "id is NULL");
    static_assert(RWR_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "RWR_DIST_ID_BYTES is RCCL's unique id size");
    const int rc = load_rccl();
    if (rc != RWR_OK) return rc;
    ncclUniqueId uid;
    RWR_NCCL_CHECK(g_rccl.GetUniqueId(&uid));
    std::memcpy(id, uid.internal, RWR_DIST_ID_BYTES);
    return RWR_OK;
}

int rwr_dist_init(rwr_context *ctx, int rank, int world, const uint8_t id[RWR_DIST_ID_BYTES])
{
    if (!ctx || !id) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return set_error(RWR_ERR_INVALID_ARGUMENT, "rank %d outside world %d", rank, world);
    if (ctx->comm) return set_error(RWR_ERR_INVALID_ARGUMENT, "the context already has a communicator");
    const int rc = load_rccl();
    if (rc != RWR_OK) return rc;
    DeviceGuard g(ctx->device);
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, RWR_DIST_ID_BYTES);
    RWR_NCCL_CHECK(g_rccl.CommInitRank(&ctx->comm, world, uid, rank));
    ctx->dist_rank = rank;
    ctx->dist_world = world;
    return RWR_OK;
}

}  // extern "C"

// ---- the gather, in stages --------------------------------------------------------------------------------------------
// Every frame slot owns a *gather set* (message, receive buffer, assembled frame, completion event), so the gather of one
// frame shares nothing with the frame rendered next in another slot: pack -> exchange -> deal-out of frame n run beside
// the render of frame n + 1.  The stages below are what BOTH the RCCL gather and the one-GPU loopback self-test run; the two
// differ in the exchange step alone (ncclSend/ncclRecv against a device copy to the same address).
namespace {

hipError_t gather_set_events(rwr_context *ctx, rwr_context::GatherSet &gs)
{
    if (!gs.done) return hipEventCreateWithFlags(&gs.done, hipEventDisableTiming);
    return hipSuccess;
}

// -- interleaved strips
int strips_stage_pack(rwr_context *ctx, rwr_context::GatherSet &gs, const FrameSlot &sl, const StripLayout &L, uint32_t me, hipStream_t stream)
{
    const uint32_t row_bytes = ctx->screen.width * 4u;
    RWR_HIP_CHECK(gs.d_pack.ensure((size_t)std::max(1u, L.strips_of(me) * kStripRows) * row_bytes));
    RWR_HIP_CHECK(launch_strips_pack(stream, L, me, row_bytes, sl.d_color.ptr, gs.d_pack.ptr));
    return RWR_OK;
}
int strips_stage_root_buffers(rwr_context *ctx, rwr_context::GatherSet &gs, const StripLayout &L)
{
    const size_t row_bytes = (size_t)ctx->screen.width * 4u;
    RWR_HIP_CHECK(gs.d_gathered.ensure(row_bytes * L.height));
    RWR_HIP_CHECK(gs.d_recv.ensure(row_bytes * std::max(1u, L.recv_rows_total())));
    return RWR_OK;
}
// where rank r's message lands in the root's receive buffer, and its size
uint8_t *strips_recv_at(rwr_context *ctx, rwr_context::GatherSet &gs, const StripLayout &L, uint32_t r) { return gs.d_recv.ptr + (size_t)L.recv_row(r) * ctx->screen.width * 4u; }
size_t strips_message_bytes(rwr_context *ctx, const StripLayout &L, uint32_t r) { return (size_t)L.rows_of(r) * ctx->screen.width * 4u; }
int strips_stage_deal(rwr_context *ctx, rwr_context::GatherSet &gs, const StripLayout &L, hipStream_t stream)
{
    RWR_HIP_CHECK(launch_strips_deal(stream, L, ctx->screen.width * 4u, gs.d_recv.ptr, gs.d_gathered.ptr));
    return RWR_OK;
}

// -- contiguous bands: they are sent from the frame and land in place, no pack, no deal-out
void band_span(rwr_context *ctx, uint32_t r, uint32_t world, size_t *offset, size_t *bytes)
{
    uint32_t a = 0, b = 0;
    (void)rwr_dist_band(r, world, ctx->screen.height, &a, &b);
    const size_t row_bytes = (size_t)ctx->screen.width * 4u;
    *offset = (size_t)a * row_bytes;
    *bytes = (size_t)(b - a) * row_bytes;
}

// One grouped RCCL exchange: `send` (may be empty) to the root; on the root one receive per rank with a non-empty
// message, at recv_at(r).  The group is closed on every path.
template <typename RecvAt, typename RecvBytes>
int rccl_gather_exchange(rwr_context *ctx, int root, const void *send, size_t send_bytes, RecvAt recv_at, RecvBytes recv_bytes, hipStream_t stream)
{
    ncclResult_t res = g_rccl.GroupStart();
    if (res != ncclSuccess) return set_error(RWR_ERR_HIP, "ncclGroupStart failed: %s", g_rccl.GetErrorString(res));
    if (send_bytes) res = g_rccl.Send(send, send_bytes, ncclUint8, root, ctx->comm, stream);
    if (ctx->dist_rank == root)
        for (int r = 0; r < ctx->dist_world && res == ncclSuccess; r++)
            if (recv_bytes((uint32_t)r)) res = g_rccl.Recv(recv_at((uint32_t)r), recv_bytes((uint32_t)r), ncclUint8, r, ctx->comm, stream);
    const ncclResult_t end = g_rccl.GroupEnd();   // also after a failed send / receive: a group must not stay open
    if (res == ncclSuccess) res = end;
    if (res != ncclSuccess) return set_error(RWR_ERR_HIP, "RCCL gather failed: %s", g_rccl.GetErrorString(res));
    return RWR_OK;
}

int gather_checks(rwr_context *ctx, int root, bool need_comm)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (need_comm) {
        if (!ctx->comm) return set_error(RWR_ERR_NOT_READY, "rwr_dist_init has not been called");
        if (root < 0 || root >= ctx->dist_world) return set_error(RWR_ERR_INVALID_ARGUMENT, "root %d outside world %d", root, ctx->dist_world);
    }
    if (ctx->screen.width == 0) return set_error(RWR_ERR_NOT_READY, "rwr_resize has not been called");
    return RWR_OK;
}

// Consecutive exchanges of one communicator run on different frame slots' streams: they are ordered among themselves
// (RCCL operations of a communicator are issued in one order on every rank), the rest of a slot's work is not held back.
int order_exchange_begin(rwr_context *ctx, hipStream_t stream)
{
    if (!ctx->exchange_done) RWR_HIP_CHECK(hipEventCreateWithFlags(&ctx->exchange_done, hipEventDisableTiming));
    else if (ctx->n_slots > 1u) RWR_HIP_CHECK(hipStreamWaitEvent(stream, ctx->exchange_done, 0));
    return RWR_OK;
}
}  // namespace

extern "C" {

int rwr_dist_gather_rgba8(rwr_context *ctx, int root)
{
    int rc = gather_checks(ctx, root, true);
    if (rc != RWR_OK) return rc;
    DeviceGuard g(ctx->device);
    FrameSlot &sl = ctx->slots[ctx->cur];
    rwr_context::GatherSet &gs = ctx->gather[ctx->cur];
    const hipStream_t stream = sl.stream;   // the frame rendered last: the exchange follows it in stream order
    const bool is_root = ctx->dist_rank == root;
    const uint32_t world = (uint32_t)ctx->dist_world;
    RWR_HIP_CHECK(gather_set_events(ctx, gs));
    if (is_root) RWR_HIP_CHECK(gs.d_gathered.ensure((size_t)ctx->screen.width * 4u * ctx->screen.height));
    size_t my_off = 0, my_bytes = 0;
    band_span(ctx, (uint32_t)ctx->dist_rank, world, &my_off, &my_bytes);
    if ((rc = order_exchange_begin(ctx, stream)) != RWR_OK) return rc;
    rc = rccl_gather_exchange(ctx, root, sl.d_color.ptr + my_off, my_bytes,
                              [&](uint32_t r) { size_t o, b; band_span(ctx, r, world, &o, &b); return gs.d_gathered.ptr + o; },   // bands land in final image order
                              [&](uint32_t r) { size_t o, b; band_span(ctx, r, world, &o, &b); return b; }, stream);
    if (rc != RWR_OK) return rc;
    RWR_HIP_CHECK(hipEventRecord(ctx->exchange_done, stream));
    RWR_HIP_CHECK(hipEventRecord(gs.done, stream));
    gs.valid = is_root;
    ctx->last_gather = ctx->cur;
    return RWR_OK;
}

// The interleaved partition (rwr_render_strips(ctx, ..., rank, world)): rank r owns strips r, r + world, ...  Every rank packs
// its strips into one contiguous message (one launch), the root receives the messages side by side and deals the strips out
// into the frame (one launch): still ONE grouped RCCL exchange per frame.  Layout: rwr_strips.h.
int rwr_dist_gather_strips_rgba8(rwr_context *ctx, int root)
{
    int rc = gather_checks(ctx, root, true);
    if (rc != RWR_OK) return rc;
    DeviceGuard g(ctx->device);
    FrameSlot &sl = ctx->slots[ctx->cur];
    rwr_context::GatherSet &gs = ctx->gather[ctx->cur];
    const hipStream_t stream = sl.stream;   // the frame rendered last: the exchange follows it in stream order
    const StripLayout L = StripLayout::make(ctx->screen.height, (uint32_t)ctx->dist_world);
    const bool is_root = ctx->dist_rank == root;
    const uint32_t me = (uint32_t)ctx->dist_rank;
    RWR_HIP_CHECK(gather_set_events(ctx, gs));
    if (is_root && (rc = strips_stage_root_buffers(ctx, gs, L)) != RWR_OK) return rc;
    if ((rc = strips_stage_pack(ctx, gs, sl, L, me, stream)) != RWR_OK) return rc;
    if ((rc = order_exchange_begin(ctx, stream)) != RWR_OK) return rc;
    rc = rccl_gather_exchange(ctx, root, gs.d_pack.ptr, strips_message_bytes(ctx, L, me),
                              [&](uint32_t r) { return strips_recv_at(ctx, gs, L, r); },
                              [&](uint32_t r) { return strips_message_bytes(ctx, L, r); }, stream);
    if (rc != RWR_OK) return rc;
    RWR_HIP_CHECK(hipEventRecord(ctx->exchange_done, stream));
    if (is_root && (rc = strips_stage_deal(ctx, gs, L, stream)) != RWR_OK) return rc;
    RWR_HIP_CHECK(hipEventRecord(gs.done, stream));
    gs.valid = is_root;
    ctx->last_gather = ctx->cur;
    return RWR_OK;
}

// One-GPU self-test of the stages above for ANY world size: the context plays every rank in turn.  After
// rwr_render_strips(ctx, ..., rank, world) (or rwr_render_rows of rank's band) _deposit runs that rank's side of the gather
// on the frame just rendered — the same pack launch, the same message size, the same receive address — with one device
// copy standing in for the ncclSend / ncclRecv pair; after the last rank _finish runs the root's side (the same deal-out
// launch).  rwr_dist_frame / rwr_dist_readback then return what a root would hold.  Needs no communicator.
int rwr_dist_loopback_deposit(rwr_context *ctx, uint32_t rank, uint32_t world, int strips)
{
    int rc = gather_checks(ctx, 0, false);
    if (rc != RWR_OK) return rc;
    if (world == 0u || rank >= world) return set_error(RWR_ERR_INVALID_ARGUMENT, "rank %u outside world %u", rank, world);
    DeviceGuard g(ctx->device);
    FrameSlot &sl = ctx->slots[ctx->cur];
    rwr_context::GatherSet &gs = ctx->gather[0];   // one "root": every deposit lands in the same set
    const hipStream_t stream = sl.stream;
    const bool first = gs.done == nullptr;
    RWR_HIP_CHECK(gather_set_events(ctx, gs));
    if (!first) RWR_HIP_CHECK(hipStreamWaitEvent(stream, gs.done, 0));   // deposits share the set's message buffer
    if (strips) {
        const StripLayout L = StripLayout::make(ctx->screen.height, world);
        if ((rc = strips_stage_root_buffers(ctx, gs, L)) != RWR_OK) return rc;
        if ((rc = strips_stage_pack(ctx, gs, sl, L, rank, stream)) != RWR_OK) return rc;
        if (strips_message_bytes(ctx, L, rank))
            RWR_HIP_CHECK(hipMemcpyAsync(strips_recv_at(ctx, gs, L, rank), gs.d_pack.ptr, strips_message_bytes(ctx, L, rank), hipMemcpyDeviceToDevice, stream));
    } else {
        RWR_HIP_CHECK(gs.d_gathered.ensure((size_t)ctx->screen.width * 4u * ctx->screen.height));
        size_t off = 0, bytes = 0;
        band_span(ctx, rank, world, &off, &bytes);
        if (bytes) RWR_HIP_CHECK(hipMemcpyAsync(gs.d_gathered.ptr + off, sl.d_color.ptr + off, bytes, hipMemcpyDeviceToDevice, stream));
    }
    RWR_HIP_CHECK(hipEventRecord(gs.done, stream));
    gs.valid = false;
    return RWR_OK;
}

int rwr_dist_loopback_finish(rwr_context *ctx, uint32_t world, int strips)
{
    int rc = gather_checks(ctx, 0, false);
    if (rc != RWR_OK) return rc;
    if (world == 0u) return set_error(RWR_ERR_INVALID_ARGUMENT, "world is 0");
    rwr_context::GatherSet &gs = ctx->gather[0];
    if (!gs.done || !gs.d_gathered.ptr) return set_error(RWR_ERR_NOT_READY, "nothing has been deposited");
    DeviceGuard g(ctx->device);
    const hipStream_t stream = ctx->slots[ctx->cur].stream;
    RWR_HIP_CHECK(hipStreamWaitEvent(stream, gs.done, 0));
    if (strips && (rc = strips_stage_deal(ctx, gs, StripLayout::make(ctx->screen.height, world), stream)) != RWR_OK) return rc;
    RWR_HIP_CHECK(hipEventRecord(gs.done, stream));
    gs.valid = true;
    ctx->last_gather = 0;
    return RWR_OK;
}

int rwr_dist_frame(rwr_context *ctx, void **d_rgba8)
{
    if (!ctx || !d_rgba8) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    rwr_context::GatherSet &gs = ctx->gather[ctx->last_gather];
    if (!gs.valid) return set_error(RWR_ERR_NOT_READY, "no gathered frame on this rank (rwr_dist_gather_rgba8 on the root)");
    *d_rgba8 = gs.d_gathered.ptr;
    return RWR_OK;
}

int rwr_dist_readback(rwr_context *ctx, uint8_t *rgba8)
{
    if (!ctx || !rgba8) return set_error(RWR_ERR_INVALID_ARGUMENT, "NULL argument");
    rwr_context::GatherSet &gs = ctx->gather[ctx->last_gather];
    if (!gs.valid) return set_error(RWR_ERR_NOT_READY, "no gathered frame on this rank (rwr_dist_gather_rgba8 on the root)");
    DeviceGuard g(ctx->device);
    RWR_HIP_CHECK(hipEventSynchronize(gs.done));
    RWR_HIP_CHECK(hipMemcpy(rgba8, gs.d_gathered.ptr, (size_t)ctx->screen.width * ctx->screen.height * 4u, hipMemcpyDeviceToHost));
    return RWR_OK;
}

int rwr_dist_barrier(rwr_context *ctx)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (!ctx->comm) return set_error(RWR_ERR_NOT_READY, "rwr_dist_init has not been called");
    DeviceGuard g(ctx->device);
    struct Scoped {
        DeviceBuffer<uint32_t> b;
        ~Scoped() { b.release(); }
    } scoped;
    RWR_HIP_CHECK(scoped.b.ensure(1));
    RWR_HIP_CHECK(sync_all(ctx));
    RWR_HIP_CHECK(hipMemsetAsync(scoped.b.ptr, 0, sizeof(uint32_t), ctx->stream));
    RWR_NCCL_CHECK(g_rccl.AllReduce(scoped.b.ptr, scoped.b.ptr, 1, ncclUint32, ncclSum, ctx->comm, ctx->stream));
    RWR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RWR_OK;
}

int rwr_dist_destroy(rwr_context *ctx)
{
    if (!ctx) return set_error(RWR_ERR_INVALID_ARGUMENT, "ctx is NULL");
    DeviceGuard g(ctx->device);
    (void)sync_all(ctx);
    if (ctx->comm) {
        (void)g_rccl.CommDestroy(ctx->comm);
        ctx->comm = nullptr;
    }
    if (ctx->exchange_done) { (void)hipEventDestroy(ctx->exchange_done); ctx->exchange_done = nullptr; }
    for (rwr_context::GatherSet &gs : ctx->gather) gs.release();
    ctx->last_gather = 0;
    ctx->dist_world = 0;
    return RWR_OK;
}

}  // extern "C"
