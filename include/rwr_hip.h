/*
 * rwr_hip.h — C ABI of the MI355X-native ray/path tracer (librwr_hip.so).
 *
 * This is the drop-in boundary for the reference's hot path
 * (clejacquet/rust-wgpu-raytracing): everything `State::render` hands to its
 * three WGSL compute passes through wgpu bind groups crosses this ABI as plain
 * pointers and sizes instead.  The reference has no FFI of its own; each entry
 * point below cites the reference interface it replaces (paths relative to
 * /root/reference/).  The Rust binding a maintainer would add is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - every function returns RWR_OK (0) or a negative rwr_status; the message
 *     for the calling thread's last failure is rwr_last_error_string().
 *     Nothing aborts or panics (the reference unwrap()s: lib.rs:275,284,303,566).
 *   - one context = one GPU = one caller thread (the reference's State is !Send,
 *     lib.rs:256).  Rendering is asynchronous on the context's HIP stream(s);
 *     rwr_readback() waits for the frame rendered last, rwr_synchronize() for
 *     every frame in flight (rwr_ctx_set_frames_in_flight).
 *   - caller owns host arrays; uploads copy; the context owns device memory
 *     until rwr_ctx_destroy().
 *   - all PODs are layout-identical to the reference's #[repr(C)] structs.
 *   - framebuffer row 0 is the BOTTOM image row (y_nds grows with the row
 *     index, compute.wgsl:152; the reference's blit compensates, lib.rs:39-64).
 */
#ifndef RWR_HIP_H
#define RWR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define RWR_API __attribute__((visibility("default")))
#else
#define RWR_API
#endif

typedef enum rwr_status {
    RWR_OK = 0,
    RWR_ERR_INVALID_ARGUMENT = -1,
    RWR_ERR_HIP = -2,            /* a HIP runtime call failed (no device, OOM, launch failure) */
    RWR_ERR_NOT_READY = -3,      /* render before resize / scene upload */
    RWR_ERR_IO = -4,             /* loader: file missing / unreadable   (anyhow::Error in resources.rs) */
    RWR_ERR_PARSE = -5,          /* loader: malformed OBJ/MTL/PNG/JPEG */
    RWR_ERR_UNSUPPORTED = -6
} rwr_status;

/* ------------------------------------------------------------------ PODs -- */

/* CameraInvUniform, src/lib.rs:86-93 — binding g0 b3 (compute.wgsl:10-14,57-58).
 * Matrices are column-major: m[col][row] (cgmath `Into<[[f32;4];4]>`). */
typedef struct rwr_camera_inv_uniform {
    float viewmodel_inv[4][4];
    float proj_inv[4][4];       /* = OPENGL_TO_WGPU_MATRIX * perspective^-1, lib.rs:109 */
    float origin[3];
    uint32_t _padding;
} rwr_camera_inv_uniform;

/* Screen, src/lib.rs:216-221 — binding g0 b4 (compute.wgsl:16-19,60-61). */
typedef struct rwr_screen {
    uint32_t width, height;
} rwr_screen;

/* ModelVertexSmall, src/model.rs:45-63 — binding g0 b5 element (compute.wgsl:27-30,63-64). */
typedef struct rwr_model_vertex_small {
    float position[3];
    float pad0;
    float tex_coords[2];
    float pad1[2];
} rwr_model_vertex_small;

/* ModelFaceSmall, src/model.rs:65-79 — binding g0 b6 element (compute.wgsl:66-67). */
typedef struct rwr_model_face_small {
    uint32_t indices[3];
    uint32_t pad0;
} rwr_model_face_small;

/* MaterialData, src/models/triangle_list/triangle_list.rs:24-33 — binding g0 b7 (compute.wgsl:32-36,69-70). */
typedef struct rwr_material_data {
    float ambient[3];  float pad0;
    float diffuse[3];  float pad1;
    float specular[3]; float pad2;
} rwr_material_data;

/* SphereBufferData, src/models/sphere/sphere.rs:10-15 — sphere binding g0 b5 (sphere/compute.wgsl:21-24,49-50). */
typedef struct rwr_sphere_buffer_data {
    float center[3];
    float radius;
} rwr_sphere_buffer_data;

/* TriangleBufferData, src/models/triangle/triangle.rs:10-19 — the single-triangle model's uniform
 * (triangle/compute.wgsl:21-25,50-51).  The reference builds this model type but never dispatches it. */
typedef struct rwr_triangle_buffer_data {
    float p0[3]; float pad0;
    float p1[3]; float pad1;
    float p2[3]; float pad2;
} rwr_triangle_buffer_data;

/* InstanceRaw, src/lib.rs:129-134 (computed but never bound by the reference;
 * used here by the instanced configs).  Column-major model matrix; must be rigid
 * (rotation + translation). */
typedef struct rwr_instance_raw {
    float model[4][4];
} rwr_instance_raw;

/* Camera, src/camera.rs:3-11 (cgmath Point3/Vector3 flattened). */
typedef struct rwr_camera {
    float eye[3];
    float target[3];
    float up[3];
    float aspect;
    float fovy;   /* degrees */
    float znear;
    float zfar;
} rwr_camera;

/* Extension (no reference counterpart): how many samples / bounces to trace.
 * With spp > 1 or max_bounces > 0 a pixel is (sum over samples of E(h0) + albedo(h0) * E(h1)) / spp, E being the reference's
 * local shading.  Part of that definition: every term a sample adds is clamped per channel — E(h0) to
 * [0, 16], albedo * E(h1) to [0, 64], NaN counting as 0 — so that the sums have a fixed range (they are added as fixed
 * point, in any order, and the frame is bit-reproducible).  The reference's materials (components <= 1) come nowhere near;
 * a material with Ka = 20 is clamped, in the oracle's or_render_path as here.  spp 1 / max_bounces 0 is the reference frame:
 * nothing is clamped before the rgba8unorm store. */
typedef struct rwr_render_params {
    uint32_t spp;          /* >= 1.  1 = the reference's single centre sample        */
    uint32_t max_bounces;  /* 0 = reference (primary rays only); 1 = one diffuse bounce */
    uint32_t seed;         /* RNG stream key; results do not depend on GPU count     */
    uint32_t flags;        /* RWR_FLAG_*                                              */
} rwr_render_params;

enum {
    RWR_FLAG_AUX_OUTPUTS = 1u << 0, /* also produce float colour, object id and hit distance planes */
    RWR_FLAG_NO_CULL     = 1u << 1, /* debug: brute-force every face for every pixel (reference loop order) */
    RWR_FLAG_USE_BVH     = 1u << 2, /* reference frame only: the mesh pass traverses the BVH per ray instead of
                                       walking per-tile candidate lists (better when many small faces share a
                                       tile, e.g. a distant mesh); same result bit for bit.  The context picks
                                       this kernel by itself when a scene of more than 256 faces projects to
                                       faces far smaller than a tile */
    RWR_FLAG_ORTHO_RAYS  = 1u << 3, /* every pass generates its rays with pixelToRay_ortho (defined, never called,
                                       in all three shaders: triangle_list/compute.wgsl:166-174): origin =
                                       camera.origin + (5 x_nds, 5 y_nds, 0), direction (0, 0, -1).  Reference
                                       frame only (spp 1, no bounce, no RWR_FLAG_USE_BVH) */
    RWR_FLAG_NORMAL_MAP  = 1u << 4  /* extension: mesh hits are shaded with the normal of their part's normal map
                                       (rwr_scene_set_normal_map; res/cube.mtl:13 `map_Bump`, which the reference parses
                                       nowhere: resources.rs:187-213 loads the diffuse texture only and compute.wgsl:226-229
                                       shades with the flat face normal).  Without the flag — the default — the frame is
                                       the reference's.  Visibility never changes; bounce rays still leave along the
                                       geometric normal */
};

#define RWR_MAX_SPHERES 8
#define RWR_MAX_TRIANGLES 8

/* Object id plane encoding (aux output): >= 0 mesh face index
 * (instance * n_faces + face), -1 background, -2-k analytic sphere k, -10-k single triangle k. */

typedef struct rwr_context rwr_context;

/* ------------------------------------------------------ context / device -- */

/* Replaces wgpu Instance/Adapter/Device/Queue creation, src/lib.rs:266-303. */
RWR_API int rwr_ctx_create(int device_id, rwr_context **out_ctx);
RWR_API void rwr_ctx_destroy(rwr_context *ctx);
RWR_API const char *rwr_last_error_string(void);
RWR_API int rwr_device_count(int *out_count);
/* Name, CU count and wavefront size of the context's device. */
RWR_API int rwr_ctx_device_info(rwr_context *ctx, char *name, size_t name_cap, int *cu_count, int *wave_size);

/* Frames in flight (1..3, default 1).  With n > 1 the context owns n sets of targets and per-frame
 * buffers, each with its own HIP stream, and consecutive rwr_render calls take them in turn, the way a
 * swapchain hands out images (the reference takes each frame from wgpu's surface and presents it, lib.rs:1013,1227): the
 * next frame's kernels fill the GPU while the previous frame's last waves drain.  rwr_readback and
 * rwr_get_device_targets refer to the frame rendered last (rwr_readback waits for that frame only,
 * rwr_get_device_targets orders the context's stream after it); a frame's
 * targets stay valid until n further frames have been rendered.  rwr_synchronize, scene changes,
 * rwr_resize and rwr_ctx_set_stream wait for all frames in flight.  Slot 0 uses the context's stream
 * (rwr_ctx_set_stream), the others internal streams. */
RWR_API int rwr_ctx_set_frames_in_flight(rwr_context *ctx, uint32_t n);

/* Launch on a caller-owned hipStream_t (e.g. torch's current stream) instead of
 * the context's own stream.  NULL restores the context's stream. */
RWR_API int rwr_ctx_set_stream(rwr_context *ctx, void *hip_stream);
RWR_API void *rwr_ctx_get_stream(rwr_context *ctx);

/* ---------------------------------------------------------------- scene -- */

/* Replaces the storage/uniform/texture bindings TriangleList feeds the mesh pass:
 * vertice_list, face_list, material, texture_diffuse + sampler
 * (src/lib.rs:617-669, triangle_list.rs:212-250, resources.rs:189-203,215-261).
 * `rgba8_srgb` is tex_w*tex_h RGBA8 texels, row 0 = first row of the image
 * file, interpreted as Rgba8UnormSrgb with a ClampToEdge / linear-mag sampler
 * (texture.rs:122,151-159).  n_faces may be 0 (spheres only). */
RWR_API int rwr_scene_upload_mesh(rwr_context *ctx,
                                  const rwr_model_vertex_small *verts, uint32_t n_verts,
                                  const rwr_model_face_small *faces, uint32_t n_faces,
                                  const rwr_material_data *material,
                                  const uint8_t *rgba8_srgb, uint32_t tex_w, uint32_t tex_h);

/* Extension — scenes of several meshes, each with its own material and texture (what
 * resources::load_model_compute returns in Model{meshes, materials}; the reference's TriangleList
 * binds meshes[0]/materials[0] only, triangle_list.rs:212-245).  Faces form one list in part order,
 * then face order, so object ids and the lowest-index tie rule extend across parts.
 * rwr_scene_upload_mesh == clear + add_mesh + commit. */
RWR_API int rwr_scene_clear(rwr_context *ctx);
RWR_API int rwr_scene_add_mesh(rwr_context *ctx,
                               const rwr_model_vertex_small *verts, uint32_t n_verts,
                               const rwr_model_face_small *faces, uint32_t n_faces,
                               const rwr_material_data *material,
                               const uint8_t *rgba8_srgb, uint32_t tex_w, uint32_t tex_h);
RWR_API int rwr_scene_commit(rwr_context *ctx);

/* Extension (RWR_FLAG_NORMAL_MAP): the normal map of scene part `part` (0 for a scene uploaded with
 * rwr_scene_upload_mesh), tex_w*tex_h RGBA8 texels in file order, decoded as LINEAR rgba8unorm (vectors, not
 * colours), same ClampToEdge / bilinear sampler and texture coordinates as the diffuse texture.  Call after the part
 * was added (before or after rwr_scene_commit); NULL removes the map.  Shading: tangent frame of the FACE from its
 * corners and texture coordinates (T = dP/du, B = -dP/dv in sampling space, Gram-Schmidt against the face normal),
 * n' = normalize(T m.x + B m.y + N m.z), m = 2 texel - 1, in place of the flat normal in compute.wgsl:226-229
 * (full definition: oracle/rt_oracle.c normal_mapped). */
RWR_API int rwr_scene_set_normal_map(rwr_context *ctx, uint32_t part, const uint8_t *rgba8_linear, uint32_t tex_w, uint32_t tex_h);
/* Parts (meshes with faces) added to the scene so far. */
RWR_API int rwr_scene_part_count(rwr_context *ctx, uint32_t *n_parts);

/* Replaces Sphere::new's uniform, one per analytic sphere pass, composited in
 * array order before the mesh (src/lib.rs:532-534, 1106-1173).  n <= RWR_MAX_SPHERES. */
RWR_API int rwr_scene_set_spheres(rwr_context *ctx, const rwr_sphere_buffer_data *spheres, uint32_t n);

/* Replaces Triangle::new's uniform (src/models/triangle/triangle.rs:37-45), one per single-triangle pass
 * (triangle/compute.wgsl:153-195).  The reference never dispatches this model, so where its passes sit
 * in a frame is defined here: after the spheres, before the mesh, in array order.  Its shading is the
 * sphere's with the face normal as triangleRayIntersect returns it there — flipped towards the ray, NOT
 * normalised (:120-124,171-187).  n <= RWR_MAX_TRIANGLES; reference frame only (spp 1, no bounce, no
 * RWR_FLAG_USE_BVH). */
RWR_API int rwr_scene_set_triangles(rwr_context *ctx, const rwr_triangle_buffer_data *triangles, uint32_t n);

/* Extension: rigid instances of the uploaded mesh (InstanceRaw layout).
 * n = 0 restores the single un-instanced mesh of the reference. */
RWR_API int rwr_scene_set_instances(rwr_context *ctx, const rwr_instance_raw *instances, uint32_t n);

/* ---------------------------------------------------------------- frame -- */

/* Replaces State::resize's target (re)creation: screen_texture (rgba8unorm),
 * depth_texture_input/output (r32float) — src/lib.rs:470-515, 772-860. */
RWR_API int rwr_resize(rwr_context *ctx, const rwr_screen *screen);

/* Replaces State::render's GPU work, src/lib.rs:1024-1184: clear, sphere passes,
 * depth copies and the mesh pass, as ONE fused launch when params are the
 * reference's (spp 1, no bounce); the wavefront integrator otherwise.
 * Asynchronous.  `params` may be NULL (= {1,0,0,0}). */
RWR_API int rwr_render(rwr_context *ctx, const rwr_camera_inv_uniform *camera, const rwr_render_params *params);

/* Same, restricted to framebuffer rows [row_begin,row_end): the band one rank
 * renders when a frame is split across GPUs.  Pixels are addressed and the RNG
 * is keyed by GLOBAL pixel coordinates, so bands assemble bit-identically. */
RWR_API int rwr_render_rows(rwr_context *ctx, const rwr_camera_inv_uniform *camera,
                            const rwr_render_params *params, uint32_t row_begin, uint32_t row_end);

/* Same, restricted to every strip_stride-th STRIP of RWR_STRIP_ROWS rows, starting with strip first_strip
 * (< strip_stride): rows 8 (first_strip + k strip_stride) + 0..7.  The INTERLEAVED partition of a frame across
 * GPUs — rwr_render_strips(ctx, camera, params, rank, world) — gives every rank the same share of whatever part
 * of the screen the scene covers; contiguous bands (rwr_render_rows) leave most ranks idle when it covers a few
 * rows (measured: the x16 instanced grid of BASELINE configs[4] sits in two of eight bands).  Same global pixel
 * addressing and RNG keys: strips assemble bit-identically. */
#define RWR_STRIP_ROWS 8u
RWR_API int rwr_render_strips(rwr_context *ctx, const rwr_camera_inv_uniform *camera,
                              const rwr_render_params *params, uint32_t first_strip, uint32_t strip_stride);

RWR_API int rwr_synchronize(rwr_context *ctx);

/* Copies finished targets to host memory (synchronises first).  Any pointer may
 * be NULL.  rgba8: W*H*4 bytes (screen_texture); depth: W*H floats
 * (depth_texture_output); the last three need RWR_FLAG_AUX_OUTPUTS on the render. */
RWR_API int rwr_readback(rwr_context *ctx, uint8_t *rgba8, float *depth,
                         float *rgba_f32, int32_t *obj_id, float *hit_t);

/* Device addresses of the targets of the frame rendered last, for zero-copy consumers
 * (presentation, interop).  Does not wait on the host: it orders the context's stream
 * (rwr_ctx_get_stream) after that frame, so work the caller enqueues there next sees the finished
 * targets; a consumer on another stream synchronises with rwr_synchronize.  The addresses stay the same from frame to
 * frame with one frame in flight (the default) and alternate between the target sets otherwise;
 * valid until the next rwr_resize / rwr_ctx_set_frames_in_flight. */
RWR_API int rwr_get_device_targets(rwr_context *ctx, void **d_rgba8, void **d_depth);

/* ------------------------------------------------------- multi-GPU frames -- */
/* The reference is single-device (one wgpu Adapter/Device, src/lib.rs:266-303, one surface, :1226); north_star asks
 * for frames partitioned across the GPUs of a node with a single RCCL gather of the finished tiles.  Model: ONE
 * PROCESS AND ONE CONTEXT PER GPU (rwr_ctx_create(device)), every rank holds the whole scene and renders its row
 * band with rwr_render_rows (pixels and the RNG are keyed by global coordinates: bands assemble bit-identically),
 * then every rank calls rwr_dist_gather_rgba8: its band goes to `root` over xGMI — RCCL point-to-point sends
 * grouped into one operation, enqueued on the stream of the frame just rendered, bands landing in final image
 * order in a buffer the root's context owns.  RCCL is loaded at run time by rwr_dist_get_unique_id / rwr_dist_init;
 * hosts that never call these need no RCCL.
 *   rank 0:  rwr_dist_get_unique_id(id)  -> the launcher hands `id` to every rank (any channel: a file, a socket,
 *            torchrun's store) ->  all ranks: rwr_dist_init(ctx, rank, world, id).                                */
#define RWR_DIST_ID_BYTES 128
RWR_API int rwr_dist_get_unique_id(uint8_t id[RWR_DIST_ID_BYTES]);
RWR_API int rwr_dist_init(rwr_context *ctx, int rank, int world, const uint8_t id[RWR_DIST_ID_BYTES]);
/* The band partition every rank must use: rows [rank*H/world, (rank+1)*H/world). */
RWR_API int rwr_dist_band(uint32_t rank, uint32_t world, uint32_t height, uint32_t *row_begin, uint32_t *row_end);
/* Collective, asynchronous (stream-ordered after the frame rendered last).  Every frame slot
 * (rwr_ctx_set_frames_in_flight) owns its own message / receive / frame buffers, so the gather of one frame runs beside
 * the render of the next; the RCCL exchanges themselves stay in call order. */
RWR_API int rwr_dist_gather_rgba8(rwr_context *ctx, int root);
/* The same for the interleaved partition (every rank rendered rwr_render_strips(ctx, ..., rank, world)): each rank packs
 * its strips into one message (one launch), the root deals the received strips out into the frame (one launch).  One
 * grouped exchange per frame. */
RWR_API int rwr_dist_gather_strips_rgba8(rwr_context *ctx, int root);
/* Root only: device address of the assembled W*H*4 frame gathered last / copy to the host (waits for that gather).  With
 * several frames in flight the address alternates between the slots' buffers. */
RWR_API int rwr_dist_frame(rwr_context *ctx, void **d_rgba8);
RWR_API int rwr_dist_readback(rwr_context *ctx, uint8_t *rgba8);
/* Collective: returns when every rank's frames in flight have finished (an all-reduce of one word). */
RWR_API int rwr_dist_barrier(rwr_context *ctx);
RWR_API int rwr_dist_destroy(rwr_context *ctx);

/* The layout of the interleaved partition's gather, as the library itself uses it (csrc/rwr_strips.h) — for a host that
 * wants to size buffers or exchange the messages by other means.  Rank `rank` of `world` owns `strips` strips of the
 * frame's `n_strips` (strip s belongs to rank s % world); its message is those strips back to back, `rows` rows
 * (a short last strip of the frame ends the message of the rank that owns it: owns_tail); in the root's receive buffer
 * (`recv_rows_total` rows) that message starts at row `recv_row` (messages start on whole-strip boundaries). */
typedef struct rwr_strip_layout {
    uint32_t n_strips, strips, rows, recv_row, recv_rows_total, owns_tail;
} rwr_strip_layout;
RWR_API int rwr_dist_strip_layout(uint32_t rank, uint32_t world, uint32_t height, rwr_strip_layout *out);
/* Pack and deal-out on HOST memory by the same layout (byte moves only; the render path stays on the GPU): for a host
 * that stages the exchange through CPU memory, and for the world-size-2/3 tests that run over gloo without a GPU.
 * message: rank's strips back to back (8 * strips rows of capacity); recv: recv_rows_total rows. */
RWR_API int rwr_dist_host_pack_strips(uint32_t rank, uint32_t world, uint32_t width, uint32_t height,
                                      const uint8_t *frame_rgba8, uint8_t *message);
RWR_API int rwr_dist_host_deal_strips(uint32_t world, uint32_t width, uint32_t height, const uint8_t *recv, uint8_t *frame_rgba8);
/* Self-test of the gather's own stages on ONE GPU for any world size, no communicator: the context plays every rank in
 * turn.  After rendering rank's share (rwr_render_strips(ctx, ..., rank, world), or rwr_render_rows of rwr_dist_band
 * with strips = 0) _deposit runs that rank's side of the gather on the frame just rendered — the same pack launch, message
 * size and receive address as rwr_dist_gather_[strips_]rgba8 — with a device copy in place of the ncclSend/ncclRecv
 * pair; after the last rank _finish runs the root's deal-out.  rwr_dist_readback / rwr_dist_frame then return the frame
 * a root would hold. */
RWR_API int rwr_dist_loopback_deposit(rwr_context *ctx, uint32_t rank, uint32_t world, int strips);
RWR_API int rwr_dist_loopback_finish(rwr_context *ctx, uint32_t world, int strips);

/* hipEvent timing on the stream(s) the kernels are launched on: rwr_timer_begin waits until nothing
 * is in flight and records; rwr_timer_end joins every frame in flight into the end event. */
RWR_API int rwr_timer_begin(rwr_context *ctx);
RWR_API int rwr_timer_end(rwr_context *ctx, float *elapsed_ms); /* synchronises on the end event */
/* The same end in two halves, for a caller that waits for the device itself: rwr_timer_stop only
 * enqueues the end event (no host wait), rwr_timer_elapsed waits for it and reads the interval. */
RWR_API int rwr_timer_stop(rwr_context *ctx);
RWR_API int rwr_timer_elapsed(rwr_context *ctx, float *elapsed_ms);

/* Per-kernel timing for roofline accounting: when every_n > 0, every n-th render call
 * brackets its DOMINANT kernel (k_primary; for the wavefront integrator all sample passes of
 * the frame) with hipEvents on the launch stream.  rwr_kernel_timing_stats synchronises and
 * returns the mean duration in microseconds over the brackets recorded since it was enabled
 * (at most 256 are kept) and their count. */
RWR_API int rwr_ctx_set_kernel_timing(rwr_context *ctx, uint32_t every_n);
RWR_API int rwr_kernel_timing_stats(rwr_context *ctx, double *mean_us, uint32_t *count);

/* Segments (rays) traced by the last render call, for Mray/s accounting:
 * W*rows*spp primary + bounce rays actually emitted. */
RWR_API int rwr_last_render_stats(rwr_context *ctx, uint64_t *primary_rays, uint64_t *bounce_rays);

/* Self-test of the kernels' short exact forms ("Numerics"): the frame kernel replaces the
 * shader's  ((1/d) - (1/kNear)) / ((1/kFar) - (1/kNear))  (compute.wgsl:78-80) and the three divisions
 * of normalize() (:162) by shorter instruction sequences that return the same bits.  This runs both
 * forms on the GPU: every float of the depth form's domain (about 2^31 inputs) and `normalize_count`
 * pseudo-random vectors; out4 = {depth inputs compared, depth mismatches, vectors compared, vector
 * mismatches}.  A non-zero mismatch count is a bug. */
RWR_API int rwr_selftest_exact_math(rwr_context *ctx, uint32_t normalize_count, uint32_t seed, uint64_t out4[4]);

/* Measurement aid for roofline accounting (bench.py): runs a short f32 VALU loop with `waves_per_simd` (1..8)
 * waves on every SIMD and stamps the shader cycle counter against the constant 100 MHz counter.
 * out4 = {shader clock in MHz under v_fma_f32 load, shader cycles a SIMD spends per wave64 v_fma_f32,
 *         shader cycles per wave64 v_pk_fma_f32, shader clock in MHz under v_pk_fma_f32 load}. */
RWR_API int rwr_measure_valu_clock(rwr_context *ctx, uint32_t waves_per_simd, double out4[4]);
/* The shader clock under the caller's OWN workload: _start launches one idle-spinning wave on a private stream for
 * `micros` microseconds (asynchronous; render while it runs), _read waits for it and returns
 * d(shader cycle counter) / d(100 MHz counter) x 100 MHz. */
RWR_API int rwr_clock_probe_start(rwr_context *ctx, uint32_t micros);
RWR_API int rwr_clock_probe_read(rwr_context *ctx, double *shader_mhz);

/* ---------------------------------------------- host-side L2 surface (CPU) -- */

/* CameraInvUniform::update_view_proj, src/lib.rs:105-111 with camera.rs:20-30. */
RWR_API int rwr_camera_build_inv_uniform(const rwr_camera *camera, rwr_camera_inv_uniform *out);

/* CircleCameraController::update_camera, src/circle_camera_control.rs:76-105. */
enum { RWR_KEY_FORWARD = 1, RWR_KEY_BACKWARD = 2, RWR_KEY_LEFT = 4, RWR_KEY_RIGHT = 8,
       RWR_KEY_UP = 16, RWR_KEY_DOWN = 32 };
RWR_API int rwr_circle_controller_update(float speed, uint32_t pressed_keys, rwr_camera *camera);

/* resources::load_model_compute, src/resources.rs:163-264: OBJ + MTL + diffuse
 * texture from `res_dir`.  Only meshes[0]/materials[0] are exposed, as in
 * TriangleList (triangle_list.rs:212-245). */
typedef struct rwr_model rwr_model;
RWR_API int rwr_load_model_compute(const char *res_dir, const char *file_name, rwr_model **out_model);
RWR_API void rwr_model_free(rwr_model *model);
RWR_API int rwr_model_info(const rwr_model *model, uint32_t *n_meshes, uint32_t *n_materials,
                           uint32_t *n_verts, uint32_t *n_faces, uint32_t *tex_w, uint32_t *tex_h);
RWR_API const rwr_model_vertex_small *rwr_model_vertices(const rwr_model *model);
RWR_API const rwr_model_face_small *rwr_model_faces(const rwr_model *model);
RWR_API const rwr_material_data *rwr_model_material(const rwr_model *model);
RWR_API const uint8_t *rwr_model_texture_rgba8(const rwr_model *model);
/* Convenience: rwr_scene_upload_mesh(ctx, <meshes[0] / materials[0] of model>) — the reference's scene. */
RWR_API int rwr_scene_upload_model(rwr_context *ctx, const rwr_model *model);
/* Extension: every mesh of the model with ITS material (mesh.material, resources.rs:257). */
RWR_API int rwr_model_part_count(const rwr_model *model, uint32_t *n_parts);
RWR_API int rwr_model_part(const rwr_model *model, uint32_t part,
                           const rwr_model_vertex_small **verts, uint32_t *n_verts,
                           const rwr_model_face_small **faces, uint32_t *n_faces,
                           rwr_material_data *material, const uint8_t **rgba8, uint32_t *tex_w, uint32_t *tex_h);
RWR_API int rwr_scene_upload_model_all(rwr_context *ctx, const rwr_model *model);
/* Extension: the decoded map_Bump image of a part's material (cube.mtl:13), *rgba8 = NULL when the material names none
 * or the file is not there.  rwr_scene_upload_model / _all hand it to rwr_scene_set_normal_map; only renders with
 * RWR_FLAG_NORMAL_MAP look at it. */
RWR_API int rwr_model_part_normal_map(const rwr_model *model, uint32_t part, const uint8_t **rgba8, uint32_t *tex_w, uint32_t *tex_h);

/* texture::Texture::from_bytes, src/texture.rs:98-106: decode PNG/JPEG bytes to
 * RGBA8.  *out_rgba is malloc'd; free with rwr_free(). */
RWR_API int rwr_decode_image_rgba8(const uint8_t *bytes, size_t n_bytes,
                                   uint8_t **out_rgba, uint32_t *out_w, uint32_t *out_h);
RWR_API void rwr_free(void *p);

/* Presentation step (role of src/screenquad.wgsl + the sRGB swapchain,
 * src/lib.rs:39-64, 310-315, 1186-1224): writes a framebuffer as an RGBA8 PNG.
 * flip_vertical != 0 puts framebuffer row 0 at the BOTTOM of the image, as the
 * reference's blit does; encode_srgb != 0 applies the linear->sRGB transfer the
 * sRGB surface format applies on store. */
RWR_API int rwr_write_png_rgba8(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height,
                                int flip_vertical, int encode_srgb);

/* The grid of Instance{position,rotation}.to_raw() of src/lib.rs:400-421 for a
 * given NUM_INSTANCES_PER_ROW / SPACE_BETWEEN; out must hold per_row*per_row. */
RWR_API int rwr_make_instance_grid(uint32_t per_row, float space_between, rwr_instance_raw *out);

#ifdef __cplusplus
}
#endif
#endif /* RWR_HIP_H */
