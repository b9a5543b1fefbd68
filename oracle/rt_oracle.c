/*
 * rt_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference renderer's per-pixel algorithm
 * (clejacquet/rust-wgpu-raytracing).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product path
 * (rust-wgpu-raytracing_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference holds no tests, golden images or known-answer
 * vectors for this path and it cannot be built here (no
 * Rust toolchain, no Vulkan ICD), so this restatement is pinned only by
 *   (1) the analytic known answers in tests/test_oracle_kat.py and
 *   (2) the provisional values SURVEY.md §8(c) recorded from an independent
 *       float32 NumPy restatement.
 * Third-party arithmetic that is not in /root/reference (cgmath 0.18.0
 * look_at_rh / perspective / Matrix4::invert, naga/driver WGSL builtins) is
 * restated from its published formulas.
 *
 * Reference lines followed (all relative to /root/reference/):
 *   src/models/triangle_list/compute.wgsl:78-240   mesh pass
 *   src/models/sphere/compute.wgsl:52-158          sphere pass
 *   src/lib.rs:31-37, 86-112                       G = OPENGL_TO_WGPU, CameraInvUniform
 *   src/camera.rs:13-30                            view/proj inverse builders
 *   src/lib.rs:1024-1184                           frame sequence (clear, sphere, copy, sphere, copy, mesh)
 *   src/circle_camera_control.rs:76-105            controller update
 *   src/texture.rs:108-166                         Rgba8UnormSrgb + clamp/bilinear sampler
 *
 * ARITHMETIC SPEC (what "the same result" means, f32 everywhere):
 *   WGSL leaves fusion and builtin precision to the implementation.  This
 *   oracle fixes the LITERAL reading so that results are reproducible: every
 *   expression is evaluated exactly as written, left to right, one IEEE-754
 *   binary32 rounding per operation, NO fused multiply-add anywhere (build
 *   with -ffp-contract=off), IEEE divide and sqrt, no fast-math:
 *     dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *     cross(a,b).x  = a.y*b.z - a.z*b.y                  (cyclic for y,z)
 *     M*v (row i)   = ((m0i*v.x + m1i*v.y) + m2i*v.z) + m3i*v.w
 *     o + t*d       = o + (t*d)  per component
 *     normalize(v)  = v / sqrt(dot(v,v))
 *   (An FMA-contracted reading is equally legal WGSL but opens cracks on
 *   exactly symmetric shared edges, e.g. the cube's face diagonals seen from
 *   the origin; the literal reading keeps edges inclusive as the shader
 *   intends, compute.wgsl:118-138.)
 *   pow(x,32) is libm powf.  sRGB decode uses a 256-entry f32 table built from
 *   the double-precision transfer function.
 *
 * EXTENSIONS beyond the reference (BASELINE.json configs 3-5; defined by this
 * project, see DESIGN.md §"Extended integrator"): sub-pixel jitter, spp > 1,
 * one cosine-weighted diffuse bounce, rigid instances.  or_render_path().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OR_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ PODs -- */
/* src/lib.rs:86-93 */
typedef struct {
    float viewmodel_inv[4][4]; /* column-major: [col][row] */
    float proj_inv[4][4];
    float origin[3];
    uint32_t _padding;
} OrCameraInvUniform;
/* src/lib.rs:216-221 */
typedef struct { uint32_t width, height; } OrScreen;
/* src/model.rs:45-63 */
typedef struct { float position[3]; float pad0; float tex_coords[2]; float pad1[2]; } OrVertex;
/* src/model.rs:65-79 */
typedef struct { uint32_t indices[3]; uint32_t pad0; } OrFace;
/* src/models/triangle_list/triangle_list.rs:24-33 */
typedef struct { float ambient[3]; float pad0; float diffuse[3]; float pad1; float specular[3]; float pad2; } OrMaterial;
/* src/models/sphere/sphere.rs:10-15 */
typedef struct { float center[3]; float radius; } OrSphere;
/* src/camera.rs:3-11 */
typedef struct { float eye[3]; float target[3]; float up[3]; float aspect, fovy, znear, zfar; } OrCamera;

_Static_assert(sizeof(OrCameraInvUniform) == 144, "CameraInvUniform is 144 B");
_Static_assert(sizeof(OrVertex) == 32, "ModelVertexSmall is 32 B");
_Static_assert(sizeof(OrFace) == 16, "ModelFaceSmall is 16 B");
_Static_assert(sizeof(OrMaterial) == 48, "MaterialData is 48 B");
_Static_assert(sizeof(OrSphere) == 16, "SphereBufferData is 16 B");

/* ------------------------------------------------------------- vec math -- */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_from(const float *p) { return V3(p[0], p[1], p[2]); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b)
{
    return V3(a.y * b.z - a.z * b.y,
              a.z * b.x - a.x * b.z,
              a.x * b.y - a.y * b.x);
}
static inline v3 normalize3(v3 a)
{
    float len = sqrtf(dot3(a, a));
    return V3(a.x / len, a.y / len, a.z / len);
}
static inline v3 madd3(float t, v3 d, v3 o) { return V3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z); }
static inline v4 mat4_mul_v4(const float m[4][4], v4 v)
{
    v4 r;
    r.x = m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z + m[3][0] * v.w;
    r.y = m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z + m[3][1] * v.w;
    r.z = m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z + m[3][2] * v.w;
    r.w = m[0][3] * v.x + m[1][3] * v.y + m[2][3] * v.z + m[3][3] * v.w;
    return r;
}

/* ------------------------------------------------------- shader pieces -- */
typedef struct { v3 origin, direction; } Ray;

/* compute.wgsl:51-53 (both shaders) */
static const float kNear = 0.01f;
static const float kFar = 100.0f;
static const float kEpsilon = 0.000001f;

/* triangle_list/compute.wgsl:78-80, sphere/compute.wgsl:59-61 */
static inline float to_non_linear_depth(float depth)
{
    return ((1.0f / depth) - (1.0f / kNear)) / ((1.0f / kFar) - (1.0f / kNear));
}

/* triangle_list/compute.wgsl:150-164 (sphere/compute.wgsl:87-101 is identical).
 * (jx, jy) = (0.5, 0.5) in the reference; other values are the jitter extension. */
static inline Ray pixel_to_ray(const OrCameraInvUniform *cam, const OrScreen *screen,
                               uint32_t x, uint32_t y, float jx, float jy)
{
    float x_nds = 2.0f * ((float)x + jx) / (float)screen->width - 1.0f;
    float y_nds = 2.0f * ((float)y + jy) / (float)screen->height - 1.0f;
    v4 proj_vec = {x_nds, y_nds, 1.0f, 1.0f};
    v4 view_vec = mat4_mul_v4(cam->proj_inv, proj_vec);
    view_vec.w = 0.0f;
    v4 world_vec = mat4_mul_v4(cam->viewmodel_inv, view_vec);
    Ray r;
    r.origin = v3_from(cam->origin);
    r.direction = normalize3(V3(world_vec.x, world_vec.y, world_vec.z));
    return r;
}

typedef struct {
    int hit;
    float distance;
    v3 normal;
    v3 barycentric;
} HitRecord;

static const HitRecord kNoHit = {0, 0.0f, {0, 0, 0}, {0, 0, 0}};

/* triangle_list/compute.wgsl:82-148 */
static inline HitRecord triangle_ray_intersect(v3 p0, v3 p1, v3 p2, Ray ray)
{
    v3 v0v1 = sub3(p1, p0);
    v3 v0v2 = sub3(p2, p0);
    v3 N = cross3(v0v1, v0v2);
    float denom = dot3(N, N);

    float NdotRayDirection = dot3(N, ray.direction);
    if (fabsf(NdotRayDirection) < kEpsilon) return kNoHit;

    float d = -dot3(N, p0);
    float t = -(dot3(N, ray.origin) + d) / NdotRayDirection;
    if (t < 0.0f) return kNoHit;

    v3 P = madd3(t, ray.direction, ray.origin);

    v3 edge0 = sub3(p1, p0);
    v3 vp0 = sub3(P, p0);
    v3 C = cross3(edge0, vp0);
    if (dot3(N, C) < 0.0f) return kNoHit;

    v3 edge1 = sub3(p2, p1);
    v3 vp1 = sub3(P, p1);
    C = cross3(edge1, vp1);
    float u = dot3(N, C);
    if (u < 0.0f) return kNoHit;

    v3 edge2 = sub3(p0, p2);
    v3 vp2 = sub3(P, p2);
    C = cross3(edge2, vp2);
    float v = dot3(N, C);
    if (v < 0.0f) return kNoHit;

    if (NdotRayDirection > 0.0f) N = neg3(N);

    u = u / denom;
    v = v / denom;

    HitRecord h;
    h.hit = 1;
    h.distance = t;
    h.normal = normalize3(N);
    h.barycentric = V3(u, v, 1.0f - u - v);
    return h;
}

/* sphere/compute.wgsl:52-85 */
static inline HitRecord sphere_ray_intersect(v3 center, float radius, Ray ray)
{
    v3 oc = sub3(ray.origin, center);
    float a = dot3(ray.direction, ray.direction);
    float b = 2.0f * dot3(oc, ray.direction);
    float c = dot3(oc, oc) - (radius * radius);
    float discriminant = b * b - 4.0f * a * c;
    if (discriminant < 0.0f) return kNoHit;
    float sq = sqrtf(discriminant);
    float t1 = (-b - sq) / (2.0f * a);
    float t2 = (-b + sq) / (2.0f * a);
    float t;
    if (t1 >= 0.0f) t = t1;
    else if (t2 >= 0.0f) t = t2;
    else return kNoHit;
    HitRecord h;
    h.hit = 1;
    h.distance = t;
    v3 P = madd3(t, ray.direction, ray.origin); /* ray.origin + ray.direction * t */
    h.normal = normalize3(sub3(P, center));
    h.barycentric = V3(0, 0, 0);
    return h;
}

/* texture.rs:122 (Rgba8UnormSrgb) — decode table, built once per call site. */
static void build_srgb_lut(float lut[256])
{
    for (int i = 0; i < 256; i++) {
        double c = (double)i / 255.0;
        double l = (c <= 0.04045) ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4);
        lut[i] = (float)l;
    }
}

typedef struct {
    const uint8_t *rgba; /* W*H*4, row 0 = first image row (top of the PNG) */
    uint32_t w, h;
    float lut[256];
} Tex;

/* textureSampleGrad(..., grad 0,0): LOD 0, magnification => bilinear,
 * ClampToEdge, texels sRGB-decoded before filtering
 * (triangle_list/compute.wgsl:225, texture.rs:151-159). */
static inline v3 tex_sample_bilinear(const Tex *t, float u, float v)
{
    float fx = u * (float)t->w - 0.5f;
    float fy = v * (float)t->h - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float ax = fx - x0f, ay = fy - y0f;
    /* clamp in float first so that huge/NaN coordinates cannot overflow int */
    float wmax = (float)(t->w - 1), hmax = (float)(t->h - 1);
    float x0c = fminf(fmaxf(x0f, 0.0f), wmax), x1c = fminf(fmaxf(x0f + 1.0f, 0.0f), wmax);
    float y0c = fminf(fmaxf(y0f, 0.0f), hmax), y1c = fminf(fmaxf(y0f + 1.0f, 0.0f), hmax);
    uint32_t x0 = (uint32_t)x0c, x1 = (uint32_t)x1c, y0 = (uint32_t)y0c, y1 = (uint32_t)y1c;
    const uint8_t *t00 = t->rgba + 4 * ((size_t)y0 * t->w + x0);
    const uint8_t *t10 = t->rgba + 4 * ((size_t)y0 * t->w + x1);
    const uint8_t *t01 = t->rgba + 4 * ((size_t)y1 * t->w + x0);
    const uint8_t *t11 = t->rgba + 4 * ((size_t)y1 * t->w + x1);
    float w00 = (1.0f - ax) * (1.0f - ay), w10 = ax * (1.0f - ay);
    float w01 = (1.0f - ax) * ay, w11 = ax * ay;
    float c[3];
    for (int k = 0; k < 3; k++) {
        c[k] = t->lut[t00[k]] * w00 + t->lut[t10[k]] * w10 + t->lut[t01[k]] * w01 + t->lut[t11[k]] * w11;
    }
    return V3(c[0], c[1], c[2]);
}

/* rgba8unorm store (compute.wgsl:237, textureStore to an rgba8unorm texture): the float -> UNORM
 * conversion of the WebGPU / Vulkan specs, clamp to [0,1] then round(c * 255).  Both specs leave the
 * handling of exact ties to the implementation (Vulkan: "round to nearest even is preferred"); this
 * oracle fixes ROUND TO NEAREST, TIES TO EVEN (rintf in the default rounding mode), which is also
 * what the store conversion of the target hardware does.  NaN stores 0. */
static inline uint8_t unorm8(float c)
{
    float cc = fminf(fmaxf(c, 0.0f), 1.0f);
    return (uint8_t)rintf(cc * 255.0f);
}

/* ----------------------------------------------------------- the scene -- */
typedef struct {
    const OrVertex *verts; uint32_t n_verts;
    const OrFace *faces;   uint32_t n_faces;
    const OrMaterial *material;
    Tex tex;
    /* EXTENSION (multi-material scenes; the reference consumes materials[0] only,
     * triangle_list.rs:212): when face_material != NULL, face i is shaded with
     * materials[face_material[i % n_base_faces]] / texs[...] instead of material / tex. */
    const uint32_t *face_material; uint32_t n_base_faces;
    const OrMaterial *materials;
    const Tex *texs;
    /* EXTENSION (normal-mapped shading, see shade_mesh): one optional map per material, consulted only when the
     * render asked for it (OR_FLAG_NORMAL_MAP). */
    const Tex *nmaps; int use_nmap;
} Mesh;

#define OR_FLAG_NORMAL_MAP (1u << 4)

/* Bilinear ClampToEdge fetch like tex_sample_bilinear, but of a LINEAR rgba8unorm texture (texel = byte / 255):
 * a normal map holds vectors, not colours. */
static inline v3 tex_sample_bilinear_linear(const Tex *t, float u, float v)
{
    float fx = u * (float)t->w - 0.5f, fy = v * (float)t->h - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float ax = fx - x0f, ay = fy - y0f;
    float wmax = (float)(t->w - 1u), hmax = (float)(t->h - 1u);
    uint32_t x0 = (uint32_t)fminf(fmaxf(x0f, 0.0f), wmax), x1 = (uint32_t)fminf(fmaxf(x0f + 1.0f, 0.0f), wmax);
    uint32_t y0 = (uint32_t)fminf(fmaxf(y0f, 0.0f), hmax), y1 = (uint32_t)fminf(fmaxf(y0f + 1.0f, 0.0f), hmax);
    const uint8_t *t00 = t->rgba + 4u * ((size_t)y0 * t->w + x0), *t10 = t->rgba + 4u * ((size_t)y0 * t->w + x1);
    const uint8_t *t01 = t->rgba + 4u * ((size_t)y1 * t->w + x0), *t11 = t->rgba + 4u * ((size_t)y1 * t->w + x1);
    float w00 = (1.0f - ax) * (1.0f - ay), w10 = ax * (1.0f - ay), w01 = (1.0f - ax) * ay, w11 = ax * ay;
    float c[3];
    for (int k = 0; k < 3; k++)
        c[k] = (float)t00[k] / 255.0f * w00 + (float)t10[k] / 255.0f * w10 + (float)t01[k] / 255.0f * w01 + (float)t11[k] / 255.0f * w11;
    return V3(c[0], c[1], c[2]);
}

/* EXTENSION — normal-mapped shading (north_star: "diffuse/normal-map shading"; res/cube.mtl:13 ships `map_Bump
 * cube-normal.png`, but the reference loads the diffuse texture only, resources.rs:187-213, and compute.wgsl:226-229
 * shades with the flat face normal).  Definition (this project's; the default frame never takes this path):
 *   tangent frame of the FACE from its world-space corners and its texture coordinates in sampling space
 *   (u, 1 - v) (compute.wgsl:224 flips v):  T = dP/du,  B = -dP/dv' (the map's green axis points up the image),
 *   n^ = normalize(cross(p1 - p0, p2 - p0)) as wound;  t^ = normalize(T - n^ (n^.T)),  b^ = +-cross(n^, t^) with the
 *   sign that makes b^.B >= 0;  a face with degenerate texture coordinates has t^ = b^ = 0.  The frame is computed
 *   in double and rounded to f32 once (per-face records on the GPU);
 *   c = the map, decoded as LINEAR rgba8unorm, bilinear + ClampToEdge at the hit's (u, 1 - v);  m = 2 c - 1;
 *   n' = normalize(t^ m.x + b^ m.y + s n^ m.z),  s = -1 when the flat normal was flipped towards the ray
 *   (compute.wgsl:140-142), n' = s n^ when that sum vanishes;
 *   n' replaces the flat normal in the Lambert and Blinn-Phong terms of compute.wgsl:226-229.  Bounce rays still leave
 *   along the geometric normal. */
static inline v3 normal_mapped(const Mesh *m, const OrFace *f, const Tex *nm, float tu, float tv, v3 flat_normal)
{
    double p[3][3], uv[3][2];
    for (int k = 0; k < 3; k++) {
        const OrVertex *vx = &m->verts[f->indices[k]];
        for (int c = 0; c < 3; c++) p[k][c] = (double)vx->position[c];
        uv[k][0] = (double)vx->tex_coords[0];
        uv[k][1] = 1.0 - (double)vx->tex_coords[1];
    }
    double d1[3], d2[3], ng[3];
    for (int c = 0; c < 3; c++) { d1[c] = p[1][c] - p[0][c]; d2[c] = p[2][c] - p[0][c]; }
    ng[0] = d1[1] * d2[2] - d1[2] * d2[1]; ng[1] = d1[2] * d2[0] - d1[0] * d2[2]; ng[2] = d1[0] * d2[1] - d1[1] * d2[0];
    const double nl = sqrt(ng[0] * ng[0] + ng[1] * ng[1] + ng[2] * ng[2]);
    for (int c = 0; c < 3; c++) ng[c] /= nl;
    const double du1 = uv[1][0] - uv[0][0], dv1 = uv[1][1] - uv[0][1], du2 = uv[2][0] - uv[0][0], dv2 = uv[2][1] - uv[0][1];
    const double det = du1 * dv2 - dv1 * du2;
    double th[3] = {0, 0, 0}, bh[3] = {0, 0, 0};
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
    const v3 t_hat = V3((float)th[0], (float)th[1], (float)th[2]), b_hat = V3((float)bh[0], (float)bh[1], (float)bh[2]);
    const v3 n_hat = V3((float)ng[0], (float)ng[1], (float)ng[2]);
    const float s = dot3(flat_normal, n_hat) < 0.0f ? -1.0f : 1.0f;
    const v3 c = tex_sample_bilinear_linear(nm, tu, tv);
    const v3 mm = V3(2.0f * c.x - 1.0f, 2.0f * c.y - 1.0f, 2.0f * c.z - 1.0f);
    v3 n = V3(t_hat.x * mm.x + b_hat.x * mm.y + s * n_hat.x * mm.z, t_hat.y * mm.x + b_hat.y * mm.y + s * n_hat.y * mm.z,
              t_hat.z * mm.x + b_hat.z * mm.y + s * n_hat.z * mm.z);
    const float l2 = dot3(n, n);
    if (!(l2 > 1e-20f)) return scale3(n_hat, s);
    return scale3(n, 1.0f / sqrtf(l2));
}

/* Local shading of a mesh hit — triangle_list/compute.wgsl:217-234.
 * Returns final_color.rgb (alpha is 2.0); *albedo receives the filtered texel
 * (needed only by the bounce extension). */
static inline v3 shade_mesh(const Mesh *m, uint32_t i_min, const HitRecord *h, Ray ray, v3 *albedo)
{
    const v3 kLightDir = {1.0f, -1.0f, -5.0f}; /* compute.wgsl:55 */
    const OrFace *f = &m->faces[i_min];
    const float *tc0 = m->verts[f->indices[0]].tex_coords;
    const float *tc1 = m->verts[f->indices[1]].tex_coords;
    const float *tc2 = m->verts[f->indices[2]].tex_coords;
    float tu = h->barycentric.x * tc0[0] + h->barycentric.y * tc1[0] + h->barycentric.z * tc2[0];
    float tv = h->barycentric.x * tc0[1] + h->barycentric.y * tc1[1] + h->barycentric.z * tc2[1];
    tv = 1.0f - tv;
    const OrMaterial *mat = m->material;
    const Tex *tx = &m->tex;
    uint32_t mid = 0;
    if (m->face_material) {
        mid = m->face_material[i_min % m->n_base_faces];
        mat = &m->materials[mid];
        tx = &m->texs[mid];
    }
    v3 tex = tex_sample_bilinear(tx, tu, tv);
    if (albedo) *albedo = tex;
    v3 normal = h->normal;
    if (m->use_nmap && m->nmaps && m->nmaps[mid].rgba) normal = normal_mapped(m, f, &m->nmaps[mid], tu, tv, h->normal);  /* extension */

    v3 nl = neg3(normalize3(kLightDir));
    float ndl = fmaxf(0.0f, dot3(normal, nl));
    v3 diffuse = scale3(tex, ndl);
    v3 half_dir = normalize3(sub3(nl, ray.direction));
    float sp = powf(fmaxf(0.0f, dot3(half_dir, normal)), 32.0f);
    v3 specular = V3(mat->specular[0] * sp, mat->specular[1] * sp, mat->specular[2] * sp);
    v3 out;
    out.x = (mat->ambient[0] + diffuse.x) + specular.x;
    out.y = (mat->ambient[1] + diffuse.y) + specular.y;
    out.z = (mat->ambient[2] + diffuse.z) + specular.z;
    return out;
}

/* Local shading of a sphere hit — sphere/compute.wgsl:137-152. */
static inline v3 shade_sphere(const HitRecord *h, Ray ray, v3 *albedo)
{
    const v3 kLightDir = {1.0f, -5.0f, 1.0f}; /* sphere/compute.wgsl:41 */
    const float ambiant_comp = 0.1f, diffuse_comp = 1.0f, specular_comp = 0.5f;
    v3 nl = neg3(normalize3(kLightDir));
    float diffuse = diffuse_comp * fmaxf(0.0f, dot3(h->normal, nl));
    v3 half_dir = normalize3(sub3(nl, ray.direction));
    float specular = specular_comp * powf(fmaxf(0.0f, dot3(half_dir, h->normal)), 32.0f);
    const v3 mat_color = {1.0f, 0.0f, 0.0f};
    if (albedo) *albedo = mat_color;
    float k = ambiant_comp + diffuse;
    return V3(k * mat_color.x + specular, k * mat_color.y + specular, k * mat_color.z + specular);
}

/* Brute-force nearest hit, lowest index wins ties — compute.wgsl:186-202. */
static inline HitRecord mesh_nearest(const Mesh *m, Ray ray, int *i_min_out)
{
    int i_min = 0;
    HitRecord min_hit = kNoHit;
    for (int i = 0; i < (int)m->n_faces; i++) {
        const OrFace *f = &m->faces[i];
        v3 p0 = v3_from(m->verts[f->indices[0]].position);
        v3 p1 = v3_from(m->verts[f->indices[1]].position);
        v3 p2 = v3_from(m->verts[f->indices[2]].position);
        HitRecord hr = triangle_ray_intersect(p0, p1, p2, ray);
        if ((!min_hit.hit && hr.hit) || (hr.hit && hr.distance < min_hit.distance)) {
            min_hit = hr;
            i_min = i;
        }
    }
    *i_min_out = i_min;
    return min_hit;
}

/* Output planes.  Any pointer may be NULL. */
typedef struct {
    uint8_t *color_u8;   /* W*H*4 rgba8unorm */
    float *color_f32;    /* W*H*4 the vec4 handed to textureStore (pre-clamp) */
    int32_t *obj_id;     /* W*H: face index >= 0 (instance*n_faces + face), -1 untouched, -2-k sphere k */
    float *hit_t;        /* W*H: distance of the winning hit */
} OrAux;

static inline void store_pixel(const OrAux *o, size_t idx, v3 rgb, float alpha, int32_t id, float t)
{
    if (o->color_u8) {
        o->color_u8[4 * idx + 0] = unorm8(rgb.x);
        o->color_u8[4 * idx + 1] = unorm8(rgb.y);
        o->color_u8[4 * idx + 2] = unorm8(rgb.z);
        o->color_u8[4 * idx + 3] = unorm8(alpha);
    }
    if (o->color_f32) {
        o->color_f32[4 * idx + 0] = rgb.x;
        o->color_f32[4 * idx + 1] = rgb.y;
        o->color_f32[4 * idx + 2] = rgb.z;
        o->color_f32[4 * idx + 3] = alpha;
    }
    if (o->obj_id) o->obj_id[idx] = id;
    if (o->hit_t) o->hit_t[idx] = t;
}

/* ------------------------------------------------ literal compute passes -- */
/* sphere/compute.wgsl:114-158, one invocation per pixel. */
OR_API void or_sphere_pass(const OrCameraInvUniform *cam, const OrScreen *screen, const OrSphere *sphere,
                           int32_t sphere_index, const float *depth_input, float *depth_output,
                           uint8_t *color_u8, float *color_f32, int32_t *obj_id, float *hit_t)
{
    OrAux aux = {color_u8, color_f32, obj_id, hit_t};
    const int W = (int)screen->width, H = (int)screen->height;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            Ray ray = pixel_to_ray(cam, screen, (uint32_t)x, (uint32_t)y, 0.5f, 0.5f);
            HitRecord h = sphere_ray_intersect(v3_from(sphere->center), sphere->radius, ray);
            if (!h.hit) continue;
            size_t idx = (size_t)y * W + x;
            float current_depth = 1.0f - depth_input[idx];
            float depth = to_non_linear_depth(h.distance);
            if (depth >= current_depth) continue;
            v3 rgb = shade_sphere(&h, ray, NULL);
            depth_output[idx] = 1.0f - depth;
            store_pixel(&aux, idx, rgb, 2.0f, -2 - sphere_index, h.distance);
        }
    }
}

/* triangle_list/compute.wgsl:177-240, one invocation per pixel. */
OR_API void or_mesh_pass(const OrCameraInvUniform *cam, const OrScreen *screen,
                         const OrVertex *verts, uint32_t n_verts, const OrFace *faces, uint32_t n_faces,
                         const OrMaterial *material, const uint8_t *tex_rgba8, uint32_t tex_w, uint32_t tex_h,
                         const float *depth_input, float *depth_output,
                         uint8_t *color_u8, float *color_f32, int32_t *obj_id, float *hit_t)
{
    Mesh m;
    m.verts = verts; m.n_verts = n_verts; m.faces = faces; m.n_faces = n_faces; m.material = material;
    m.tex.rgba = tex_rgba8; m.tex.w = tex_w; m.tex.h = tex_h;
    m.face_material = NULL; m.n_base_faces = n_faces; m.materials = NULL; m.texs = NULL; m.nmaps = NULL; m.use_nmap = 0;
    build_srgb_lut(m.tex.lut);
    OrAux aux = {color_u8, color_f32, obj_id, hit_t};
    const int W = (int)screen->width, H = (int)screen->height;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            Ray ray = pixel_to_ray(cam, screen, (uint32_t)x, (uint32_t)y, 0.5f, 0.5f);
            int i_min;
            HitRecord h = mesh_nearest(&m, ray, &i_min);
            if (!h.hit) continue;
            size_t idx = (size_t)y * W + x;
            float current_depth = 1.0f - depth_input[idx];
            float depth = to_non_linear_depth(h.distance);
            if (depth >= current_depth) continue;
            v3 rgb = shade_mesh(&m, (uint32_t)i_min, &h, ray, NULL);
            depth_output[idx] = 1.0f - depth;
            store_pixel(&aux, idx, rgb, 2.0f, i_min, h.distance);
        }
    }
}

/* State::render, src/lib.rs:1024-1184: clear screen/depth_in/depth_out to 0,
 * then for each sphere {pass; copy depth_out -> depth_in}, then the mesh pass.
 * Outputs: color_u8 (W*H*4), depth_out (W*H) and the optional aux planes.
 * Returns 0, or -1 on allocation failure. */
OR_API int or_render_frame(const OrCameraInvUniform *cam, const OrScreen *screen,
                           const OrSphere *spheres, uint32_t n_spheres,
                           const OrVertex *verts, uint32_t n_verts, const OrFace *faces, uint32_t n_faces,
                           const OrMaterial *material, const uint8_t *tex_rgba8, uint32_t tex_w, uint32_t tex_h,
                           uint8_t *color_u8, float *depth_out, float *color_f32, int32_t *obj_id, float *hit_t)
{
    size_t n = (size_t)screen->width * screen->height;
    float *depth_in = (float *)calloc(n, sizeof(float));
    if (!depth_in) return -1;
    memset(depth_out, 0, n * sizeof(float));
    if (color_u8) memset(color_u8, 0, n * 4);
    if (color_f32) memset(color_f32, 0, n * 4 * sizeof(float));
    if (obj_id) for (size_t i = 0; i < n; i++) obj_id[i] = -1;
    if (hit_t) memset(hit_t, 0, n * sizeof(float));
    for (uint32_t s = 0; s < n_spheres; s++) {
        or_sphere_pass(cam, screen, &spheres[s], (int32_t)s, depth_in, depth_out, color_u8, color_f32, obj_id, hit_t);
        memcpy(depth_in, depth_out, n * sizeof(float));
    }
    if (n_faces > 0)
        or_mesh_pass(cam, screen, verts, n_verts, faces, n_faces, material, tex_rgba8, tex_w, tex_h,
                     depth_in, depth_out, color_u8, color_f32, obj_id, hit_t);
    free(depth_in);
    return 0;
}


/* ---------------------------------------------- dormant parts of the reference --
 * The reference carries, but never dispatches, a single-triangle model
 * (src/models/triangle/{triangle.rs, compute.wgsl}) and an orthographic ray generator that
 * every shader defines and none calls (pixelToRay_ortho: triangle_list/compute.wgsl:166-174,
 * sphere/compute.wgsl:103-111, triangle/compute.wgsl:143-151).  Restated here for SURVEY
 * §8(f) rank 4.  Where a triangle pass sits in the frame is this project's choice (after the
 * sphere passes, before the mesh pass): each pass only composites through the depth test. */
typedef struct { float p0[3], pad0, p1[3], pad1, p2[3], pad2; } OrTriangle; /* triangle.rs:10-19 */

/* pixelToRay_ortho, triangle_list/compute.wgsl:166-174 */
static inline Ray pixel_to_ray_ortho(const OrCameraInvUniform *cam, const OrScreen *screen, uint32_t x, uint32_t y)
{
    float x_nds = 2.0f * ((float)x + 0.5f) / (float)screen->width - 1.0f;
    float y_nds = 2.0f * ((float)y + 0.5f) / (float)screen->height - 1.0f;
    Ray r;
    r.origin = add3(v3_from(cam->origin), V3(x_nds * 5.0f, y_nds * 5.0f, 0.0f));
    r.direction = V3(0.0f, 0.0f, -1.0f);
    return r;
}
static inline Ray frame_ray(const OrCameraInvUniform *cam, const OrScreen *screen, uint32_t x, uint32_t y, int ortho)
{
    return ortho ? pixel_to_ray_ortho(cam, screen, x, y) : pixel_to_ray(cam, screen, x, y, 0.5f, 0.5f);
}

/* triangleRayIntersect of the single-triangle model, triangle/compute.wgsl:65-125: the same plane +
 * three inclusive edge tests, but the HitRecord carries N as it is — flipped towards the ray, NOT
 * normalised (:120-124) — and no barycentrics. */
static inline HitRecord single_triangle_ray_intersect(v3 p0, v3 p1, v3 p2, Ray ray)
{
    v3 v0v1 = sub3(p1, p0);
    v3 v0v2 = sub3(p2, p0);
    v3 N = cross3(v0v1, v0v2);
    float NdotRayDirection = dot3(N, ray.direction);
    if (fabsf(NdotRayDirection) < kEpsilon) return kNoHit;
    float d = -dot3(N, p0);
    float t = -(dot3(N, ray.origin) + d) / NdotRayDirection;
    if (t < 0.0f) return kNoHit;
    v3 P = madd3(t, ray.direction, ray.origin);
    v3 C = cross3(sub3(p1, p0), sub3(P, p0));
    if (dot3(N, C) < 0.0f) return kNoHit;
    C = cross3(sub3(p2, p1), sub3(P, p1));
    if (dot3(N, C) < 0.0f) return kNoHit;
    C = cross3(sub3(p0, p2), sub3(P, p2));
    if (dot3(N, C) < 0.0f) return kNoHit;
    if (NdotRayDirection > 0.0f) N = neg3(N);
    HitRecord h;
    h.hit = 1;
    h.distance = t;
    h.normal = N;
    h.barycentric = V3(0, 0, 0);
    return h;
}

/* triangle/compute.wgsl:153-195, one invocation per pixel; its shading (:171-187) is the sphere's with
 * the un-normalised N.  Object id of triangle k: -10 - k. */
static void triangle_pass(const OrCameraInvUniform *cam, const OrScreen *screen, const OrTriangle *tri, int32_t index,
                          int ortho, const float *depth_input, float *depth_output, const OrAux *aux)
{
    const int W = (int)screen->width, H = (int)screen->height;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            Ray ray = frame_ray(cam, screen, (uint32_t)x, (uint32_t)y, ortho);
            HitRecord h = single_triangle_ray_intersect(v3_from(tri->p0), v3_from(tri->p1), v3_from(tri->p2), ray);
            if (!h.hit) continue;
            size_t idx = (size_t)y * W + x;
            float current_depth = 1.0f - depth_input[idx];
            float depth = to_non_linear_depth(h.distance);
            if (depth >= current_depth) continue;
            v3 rgb = shade_sphere(&h, ray, NULL);
            depth_output[idx] = 1.0f - depth;
            store_pixel(aux, idx, rgb, 2.0f, -10 - index, h.distance);  /* alpha: 1.0 + 1.0, :184-187 */
        }
    }
}

/* or_render_frame with the dormant parts switched on: `triangles` (may be NULL) are single-triangle
 * passes run after the spheres; `ortho` != 0 makes every pass use pixelToRay_ortho. */
OR_API int or_render_frame_ex(const OrCameraInvUniform *cam, const OrScreen *screen,
                              const OrSphere *spheres, uint32_t n_spheres, const OrTriangle *triangles, uint32_t n_triangles,
                              const OrVertex *verts, uint32_t n_verts, const OrFace *faces, uint32_t n_faces,
                              const OrMaterial *material, const uint8_t *tex_rgba8, uint32_t tex_w, uint32_t tex_h,
                              uint32_t ortho,
                              uint8_t *color_u8, float *depth_out, float *color_f32, int32_t *obj_id, float *hit_t)
{
    size_t n = (size_t)screen->width * screen->height;
    float *depth_in = (float *)calloc(n, sizeof(float));
    if (!depth_in) return -1;
    memset(depth_out, 0, n * sizeof(float));
    if (color_u8) memset(color_u8, 0, n * 4);
    if (color_f32) memset(color_f32, 0, n * 4 * sizeof(float));
    if (obj_id) for (size_t i = 0; i < n; i++) obj_id[i] = -1;
    if (hit_t) memset(hit_t, 0, n * sizeof(float));
    OrAux aux = {color_u8, color_f32, obj_id, hit_t};
    const int W = (int)screen->width, H = (int)screen->height;
    for (uint32_t s = 0; s < n_spheres; s++) {
#pragma omp parallel for schedule(dynamic, 4)
        for (int y = 0; y < H; y++) {
            for (int x = 0; x < W; x++) {   /* or_sphere_pass with the selectable ray */
                Ray ray = frame_ray(cam, screen, (uint32_t)x, (uint32_t)y, (int)ortho);
                HitRecord h = sphere_ray_intersect(v3_from(spheres[s].center), spheres[s].radius, ray);
                if (!h.hit) continue;
                size_t idx = (size_t)y * W + x;
                float current_depth = 1.0f - depth_in[idx];
                float depth = to_non_linear_depth(h.distance);
                if (depth >= current_depth) continue;
                v3 rgb = shade_sphere(&h, ray, NULL);
                depth_out[idx] = 1.0f - depth;
                store_pixel(&aux, idx, rgb, 2.0f, -2 - (int32_t)s, h.distance);
            }
        }
        memcpy(depth_in, depth_out, n * sizeof(float));
    }
    for (uint32_t k = 0; k < n_triangles; k++) {
        triangle_pass(cam, screen, &triangles[k], (int32_t)k, (int)ortho, depth_in, depth_out, &aux);
        memcpy(depth_in, depth_out, n * sizeof(float));
    }
    if (n_faces > 0) {
        Mesh m;
        m.verts = verts; m.n_verts = n_verts; m.faces = faces; m.n_faces = n_faces; m.material = material;
        m.tex.rgba = tex_rgba8; m.tex.w = tex_w; m.tex.h = tex_h;
        m.face_material = NULL; m.n_base_faces = n_faces; m.materials = NULL; m.texs = NULL; m.nmaps = NULL; m.use_nmap = 0; m.nmaps = NULL; m.use_nmap = 0;
        build_srgb_lut(m.tex.lut);
#pragma omp parallel for schedule(dynamic, 4)
        for (int y = 0; y < H; y++) {
            for (int x = 0; x < W; x++) {   /* or_mesh_pass with the selectable ray */
                Ray ray = frame_ray(cam, screen, (uint32_t)x, (uint32_t)y, (int)ortho);
                int i_min;
                HitRecord h = mesh_nearest(&m, ray, &i_min);
                if (!h.hit) continue;
                size_t idx = (size_t)y * W + x;
                float current_depth = 1.0f - depth_in[idx];
                float depth = to_non_linear_depth(h.distance);
                if (depth >= current_depth) continue;
                v3 rgb = shade_mesh(&m, (uint32_t)i_min, &h, ray, NULL);
                depth_out[idx] = 1.0f - depth;
                store_pixel(&aux, idx, rgb, 2.0f, i_min, h.distance);
            }
        }
    }
    free(depth_in);
    return 0;
}

/* ================================================================ EXTENSION ==
 * Multi-sample / one-bounce / instanced integrator (BASELINE.json configs 3-5).
 * The reference has none of this: the definition below is this
 * project's own (DESIGN.md "Extended integrator"); spp = 1, bounces = 0, no
 * instances reproduces or_render_frame bit for bit.
 *
 *  - instances: the mesh is replicated once per rigid InstanceRaw matrix,
 *    world = model * vec4(p, 1) (WGSL mat*vec order), face index =
 *    instance * n_faces + face; the reference loop then runs over that flat list.
 *  - sample s of pixel (x, y): sub-pixel position (jx, jy) = (0.5, 0.5) when
 *    spp == 1, else two uniform numbers from the counter-based hash below keyed
 *    by (GLOBAL pixel index, s, dimension, seed) — never by tile, band or rank.
 *  - primary visibility: the reference's passes in order (spheres, then mesh)
 *    with its depth test.
 *  - radiance of the sample: L = E(h0) + albedo(h0) * E(h1), where E is the
 *    reference's local shading of a surface point (ambient + Lambert + Blinn-
 *    Phong, compute.wgsl:217-234 / sphere 137-152), h1 the nearest hit of ONE
 *    cosine-distributed bounce ray (no shadow rays; a miss adds nothing).
 *  - bounce ray: origin P + 1e-4*n, direction from a rejection-sampled unit disk
 *    point lifted to the hemisphere (Malley) in the orthonormal basis of Duff et
 *    al. 2017 — only + - * / sqrt, so CPU and GPU agree bit for bit.
 *  - bounce visibility: nearest t over spheres (in order) then faces (in order),
 *    strict '<' (earlier candidate wins ties).
 *  - pixel = (sum over s of L_s, in order, f32) / spp; alpha likewise from 2.0 per
 *    primary hit.  depth / obj_id / hit_t planes report sample 0.
 */
typedef struct { uint32_t spp, max_bounces, seed, flags; } OrRenderParams;
typedef struct { float model[4][4]; } OrInstance;

static inline uint32_t rng_mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static inline uint32_t rng_hash(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed)
{
    uint32_t h = seed ^ 0x9E3779B9u;
    h = rng_mix(h ^ pixel);
    h = rng_mix(h ^ (sample * 0x85EBCA6Bu));
    h = rng_mix(h ^ (dim * 0xC2B2AE35u));
    return h;
}
/* uniform in [0,1) with 24 bits, exact in f32 */
static inline float rng_uniform(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed)
{
    return (float)(rng_hash(pixel, sample, dim, seed) >> 8) * (1.0f / 16777216.0f);
}
OR_API uint32_t or_rng_hash(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed) { return rng_hash(pixel, sample, dim, seed); }

/* Cosine-distributed direction about unit normal n. */
static inline v3 bounce_direction(v3 n, uint32_t pixel, uint32_t sample, uint32_t seed)
{
    float a = 0.0f, b = 0.0f;
    for (uint32_t k = 0; k < 8; k++) {  /* rejection-sample the unit disk; dims 2,3 / 4,5 / ... */
        float ua = 2.0f * rng_uniform(pixel, sample, 2u + 2u * k, seed) - 1.0f;
        float ub = 2.0f * rng_uniform(pixel, sample, 3u + 2u * k, seed) - 1.0f;
        if (ua * ua + ub * ub <= 1.0f) { a = ua; b = ub; break; }
    }
    float dz = sqrtf(fmaxf(0.0f, 1.0f - a * a - b * b));
    /* orthonormal basis (Duff, Burgess, Christensen, Hery, Kensler, Liani, Villemin 2017) */
    float sign = copysignf(1.0f, n.z);
    float aa = -1.0f / (sign + n.z);
    float bb = n.x * n.y * aa;
    v3 b1 = V3(1.0f + sign * n.x * n.x * aa, sign * bb, -sign * n.x);
    v3 b2 = V3(bb, sign + n.y * n.y * aa, -n.y);
    v3 d = V3(a * b1.x + b * b2.x + dz * n.x, a * b1.y + b * b2.y + dz * n.y, a * b1.z + b * b2.z + dz * n.z);
    return normalize3(d);
}
OR_API void or_bounce_direction(const float n[3], uint32_t pixel, uint32_t sample, uint32_t seed, float out[3])
{
    v3 d = bounce_direction(v3_from(n), pixel, sample, seed);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}

typedef struct {
    const OrSphere *spheres; uint32_t n_spheres;
    Mesh mesh;            /* flattened: faces index into world-space verts */
} Scene;

/* Nearest hit of a bounce ray over spheres then faces; returns kind: -1 none, else
 * obj id (>= 0 face, -2-k sphere). */
static inline int32_t scene_nearest(const Scene *sc, Ray ray, HitRecord *out)
{
    int32_t id = -1;
    HitRecord best = kNoHit;
    for (uint32_t k = 0; k < sc->n_spheres; k++) {
        HitRecord h = sphere_ray_intersect(v3_from(sc->spheres[k].center), sc->spheres[k].radius, ray);
        if ((!best.hit && h.hit) || (h.hit && h.distance < best.distance)) { best = h; id = -2 - (int32_t)k; }
    }
    if (sc->mesh.n_faces) {
        int i_min;
        HitRecord h = mesh_nearest(&sc->mesh, ray, &i_min);
        if ((!best.hit && h.hit) || (h.hit && h.distance < best.distance)) { best = h; id = i_min; }
    }
    *out = best;
    return id;
}

static inline v3 shade_any(const Scene *sc, int32_t id, const HitRecord *h, Ray ray, v3 *albedo)
{
    if (id >= 0) return shade_mesh(&sc->mesh, (uint32_t)id, h, ray, albedo);
    return shade_sphere(h, ray, albedo);
}

/* Range of one sample's terms (extension; include/rwr_hip.h rwr_render_params): see render_path_core. */
#define OR_PATH_E0_CAP 16.0f
#define OR_PATH_E1_CAP 64.0f
static inline float term_clamp(float x, float cap) { return x > 0.0f ? (x < cap ? x : cap) : 0.0f; }

/* Multi-material form: face i (before instancing) uses materials[face_material[i]] and the
 * texture (tex_ptrs[k], tex_ws[k], tex_hs[k]) of that material; face_material == NULL means
 * "everything uses material 0".  Faces of all parts are one flat list (part order, then face
 * order), so the lowest-index tie rule extends across parts. */
static int render_path_core(const OrCameraInvUniform *cam, const OrScreen *screen, const OrRenderParams *params,
                            const OrSphere *spheres, uint32_t n_spheres,
                            const OrVertex *verts, uint32_t n_verts, const OrFace *faces, uint32_t n_faces,
                            const OrInstance *instances, uint32_t n_instances,
                            const OrMaterial *materials, uint32_t n_materials, const uint32_t *face_material,
                            const uint8_t *const *tex_ptrs, const uint32_t *tex_ws, const uint32_t *tex_hs,
                            const uint8_t *const *nmap_ptrs, const uint32_t *nmap_ws, const uint32_t *nmap_hs,
                            uint32_t row_begin, uint32_t row_end,
                            uint8_t *color_u8, float *depth_out, float *color_f32, int32_t *obj_id, float *hit_t)
{
    const OrMaterial *material = materials;
    const uint8_t *tex_rgba8 = n_materials ? tex_ptrs[0] : NULL;
    const uint32_t tex_w = n_materials ? tex_ws[0] : 0u, tex_h = n_materials ? tex_hs[0] : 0u;
    Tex *texs = (Tex *)calloc(n_materials ? n_materials : 1u, sizeof(Tex));
    if (!texs) return -1;
    Tex *nmaps = (Tex *)calloc(n_materials ? n_materials : 1u, sizeof(Tex));
    if (!nmaps) { free(texs); return -1; }
    for (uint32_t k = 0; k < n_materials; k++) {
        texs[k].rgba = tex_ptrs[k]; texs[k].w = tex_ws[k]; texs[k].h = tex_hs[k];
        build_srgb_lut(texs[k].lut);
        if (nmap_ptrs && nmap_ptrs[k] && nmap_ws[k] && nmap_hs[k]) { nmaps[k].rgba = nmap_ptrs[k]; nmaps[k].w = nmap_ws[k]; nmaps[k].h = nmap_hs[k]; }
    }
    const uint32_t W = screen->width, H = screen->height;
    if (row_end > H) row_end = H;
    /* flatten instances */
    OrVertex *wverts = NULL; OrFace *wfaces = NULL;
    Scene sc;
    sc.spheres = spheres; sc.n_spheres = n_spheres;
    sc.mesh.material = material;
    sc.mesh.tex.rgba = tex_rgba8; sc.mesh.tex.w = tex_w; sc.mesh.tex.h = tex_h;
    build_srgb_lut(sc.mesh.tex.lut);
    sc.mesh.face_material = (n_materials > 1) ? face_material : NULL;
    sc.mesh.n_base_faces = n_faces ? n_faces : 1u;
    sc.mesh.materials = materials;
    sc.mesh.texs = texs;
    sc.mesh.nmaps = nmaps;
    sc.mesh.use_nmap = (params->flags & OR_FLAG_NORMAL_MAP) != 0u;
    if (n_instances && n_faces) {
        wverts = (OrVertex *)malloc((size_t)n_verts * n_instances * sizeof(OrVertex));
        wfaces = (OrFace *)malloc((size_t)n_faces * n_instances * sizeof(OrFace));
        if (!wverts || !wfaces) { free(wverts); free(wfaces); free(texs); free(nmaps); return -1; }
        for (uint32_t k = 0; k < n_instances; k++) {
            for (uint32_t i = 0; i < n_verts; i++) {
                OrVertex v = verts[i];
                v4 p = {v.position[0], v.position[1], v.position[2], 1.0f};
                v4 q = mat4_mul_v4(instances[k].model, p);
                v.position[0] = q.x; v.position[1] = q.y; v.position[2] = q.z;
                wverts[(size_t)k * n_verts + i] = v;
            }
            for (uint32_t i = 0; i < n_faces; i++) {
                OrFace f = faces[i];
                f.indices[0] += k * n_verts; f.indices[1] += k * n_verts; f.indices[2] += k * n_verts;
                wfaces[(size_t)k * n_faces + i] = f;
            }
        }
        sc.mesh.verts = wverts; sc.mesh.n_verts = n_verts * n_instances;
        sc.mesh.faces = wfaces; sc.mesh.n_faces = n_faces * n_instances;
    } else {
        sc.mesh.verts = verts; sc.mesh.n_verts = n_verts; sc.mesh.faces = faces; sc.mesh.n_faces = n_faces;
    }
    const uint32_t spp = params->spp ? params->spp : 1u;
    const int bounce = params->max_bounces >= 1;

#pragma omp parallel for schedule(dynamic, 2)
    for (int y = (int)row_begin; y < (int)row_end; y++) {
        for (uint32_t x = 0; x < W; x++) {
            const uint32_t pixel = (uint32_t)y * W + x;
            const size_t idx = (size_t)pixel;
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            float depth0 = 0.0f, t0 = 0.0f;
            int32_t id0 = -1;
            for (uint32_t s = 0; s < spp; s++) {
                float jx = 0.5f, jy = 0.5f;
                if (spp > 1) {
                    jx = rng_uniform(pixel, s, 0u, params->seed);
                    jy = rng_uniform(pixel, s, 1u, params->seed);
                }
                Ray ray = pixel_to_ray(cam, screen, x, (uint32_t)y, jx, jy);
                /* the reference's passes, depth-composited in registers */
                float depth_tex = 0.0f;
                int32_t id = -1;
                HitRecord win = kNoHit;
                for (uint32_t k = 0; k < n_spheres; k++) {
                    HitRecord h = sphere_ray_intersect(v3_from(spheres[k].center), spheres[k].radius, ray);
                    if (!h.hit) continue;
                    float current_depth = 1.0f - depth_tex;
                    float depth = to_non_linear_depth(h.distance);
                    if (depth >= current_depth) continue;
                    depth_tex = 1.0f - depth; id = -2 - (int32_t)k; win = h;
                }
                if (sc.mesh.n_faces) {
                    int i_min;
                    HitRecord h = mesh_nearest(&sc.mesh, ray, &i_min);
                    if (h.hit) {
                        float current_depth = 1.0f - depth_tex;
                        float depth = to_non_linear_depth(h.distance);
                        if (!(depth >= current_depth)) { depth_tex = 1.0f - depth; id = i_min; win = h; }
                    }
                }
                if (s == 0) { depth0 = depth_tex; id0 = id; t0 = win.hit ? win.distance : 0.0f; }
                if (id == -1) continue;
                v3 albedo;
                v3 e0 = shade_any(&sc, id, &win, ray, &albedo);
                /* the extension's definition: every term a sample adds to a pixel is clamped per channel — E(h0) to
                 * [0, 16], albedo * E(h1) to [0, 64] (NaN counts as 0) — so that sums of terms have a fixed range (the HIP
                 * integrator adds them as fixed point).  Materials with components <= 1 never come near either. */
                if (spp != 1 || bounce) {   /* (spp 1, no bounce IS the reference frame: nothing is clamped before the store) */
                    acc[0] += term_clamp(e0.x, OR_PATH_E0_CAP); acc[1] += term_clamp(e0.y, OR_PATH_E0_CAP); acc[2] += term_clamp(e0.z, OR_PATH_E0_CAP);
                } else {
                    acc[0] += e0.x; acc[1] += e0.y; acc[2] += e0.z;
                }
                acc[3] += 2.0f;
                if (bounce) {
                    v3 P = madd3(win.distance, ray.direction, ray.origin);
                    Ray br;
                    br.origin = V3(P.x + win.normal.x * 1e-4f, P.y + win.normal.y * 1e-4f, P.z + win.normal.z * 1e-4f);
                    br.direction = bounce_direction(win.normal, pixel, s, params->seed);
                    HitRecord h1;
                    int32_t id1 = scene_nearest(&sc, br, &h1);
                    if (id1 != -1) {
                        v3 e1 = shade_any(&sc, id1, &h1, br, NULL);
                        acc[0] += term_clamp(albedo.x * e1.x, OR_PATH_E1_CAP); acc[1] += term_clamp(albedo.y * e1.y, OR_PATH_E1_CAP);
                        acc[2] += term_clamp(albedo.z * e1.z, OR_PATH_E1_CAP);
                    }
                }
            }
            const float fs = (float)spp;
            v3 rgb = V3(acc[0] / fs, acc[1] / fs, acc[2] / fs);
            float alpha = acc[3] / fs;
            if (depth_out) depth_out[idx] = depth0;
            if (color_u8) {
                color_u8[4 * idx + 0] = unorm8(rgb.x); color_u8[4 * idx + 1] = unorm8(rgb.y);
                color_u8[4 * idx + 2] = unorm8(rgb.z); color_u8[4 * idx + 3] = unorm8(alpha);
            }
            if (color_f32) {
                color_f32[4 * idx + 0] = rgb.x; color_f32[4 * idx + 1] = rgb.y;
                color_f32[4 * idx + 2] = rgb.z; color_f32[4 * idx + 3] = alpha;
            }
            if (obj_id) obj_id[idx] = id0;
            if (hit_t) hit_t[idx] = t0;
        }
    }
    free(wverts); free(wfaces); free(texs); free(nmaps);
    return 0;
}

/* Multi-material form (see render_path_core). */
OR_API int or_render_path_mm(const OrCameraInvUniform *cam, const OrScreen *screen, const OrRenderParams *params,
                             const OrSphere *spheres, uint32_t n_spheres,
                             const OrVertex *verts, uint32_t n_verts, const OrFace *faces, uint32_t n_faces,
                             const OrInstance *instances, uint32_t n_instances,
                             const OrMaterial *materials, uint32_t n_materials, const uint32_t *face_material,
                             const uint8_t *const *tex_ptrs, const uint32_t *tex_ws, const uint32_t *tex_hs,
                             uint32_t row_begin, uint32_t row_end,
                             uint8_t *color_u8, float *depth_out, float *color_f32, int32_t *obj_id, float *hit_t)
{
    return render_path_core(cam, screen, params, spheres, n_spheres, verts, n_verts, faces, n_faces, instances, n_instances, materials,
                            n_materials, face_material, tex_ptrs, tex_ws, tex_hs, NULL, NULL, NULL, row_begin, row_end, color_u8,
                            depth_out, color_f32, obj_id, hit_t);
}

/* ... with one optional normal map per material (nmap_ptrs[k] == NULL: none), used when params->flags has
 * OR_FLAG_NORMAL_MAP (extension: normal_mapped()). */
OR_API int or_render_path_nm(const OrCameraInvUniform *cam, const OrScreen *screen, const OrRenderParams *params,
                             const OrSphere *spheres, uint32_t n_spheres,
                             const OrVertex *verts, uint32_t n_verts, const OrFace *faces, uint32_t n_faces,
                             const OrInstance *instances, uint32_t n_instances,
                             const OrMaterial *materials, uint32_t n_materials, const uint32_t *face_material,
                             const uint8_t *const *tex_ptrs, const uint32_t *tex_ws, const uint32_t *tex_hs,
                             const uint8_t *const *nmap_ptrs, const uint32_t *nmap_ws, const uint32_t *nmap_hs,
                             uint32_t row_begin, uint32_t row_end,
                             uint8_t *color_u8, float *depth_out, float *color_f32, int32_t *obj_id, float *hit_t)
{
    return render_path_core(cam, screen, params, spheres, n_spheres, verts, n_verts, faces, n_faces, instances, n_instances, materials,
                            n_materials, face_material, tex_ptrs, tex_ws, tex_hs, nmap_ptrs, nmap_ws, nmap_hs, row_begin, row_end,
                            color_u8, depth_out, color_f32, obj_id, hit_t);
}

OR_API int or_render_path(const OrCameraInvUniform *cam, const OrScreen *screen, const OrRenderParams *params,
                          const OrSphere *spheres, uint32_t n_spheres,
                          const OrVertex *verts, uint32_t n_verts, const OrFace *faces, uint32_t n_faces,
                          const OrInstance *instances, uint32_t n_instances,
                          const OrMaterial *material, const uint8_t *tex_rgba8, uint32_t tex_w, uint32_t tex_h,
                          uint32_t row_begin, uint32_t row_end,
                          uint8_t *color_u8, float *depth_out, float *color_f32, int32_t *obj_id, float *hit_t)
{
    const uint8_t *ptrs[1] = {tex_rgba8};
    return or_render_path_mm(cam, screen, params, spheres, n_spheres, verts, n_verts, faces, n_faces, instances, n_instances,
                             material, 1u, NULL, ptrs, &tex_w, &tex_h, row_begin, row_end, color_u8, depth_out, color_f32,
                             obj_id, hit_t);
}

/* Single-ray probes used by the known-answer tests. */
OR_API void or_pixel_to_ray(const OrCameraInvUniform *cam, const OrScreen *screen, uint32_t x, uint32_t y,
                            float jx, float jy, float out_origin[3], float out_dir[3], float out_view_vec[4])
{
    Ray r = pixel_to_ray(cam, screen, x, y, jx, jy);
    out_origin[0] = r.origin.x; out_origin[1] = r.origin.y; out_origin[2] = r.origin.z;
    out_dir[0] = r.direction.x; out_dir[1] = r.direction.y; out_dir[2] = r.direction.z;
    if (out_view_vec) {
        float x_nds = 2.0f * ((float)x + jx) / (float)screen->width - 1.0f;
        float y_nds = 2.0f * ((float)y + jy) / (float)screen->height - 1.0f;
        v4 pv = {x_nds, y_nds, 1.0f, 1.0f};
        v4 vv = mat4_mul_v4(cam->proj_inv, pv);
        out_view_vec[0] = vv.x; out_view_vec[1] = vv.y; out_view_vec[2] = vv.z; out_view_vec[3] = vv.w;
    }
}

OR_API int or_triangle_ray_intersect(const float p0[3], const float p1[3], const float p2[3],
                                     const float origin[3], const float dir[3],
                                     float *t, float normal[3], float bary[3])
{
    Ray r; r.origin = v3_from(origin); r.direction = v3_from(dir);
    HitRecord h = triangle_ray_intersect(v3_from(p0), v3_from(p1), v3_from(p2), r);
    if (!h.hit) return 0;
    *t = h.distance;
    normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
    bary[0] = h.barycentric.x; bary[1] = h.barycentric.y; bary[2] = h.barycentric.z;
    return 1;
}

OR_API int or_sphere_ray_intersect(const float center[3], float radius, const float origin[3], const float dir[3],
                                   float *t, float normal[3])
{
    Ray r; r.origin = v3_from(origin); r.direction = v3_from(dir);
    HitRecord h = sphere_ray_intersect(v3_from(center), radius, r);
    if (!h.hit) return 0;
    *t = h.distance;
    normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
    return 1;
}

OR_API float or_to_non_linear_depth(float t) { return to_non_linear_depth(t); }
OR_API uint8_t or_unorm8(float c) { return unorm8(c); }
OR_API void or_srgb_lut(float lut[256]) { build_srgb_lut(lut); }
OR_API void or_tex_sample(const uint8_t *rgba, uint32_t w, uint32_t h, float u, float v, float out[3])
{
    Tex t; t.rgba = rgba; t.w = w; t.h = h; build_srgb_lut(t.lut);
    v3 c = tex_sample_bilinear(&t, u, v);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

/* ----------------------------------------------- host-side camera maths -- */
/* cgmath 0.18.0 is not vendored in /root/reference; these restate its
 * published formulas with Rust's evaluation order (no fusion). */
typedef struct { float m[4][4]; } M4; /* column-major [col][row] */

static inline float cg_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cg_cross(v3 a, v3 b)
{
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float cg_magnitude(v3 a) { return sqrtf(cg_dot(a, a)); }
static inline v3 cg_normalize(v3 a) { return scale3(a, 1.0f / cg_magnitude(a)); } /* normalize_to(1) */

/* Matrix4::look_to_rh(eye, center - eye, up) — camera.rs:15,21 */
static M4 cg_look_at_rh(v3 eye, v3 center, v3 up)
{
    v3 f = cg_normalize(sub3(center, eye));
    v3 s = cg_normalize(cg_cross(f, up));
    v3 u = cg_cross(s, f);
    M4 r = {{{s.x, u.x, -f.x, 0.0f},
             {s.y, u.y, -f.y, 0.0f},
             {s.z, u.z, -f.z, 0.0f},
             {-cg_dot(eye, s), -cg_dot(eye, u), cg_dot(eye, f), 1.0f}}};
    return r;
}

/* cgmath::perspective(Deg(fovy), aspect, near, far) — camera.rs:16,27 */
static M4 cg_perspective(float fovy_deg, float aspect, float near, float far)
{
    float fovy_rad = fovy_deg * (float)(3.14159265358979323846 / 180.0);
    float f = 1.0f / tanf(fovy_rad / 2.0f);
    M4 r;
    memset(&r, 0, sizeof r);
    r.m[0][0] = f / aspect;
    r.m[1][1] = f;
    r.m[2][2] = (far + near) / (near - far);
    r.m[2][3] = -1.0f;
    r.m[3][2] = (2.0f * far * near) / (near - far);
    return r;
}

static inline float det3(float a00, float a01, float a02, float a10, float a11, float a12, float a20, float a21, float a22)
{
    /* columns a0*, a1*, a2* */
    return a00 * (a11 * a22 - a21 * a12) - a10 * (a01 * a22 - a21 * a02) + a20 * (a01 * a12 - a11 * a02);
}

/* Matrix4::invert: adjugate / determinant. Returns 0 when singular. */
static int cg_invert(const M4 *a, M4 *out)
{
    const float (*m)[4] = a->m;
    float cof[4][4]; /* cof[c][r]: cofactor of element (col c,row r) */
    for (int c = 0; c < 4; c++) {
        for (int r = 0; r < 4; r++) {
            float s[3][3];
            int cc = 0;
            for (int c2 = 0; c2 < 4; c2++) {
                if (c2 == c) continue;
                int rr = 0;
                for (int r2 = 0; r2 < 4; r2++) {
                    if (r2 == r) continue;
                    s[cc][rr++] = m[c2][r2];
                }
                cc++;
            }
            float d = det3(s[0][0], s[0][1], s[0][2], s[1][0], s[1][1], s[1][2], s[2][0], s[2][1], s[2][2]);
            cof[c][r] = ((c + r) & 1) ? -d : d;
        }
    }
    float det = m[0][0] * cof[0][0] + m[1][0] * cof[1][0] + m[2][0] * cof[2][0] + m[3][0] * cof[3][0];
    if (det == 0.0f) return 0;
    float inv_det = 1.0f / det;
    /* inverse = adjugate/det; adjugate = transpose of cofactor matrix */
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            out->m[c][r] = cof[r][c] * inv_det;
    return 1;
}

static M4 cg_mul(const M4 *a, const M4 *b)
{
    M4 r;
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 4; i++)
            r.m[c][i] = a->m[0][i] * b->m[c][0] + a->m[1][i] * b->m[c][1] + a->m[2][i] * b->m[c][2] + a->m[3][i] * b->m[c][3];
    return r;
}

/* CameraInvUniform::update_view_proj — src/lib.rs:105-111 (note G * P^-1, the quirk). */
OR_API int or_camera_build_inv_uniform(const OrCamera *cam, OrCameraInvUniform *out)
{
    static const M4 G = {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 0.5f, 0}, {0, 0, 0.5f, 1}}}; /* lib.rs:31-37 */
    M4 view = cg_look_at_rh(v3_from(cam->eye), v3_from(cam->target), v3_from(cam->up));
    M4 proj = cg_perspective(cam->fovy, cam->aspect, cam->znear, cam->zfar);
    M4 view_inv, proj_inv;
    if (!cg_invert(&view, &view_inv)) return -1;
    if (!cg_invert(&proj, &proj_inv)) return -1;
    M4 gp = cg_mul(&G, &proj_inv);
    memcpy(out->viewmodel_inv, view_inv.m, sizeof view_inv.m);
    memcpy(out->proj_inv, gp.m, sizeof gp.m);
    out->origin[0] = cam->eye[0]; out->origin[1] = cam->eye[1]; out->origin[2] = cam->eye[2];
    out->_padding = 0;
    return 0;
}

/* CircleCameraController::update_camera — src/circle_camera_control.rs:76-105.
 * keys bit0 forward, bit1 backward, bit2 left, bit3 right (up/down are recorded
 * by the reference but never used). */
OR_API void or_controller_update(float speed, uint32_t keys, OrCamera *camera)
{
    v3 eye = v3_from(camera->eye), target = v3_from(camera->target), up = v3_from(camera->up);
    v3 forward = sub3(target, eye);
    v3 forward_norm = cg_normalize(forward);
    float forward_mag = cg_magnitude(forward);
    if ((keys & 1u) && forward_mag > speed) eye = add3(eye, scale3(forward_norm, speed));
    if (keys & 2u) eye = sub3(eye, scale3(forward_norm, speed));
    v3 right = cg_cross(forward_norm, up);
    forward = sub3(target, eye);
    forward_mag = cg_magnitude(forward);
    if (keys & 8u) eye = sub3(target, scale3(cg_normalize(add3(forward, scale3(right, speed))), forward_mag));
    if (keys & 4u) eye = sub3(target, scale3(cg_normalize(sub3(forward, scale3(right, speed))), forward_mag));
    camera->eye[0] = eye.x; camera->eye[1] = eye.y; camera->eye[2] = eye.z;
}

OR_API void or_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

OR_API int or_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
