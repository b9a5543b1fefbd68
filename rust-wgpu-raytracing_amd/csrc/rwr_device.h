// Device-side restatement of the reference shaders' per-ray functions.
//
// ARITHMETIC CONTRACT (DESIGN.md §"Numerics"): this translation unit is built
// with -ffp-contract=off and without fast-math; f32 divide and sqrt are the
// IEEE-correct expansions (-fhip-fp32-correctly-rounded-divide-sqrt); f32
// denormals are kept (gfx950 default).  Every expression that decides WHICH
// surface a pixel shows (ray generation, hit tests, nearest selection, depth
// compositing) is written operation-for-operation as the WGSL has it, so those
// results are bit-identical to a literal (unfused) evaluation of the shader.
// Conservative culling code (tile frusta, BVH boxes) is free to use FMA: it can
// only skip faces no ray of the wave can hit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rwr_internal.h"

#define RWR_DEV __device__ __forceinline__

namespace rwr {

struct f3 { float x, y, z; };
// 2-wide vectors: clang's ext-vector operators apply the scalar operation to each element (one IEEE
// rounding per operation, no contraction in this translation unit) and map to v_pk_*_f32.
typedef float f2 __attribute__((ext_vector_type(2)));
RWR_DEV f2 splat(float s) { return f2{s, s}; }
RWR_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }  // v_pk_fma_f32: colour / culling code only
RWR_DEV f2 ld2(const float *p) { return f2{p[0], p[1]}; }

RWR_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RWR_DEV f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
RWR_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RWR_DEV f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RWR_DEV f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
// WGSL dot / cross, literal: products rounded, sums left to right.
RWR_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RWR_DEV f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RWR_DEV f3 normalize3(f3 a)
{
    float len = sqrtf(dot3(a, a));
    return mk3(a.x / len, a.y / len, a.z / len);
}
// normalize3's three divisions in 18 instead of 33 instructions (and its sqrt in 5 instead of 14), same bits, for vectors whose components all lie
// in [2^-40, 2^40] in magnitude (normalize_fast_domain).  a.x / len, a.y / len, a.z / len share the
// denominator, and this is the compiler's own IEEE expansion of each quotient — v_div_scale (x2),
// v_rcp, Newton step, q = n*r, two FMA-residual corrections (the second one as v_div_fmas),
// v_div_fixup — with the reciprocal refinement done once and the three instructions left out that
// are the identity on such operands (ISA: v_div_scale rescales only zero / denormal / tiny numerators
// (< 2^-103), denormal quotients (here every quotient is at least 2^-81) or exponent differences
// >= 96, and then sets VCC for v_div_fmas; v_div_fixup replaces the quotient only for zero / infinite /
// NaN operands or a quotient outside the exponent range).  rwr_selftest_exact_math() compares it with normalize3 on 2^30
// pseudo-random in-domain vectors (tests/test_gpu_exact_math.py).
RWR_DEV bool normalize_fast_domain(f3 a)
{
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    return lo >= 0x1p-40f && hi <= 0x1p40f;
}
RWR_DEV float div_shared_rcp(float n, float d, float r)
{
    const float q0 = n * r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), r, q1);
}
// sqrtf in 5 instead of 14 instructions, same bits, for 2^-100 <= x <= 2^100: v_rsq_f32, g = x * y,
// one FMA-residual correction with h = y / 2.  By exhaustion (tools/ubench/exact_sqrt.hip: the only
// mismatches against the IEEE expansion are below 2^-102, where the residual underflows;
// rwr_selftest_exact_math() re-checks every float of the range).  dot(a, a) of a vector of
// normalize_fast_domain lies in [2^-80, 2^82].
RWR_DEV float sqrt_fast(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}
RWR_DEV f3 normalize3_fast(f3 a)
{
    const float len = sqrt_fast(dot3(a, a));
    float r = __builtin_amdgcn_rcpf(len);
    r = __builtin_fmaf(__builtin_fmaf(-len, r, 1.0f), r, r);
    return mk3(div_shared_rcp(a.x, len, r), div_shared_rcp(a.y, len, r), div_shared_rcp(a.z, len, r));
}
// origin + t * direction
RWR_DEV f3 along(f3 o, float t, f3 d) { return mk3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z); }

// compute.wgsl:51-53
constexpr float kNear = 0.01f;
constexpr float kFar = 100.0f;
constexpr float kEpsilon = 0.000001f;

// compute.wgsl:78-80
RWR_DEV float to_non_linear_depth(float depth)
{
    return ((1.0f / depth) - (1.0f / kNear)) / ((1.0f / kFar) - (1.0f / kNear));
}

// The same value in 8 instead of 22 VALU instructions, for 2^-126 <= depth < 2^126 (depth_fast_domain):
//   1 / depth:  v_rcp_f32 + one Newton step equals the IEEE quotient for every such depth on gfx950;
//   x / C, C = 1/kFar - 1/kNear:  q = x * RN(1/C) corrected once by its FMA residual equals the IEEE
//   quotient for x = 0 and every |x| >= 2^-103 — and x = RN(1/depth) - 100 is 0 or at least 2^-18.
// Both by exhaustion over all 2^32 inputs: tools/ubench/exact_div.hip, and rwr_selftest_exact_math()
// (kernels_selftest.hip, run by tests/test_gpu_exact_math.py) checks this very function against
// to_non_linear_depth for every float of the domain.
constexpr float kDepthC = (1.0f / kFar) - (1.0f / kNear);
constexpr float kDepthRC = 1.0f / kDepthC;
RWR_DEV bool depth_fast_domain(float depth) { return (__float_as_uint(depth) - 0x00800000u) < 0x7e000000u; }
RWR_DEV float to_non_linear_depth_fast(float depth)
{
    float r = __builtin_amdgcn_rcpf(depth);
    r = __builtin_fmaf(__builtin_fmaf(-depth, r, 1.0f), r, r);
    const float x = r - (1.0f / kNear);
    const float q = x * kDepthRC;
    return __builtin_fmaf(__builtin_fmaf(-kDepthC, q, x), kDepthRC, q);
}

// to_non_linear_depth for the active lanes: the short form when every one of their distances is in its
// domain (a wave-uniform choice; the same bits either way).
RWR_DEV float to_non_linear_depth_auto(float depth)
{
    if (__all(depth_fast_domain(depth))) return to_non_linear_depth_fast(depth);
    return to_non_linear_depth(depth);
}

// mat4x4 * vec4 with column-major m[col][row]; WGSL: m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]*v.w
RWR_DEV void mat4_mul(const float (&m)[4][4], float vx, float vy, float vz, float vw,
                      float &rx, float &ry, float &rz, float &rw)
{
    rx = m[0][0] * vx + m[1][0] * vy + m[2][0] * vz + m[3][0] * vw;
    ry = m[0][1] * vx + m[1][1] * vy + m[2][1] * vz + m[3][1] * vw;
    rz = m[0][2] * vx + m[1][2] * vy + m[2][2] * vz + m[3][2] * vw;
    rw = m[0][3] * vx + m[1][3] * vy + m[2][3] * vz + m[3][3] * vw;
}

// World-space direction of the ray through pixel-space point (fx, fy) BEFORE
// normalisation — compute.wgsl:151-159.
RWR_DEV f3 ray_dir_unnormalized(const rwr_camera_inv_uniform &cam, float fx, float fy, float width, float height)
{
    float x_nds = 2.0f * fx / width - 1.0f;
    float y_nds = 2.0f * fy / height - 1.0f;
    float vx, vy, vz, vw;
    mat4_mul(cam.proj_inv, x_nds, y_nds, 1.0f, 1.0f, vx, vy, vz, vw);
    vw = 0.0f;
    float wx, wy, wz, ww;
    mat4_mul(cam.viewmodel_inv, vx, vy, vz, vw, wx, wy, wz, ww);
    return mk3(wx, wy, wz);
}

// pixelToRay, compute.wgsl:150-164; (jx,jy) = (0.5,0.5) in the reference.
RWR_DEV f3 pixel_to_ray_dir(const rwr_camera_inv_uniform &cam, uint32_t x, uint32_t y, float jx, float jy,
                            uint32_t width, uint32_t height)
{
    const f3 w = ray_dir_unnormalized(cam, (float)x + jx, (float)y + jy, (float)width, (float)height);
    if (__all(normalize_fast_domain(w))) return normalize3_fast(w);  // same bits, fewer instructions (see normalize3_fast)
    return normalize3(w);
}

// FMA dot product and hardware-rsq normalisation for the colour path only.
RWR_DEV float cdot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
RWR_DEV f3 cnormalize(f3 a)
{
    const float r = __builtin_amdgcn_rsqf(cdot(a, a));
    return mk3(a.x * r, a.y * r, a.z * r);
}

// sphereRayIntersect, sphere/compute.wgsl:63-85: hit and distance (the normal, which
// feeds only colour / the bounce, is derived from t by the caller).
RWR_DEV bool sphere_ray_intersect_t(f3 center, float radius, f3 O, f3 D, float &t_out)
{
    f3 oc = sub3(O, center);
    float a = dot3(D, D);
    float b = 2.0f * dot3(oc, D);
    float c = dot3(oc, oc) - (radius * radius);
    float discriminant = b * b - 4.0f * a * c;
    if (discriminant < 0.0f) return false;
    float sq = sqrtf(discriminant);
    float t1 = (-b - sq) / (2.0f * a);
    float t2 = (-b + sq) / (2.0f * a);
    if (t1 >= 0.0f) t_out = t1;
    else if (t2 >= 0.0f) t_out = t2;
    else return false;
    return true;
}

// pow(x, 32) by five squarings (x >= 0).  Within 16 ulp of the correctly rounded
// power; colours are compared at 1e-4 absolute (DESIGN.md §"Numerics").
RWR_DEV float pow32(float x)
{
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
    return x16 * x16;
}

// Sphere shading, sphere/compute.wgsl:137-152.
RWR_DEV f3 shade_sphere(f3 n, f3 D)
{
    const f3 nl = neg3(normalize3(mk3(1.0f, -5.0f, 1.0f)));  // -normalize(kLightDir), :41 (folded at compile time)
    float diffuse = 1.0f * fmaxf(0.0f, cdot(n, nl));
    f3 half_dir = cnormalize(sub3(nl, D));
    float specular = 0.5f * pow32(fmaxf(0.0f, cdot(half_dir, n)));
    float k = 0.1f + diffuse;
    return mk3(k * 1.0f + specular, k * 0.0f + specular, k * 0.0f + specular);
}

// -normalize(kLightDir) of the mesh shader, compute.wgsl:55 (folded at compile time).
RWR_DEV f3 mesh_light_dir() { return neg3(normalize3(mk3(1.0f, -1.0f, -5.0f))); }

// Rgba8UnormSrgb texel fetch + bilinear filter with ClampToEdge
// (texture.rs:122,151-159; textureSampleGrad at LOD 0, compute.wgsl:225) at texel-space
// position fxy = (u * tex_w - 0.5, v * tex_h - 0.5).
// `tex` holds the texels already decoded to linear f32 (one float4 per texel, made at
// upload from the same 256-entry sRGB table the oracle uses): what the sampler
// hardware does before filtering, done once instead of per tap.  pitch = 16 * tex_w bytes
// (< 2^24, checked at upload: row * pitch is a v_mul_u32_u24, + column * 16 a v_lshl_add_u32).
// v_med3 returns a finite operand when fx / fy is NaN, so no tap can leave the texture.
struct TexTaps { uint32_t o00, o10, o01, o11; f2 a; };  // byte offsets of the 2x2 footprint, fractional position
RWR_DEV TexTaps tex_taps(uint32_t pitch, float wmax, float hmax, f2 fxy)
{
    const f2 f0 = f2{floorf(fxy.x), floorf(fxy.y)};
    const f2 f1 = f0 + 1.0f;
    const uint32_t x0 = (uint32_t)__builtin_amdgcn_fmed3f(f0.x, 0.0f, wmax) << 4;
    const uint32_t x1 = (uint32_t)__builtin_amdgcn_fmed3f(f1.x, 0.0f, wmax) << 4;
    const uint32_t y0 = (uint32_t)__builtin_amdgcn_fmed3f(f0.y, 0.0f, hmax);
    const uint32_t y1 = (uint32_t)__builtin_amdgcn_fmed3f(f1.y, 0.0f, hmax);
    TexTaps t;
    t.o00 = __umul24(y0, pitch) + x0; t.o10 = __umul24(y0, pitch) + x1;
    t.o01 = __umul24(y1, pitch) + x0; t.o11 = __umul24(y1, pitch) + x1;
    t.a = fxy - f0;
    return t;
}
RWR_DEV f3 tex_filter(const float4 *__restrict__ tex, const TexTaps &t)
{
    const char *base = reinterpret_cast<const char *>(tex);
    const float4 t00 = *reinterpret_cast<const float4 *>(base + t.o00);
    const float4 t10 = *reinterpret_cast<const float4 *>(base + t.o10);
    const float4 t01 = *reinterpret_cast<const float4 *>(base + t.o01);
    const float4 t11 = *reinterpret_cast<const float4 *>(base + t.o11);
    const f2 a = t.a, b = 1.0f - a;
    const float w00 = b.x * b.y, w10 = a.x * b.y, w01 = b.x * a.y, w11 = a.x * a.y;
    // (r, g) as a pair (adjacent in the loaded texel), b alone
    const f2 rg = fma2(f2{t11.x, t11.y}, splat(w11), fma2(f2{t01.x, t01.y}, splat(w01),
                  fma2(f2{t10.x, t10.y}, splat(w10), f2{t00.x, t00.y} * w00)));
    const float bl = __builtin_fmaf(t11.z, w11, __builtin_fmaf(t01.z, w01, __builtin_fmaf(t10.z, w10, t00.z * w00)));
    return mk3(rg.x, rg.y, bl);
}
RWR_DEV f3 tex_sample_bilinear(const float4 *__restrict__ tex, uint32_t pitch, float wmax, float hmax, f2 fxy)
{
    return tex_filter(tex, tex_taps(pitch, wmax, hmax, fxy));
}

// Mesh shading, triangle_list/compute.wgsl:217-234 (colour path: see the header), in three steps
// so that callers with two pixels per lane can run the middle one packed.
//
// View-dependent, surface-independent part of Blinn-Phong: h = L - D (un-normalised half vector,
// :229) and 1 / |h|.
struct HalfVec { f3 h; float rh; };
RWR_DEV HalfVec mesh_half_vector(f3 D)
{
    HalfVec hv;
    hv.h = sub3(mesh_light_dir(), D);
    hv.rh = __builtin_amdgcn_rsqf(cdot(hv.h, hv.h));
    return hv;
}
// Lambert term max(0, n.L) (:227) and the Blinn-Phong base max(0, dot(half_dir, n)) (:230) for the normal
// flipped towards the ray (N = -N when dot(N, D) > 0, :140-142): the flip is a sign flip of both dot
// products, applied as an XOR with the sign bit of dot(N, D) (never +-0 for an accepted hit, :94).
RWR_DEV void mesh_light_terms(const ShadeRec &S, float ndotd, const HalfVec &hv, float &ndl, float &hn)
{
    const uint32_t sb = __float_as_uint(ndotd) & 0x80000000u;
    ndl = fmaxf(-__uint_as_float(__float_as_uint(S.ndl0) ^ sb), 0.0f);
    const float c = cdot(hv.h, ld3(S.n)) * hv.rh;
    hn = fmaxf(-__uint_as_float(__float_as_uint(c) ^ sb), 0.0f);
}
// Filtered diffuse texel at the hit: (eu, ev) are the winner's un-normalised edge functions
// (:126,135); the barycentric division, the tex_coords interpolation, the v flip and the texel-space
// scale (:144-147,218-225) are the affine map prebaked into the ShadeRec.
RWR_DEV f2 mesh_texel_pos(const ShadeRec &S, float eu, float ev)
{
    return fma2(splat(ev), ld2(S.c2), fma2(splat(eu), ld2(S.c1), ld2(S.c0)));
}
RWR_DEV f3 mesh_texel(const ShadeRec &S, float eu, float ev, const float4 *__restrict__ tex, uint32_t pitch, float wmax,
                      float hmax)
{
    return tex_sample_bilinear(tex, pitch, wmax, hmax, mesh_texel_pos(S, eu, ev));
}
// (ambient + texel * n.L) + specular * pow(., 32), :231-233
RWR_DEV f3 mesh_combine(f3 texel, float ndl, float sp, const float *ka, const float *ks)
{
    return mk3(__builtin_fmaf(ks[0], sp, __builtin_fmaf(texel.x, ndl, ka[0])),
               __builtin_fmaf(ks[1], sp, __builtin_fmaf(texel.y, ndl, ka[1])),
               __builtin_fmaf(ks[2], sp, __builtin_fmaf(texel.z, ndl, ka[2])));
}
// Normal-mapped light terms (extension, RWR_FLAG_NORMAL_MAP; defined in oracle/rt_oracle.c normal_mapped): m = 2 c - 1 from
// the filtered LINEAR texel c of the map, n' = normalize(t m.x + b m.y + s n m.z) with s = -1 when the flat normal was
// flipped towards the ray (:140-142), n' = s n when the sum vanishes; n' replaces the flat normal in :226-229.
RWR_DEV void mesh_light_terms_nm(const ShadeRec &S, const TangentRec &G, f3 c, float ndotd, const HalfVec &hv, float &ndl, float &hn)
{
    const float s = ndotd > 0.0f ? -1.0f : 1.0f;
    const f3 m = mk3(2.0f * c.x - 1.0f, 2.0f * c.y - 1.0f, (2.0f * c.z - 1.0f) * s);
    f3 n = mk3(__builtin_fmaf(G.t[0], m.x, __builtin_fmaf(G.b[0], m.y, S.n[0] * m.z)),
               __builtin_fmaf(G.t[1], m.x, __builtin_fmaf(G.b[1], m.y, S.n[1] * m.z)),
               __builtin_fmaf(G.t[2], m.x, __builtin_fmaf(G.b[2], m.y, S.n[2] * m.z)));
    const float l2 = cdot(n, n);
    if (!(l2 > 1e-20f)) n = mk3(S.n[0] * s, S.n[1] * s, S.n[2] * s);
    else { const float r = __builtin_amdgcn_rsqf(l2); n = mk3(n.x * r, n.y * r, n.z * r); }
    ndl = fmaxf(cdot(n, mesh_light_dir()), 0.0f);
    hn = fmaxf(cdot(hv.h, n) * hv.rh, 0.0f);
}
// texel-space position in a part's normal map of the diffuse texel-space position `pos` (the maps may differ in size)
RWR_DEV f2 nmap_texel_pos(const MaterialRec &M, f2 pos)
{
    const f2 scale = f2{(float)M.nmap_w / (float)M.tex_w, (float)M.nmap_h / (float)M.tex_h};
    return fma2(pos + 0.5f, scale, splat(-0.5f));
}

struct Shaded { f3 colour, albedo; };  // local shading E and the surface's diffuse reflectance
// G, Mn: the face's tangent frame and its material when the render asked for normal-mapped shading, else nullptr
RWR_DEV Shaded shade_mesh(const ShadeRec &S, float eu, float ev, float ndotd, f3 D, const float *ka, const float *ks,
                          const float4 *__restrict__ tex, uint32_t pitch, float wmax, float hmax,
                          const TangentRec *__restrict__ G = nullptr, const MaterialRec *__restrict__ Mn = nullptr)
{
    float ndl, hn;
    const HalfVec hv = mesh_half_vector(D);
    const f2 pos = mesh_texel_pos(S, eu, ev);
    if (G && Mn && Mn->nmap) {
        const f3 c = tex_sample_bilinear(Mn->nmap, Mn->nmap_w * 16u, (float)(Mn->nmap_w - 1u), (float)(Mn->nmap_h - 1u), nmap_texel_pos(*Mn, pos));
        mesh_light_terms_nm(S, *G, c, ndotd, hv, ndl, hn);
    } else {
        mesh_light_terms(S, ndotd, hv, ndl, hn);
    }
    Shaded r;
    r.albedo = tex_sample_bilinear(tex, pitch, wmax, hmax, pos);
    r.colour = mesh_combine(r.albedo, ndl, pow32(hn), ka, ks);
    return r;
}

// rgba8unorm store conversion, one v_cvt_pk_u8_f32 per channel: it saturates to [0, 255] (NaN -> 0)
// and rounds to nearest, ties to even — the rule the oracle states for the store (oracle/rt_oracle.c
// unorm8; the WebGPU / Vulkan float -> UNORM conversion leaves exact ties to the implementation).
RWR_DEV uint32_t pack_rgba8_scaled(float r255, float g255, float b255, uint32_t alpha_bits)
{
    uint32_t v = alpha_bits;
    v = __builtin_amdgcn_cvt_pk_u8_f32(r255, 0, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(g255, 1, v);
    return __builtin_amdgcn_cvt_pk_u8_f32(b255, 2, v);
}
RWR_DEV uint32_t pack_rgba8(float r, float g, float b, float a)
{
    const uint32_t alpha = __builtin_amdgcn_cvt_pk_u8_f32(a * 255.0f, 3, 0u);
    return pack_rgba8_scaled(r * 255.0f, g * 255.0f, b * 255.0f, alpha);
}

// Running nearest-hit state of the face loop (compute.wgsl:186-202).
struct MeshHit {
    bool have;
    float t;      // distance
    float u, v;   // un-normalised edge functions of the winner (compute.wgsl:126,135)
    float ndotd;  // N . D of the winner (sign decides the normal flip, :140)
    uint32_t idx; // i_min
};

// One iteration of the loop: triangleRayIntersect (compute.wgsl:82-148) against a
// prebaked record, then the selection rule of :198-201.  Written branch-free:
// every early `return kNoHit` of the shader becomes a term of `hit`, with the
// same comparison sense (a NaN never rejects, exactly as `x < 0.0` in WGSL), so
// the accepted set is identical while the whole record is fetched up front and
// 64 rays proceed in lock-step.  `T` may live in SGPRs (wave-uniform face) or
// VGPRs (per-lane face).
RWR_DEV void intersect_and_select(const TriRecord &T, uint32_t idx, f3 O, f3 D, MeshHit &best)
{
    const f3 N = ld3(T.N);
    const float ndotd = dot3(N, D);
    bool hit = !(fabsf(ndotd) < kEpsilon);              // :94
    const float t = -(dot3(N, O) + T.d) / ndotd;        // :99-102
    hit &= !(t < 0.0f);                                 // :105
    const f3 P = along(O, t, D);                        // :110
    f3 C = cross3(ld3(T.e0), sub3(P, ld3(T.p0)));       // :115-117
    hit &= !(dot3(N, C) < 0.0f);                        // :118
    C = cross3(ld3(T.e1), sub3(P, ld3(T.p1)));          // :123-125
    const float u = dot3(N, C);
    hit &= !(u < 0.0f);                                 // :127
    C = cross3(ld3(T.e2), sub3(P, ld3(T.p2)));          // :132-134
    const float v = dot3(N, C);
    hit &= !(v < 0.0f);                                 // :136
    // (!min_hit.hit && hit) || (hit && distance < min_hit.distance), :198; callers
    // visit faces in ascending index order, so ties keep the lowest index.
    if (hit && (!best.have || t < best.t)) {
        best.have = true;
        best.t = t;
        best.u = u;
        best.v = v;
        best.ndotd = ndotd;
        best.idx = idx;
    }
}

// ---------------------------------------------------------------------------
// Extension (no reference counterpart; DESIGN.md "Extended integrator"): the
// counter-based RNG and the cosine-distributed bounce direction.  Integer hash +
// only + - * / sqrt on floats, written operation for operation like the oracle, so
// CPU and GPU agree bit for bit and the result never depends on tile, band or rank.
RWR_DEV uint32_t rng_mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
RWR_DEV uint32_t rng_hash(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed)
{
    uint32_t h = seed ^ 0x9E3779B9u;
    h = rng_mix(h ^ pixel);
    h = rng_mix(h ^ (sample * 0x85EBCA6Bu));
    h = rng_mix(h ^ (dim * 0xC2B2AE35u));
    return h;
}
RWR_DEV float rng_uniform(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed)
{
    return (float)(rng_hash(pixel, sample, dim, seed) >> 8) * (1.0f / 16777216.0f);
}

RWR_DEV f3 bounce_direction(f3 n, uint32_t pixel, uint32_t sample, uint32_t seed)
{
    float a = 0.0f, b = 0.0f;
    for (uint32_t k = 0; k < 8u; k++) {  // rejection-sample the unit disk
        const float ua = 2.0f * rng_uniform(pixel, sample, 2u + 2u * k, seed) - 1.0f;
        const float ub = 2.0f * rng_uniform(pixel, sample, 3u + 2u * k, seed) - 1.0f;
        if (ua * ua + ub * ub <= 1.0f) { a = ua; b = ub; break; }
    }
    const float dz = sqrtf(fmaxf(0.0f, 1.0f - a * a - b * b));
    const float sign = copysignf(1.0f, n.z);
    const float aa = -1.0f / (sign + n.z);
    const float bb = n.x * n.y * aa;
    const f3 b1 = mk3(1.0f + sign * n.x * n.x * aa, sign * bb, -sign * n.x);
    const f3 b2 = mk3(bb, sign + n.y * n.y * aa, -n.y);
    const f3 d = mk3(a * b1.x + b * b2.x + dz * n.x, a * b1.y + b * b2.y + dz * n.y, a * b1.z + b * b2.z + dz * n.z);
    return normalize3(d);
}

// Wavefront integrator: the key a tile's ray pool is sorted by (kernels_wf_primary.hip stores it, kernels_wf_bounce.hip sorts).
// Direction bin: 3 bits of octant (Gray-coded so that neighbours share two signs) and 2 x kWfDirCellBits bits of position
// inside the octant's triangle of the octahedral map (Morton order of a 2^bits x 2^bits grid).
RWR_DEV uint32_t wf_direction_bin(f3 D)
{
    const uint32_t sx = __float_as_uint(D.x) >> 31, sy = __float_as_uint(D.y) >> 31, sz = __float_as_uint(D.z) >> 31;
    const uint32_t oct = sz * 4u + (sy ^ sz) * 2u + (sx ^ sy);  // reflected Gray code of (sz, sy, sx)
    const float ax = fabsf(D.x), ay = fabsf(D.y), az = fabsf(D.z);
    const float inv = __builtin_amdgcn_rcpf(ax + ay + az + 1e-30f);
    constexpr uint32_t kCells = 1u << kWfDirCellBits;
    const uint32_t iu = min(kCells - 1u, (uint32_t)(ax * inv * (float)kCells)), iv = min(kCells - 1u, (uint32_t)(ay * inv * (float)kCells));
    uint32_t mu = 0, mv = 0;   // Morton order inside the octant
#pragma unroll
    for (uint32_t b = 0; b < kWfDirCellBits; b++) { mu |= (iu & (1u << b)) << b; mv |= (iv & (1u << b)) << b; }
    return oct * (kCells * kCells) + (mu | (mv << 1));
}

// Wavefront ray records (rwr_internal.h WfBuffers::rays): the throughput's three channels as unorm16 in the two w components.
RWR_DEV float wf_pack_unorm16x2(float a, float b)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 pk = __builtin_amdgcn_cvt_pknorm_u16(a, b);   // v_cvt_pknorm_u16_f32: clamps to [0, 1], rounds to nearest
    return __uint_as_float((uint32_t)pk.x | ((uint32_t)pk.y << 16));
}
RWR_DEV float wf_unorm16_lo(float w) { return (float)(__float_as_uint(w) & 0xffffu) * (1.0f / 65535.0f); }
RWR_DEV float wf_unorm16_hi(float w) { return (float)(__float_as_uint(w) >> 16) * (1.0f / 65535.0f); }

}  // namespace rwr
