// Two pixels per lane: the per-ray functions of rwr_device.h on 2-wide vectors.
//
// Why: the frame kernel is VALU-bound and on gfx950 a wave64 f32 VALU instruction
// occupies its SIMD for 4 cycles whether it is plain or packed — v_pk_mul_f32 /
// v_pk_add_f32 do two operations in the same 4 cycles (tools/ubench/valu_rate.hip).
// Giving every lane two horizontally adjacent pixels turns the multiplies, adds
// and subtracts of ray generation and of the hit test into packed instructions.
//
// Numerics are unchanged: clang's ext-vector operators apply the scalar operation
// to each element (one IEEE rounding per operation, no contraction in this
// translation unit; vector divide and sqrt are scalarised into the same IEEE-
// correct expansions), so every element goes through exactly the operation
// sequence of the scalar code and the results are bit-identical to it.
#pragma once

#include "rwr_device.h"

namespace rwr {

typedef int i2 __attribute__((ext_vector_type(2)));        // comparison masks: -1 / 0 per element
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

struct v3 { f2 x, y, z; };

RWR_DEV v3 splat3(f3 a) { return v3{splat(a.x), splat(a.y), splat(a.z)}; }
RWR_DEV v3 sub3(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
// WGSL dot / cross, literal (cf. rwr_device.h)
RWR_DEV f2 dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RWR_DEV v3 cross3(v3 a, v3 b) { return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RWR_DEV v3 along(v3 o, f2 t, v3 d) { return v3{o.x + t * d.x, o.y + t * d.y, o.z + t * d.z}; }
RWR_DEV f2 abs2(f2 a) { return __builtin_elementwise_abs(a); }
RWR_DEV f2 sqrt2(f2 a) { return f2{sqrtf(a.x), sqrtf(a.y)}; }
RWR_DEV f3 lane3(v3 a, int k) { return k ? mk3(a.x.y, a.y.y, a.z.y) : mk3(a.x.x, a.y.x, a.z.x); }
RWR_DEV bool any2(i2 m) { return (m.x | m.y) != 0; }

RWR_DEV v3 normalize3(v3 a)
{
    const f2 len = sqrt2(dot3(a, a));
    return v3{a.x / len, a.y / len, a.z / len};
}

// element-wise normalize3_fast / normalize_fast_domain (rwr_device.h)
RWR_DEV bool normalize_fast_domain(v3 a)
{
    return normalize_fast_domain(mk3(a.x.x, a.y.x, a.z.x)) && normalize_fast_domain(mk3(a.x.y, a.y.y, a.z.y));
}
RWR_DEV f2 div_shared_rcp(f2 n, f2 d, f2 r)
{
    const f2 q0 = n * r;
    const f2 q1 = fma2(fma2(-d, q0, n), r, q0);
    return fma2(fma2(-d, q1, n), r, q1);
}
RWR_DEV f2 sqrt_fast(f2 x)
{
    const f2 y = f2{__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    const f2 g = x * y, h = 0.5f * y;
    return fma2(fma2(-g, g, x), h, g);
}
RWR_DEV v3 normalize3_fast(v3 a)
{
    const f2 len = sqrt_fast(dot3(a, a));
    f2 r = f2{__builtin_amdgcn_rcpf(len.x), __builtin_amdgcn_rcpf(len.y)};
    r = fma2(fma2(-len, r, splat(1.0f)), r, r);
    return v3{div_shared_rcp(a.x, len, r), div_shared_rcp(a.y, len, r), div_shared_rcp(a.z, len, r)};
}

// compute.wgsl:78-80
RWR_DEV f2 to_non_linear_depth(f2 depth)
{
    return ((1.0f / depth) - (1.0f / kNear)) / ((1.0f / kFar) - (1.0f / kNear));
}

RWR_DEV i2 depth_fast_domain(f2 depth)
{
    return i2{depth_fast_domain(depth.x) ? -1 : 0, depth_fast_domain(depth.y) ? -1 : 0};
}
// element-wise to_non_linear_depth_fast (same operations per element, packed)
RWR_DEV f2 to_non_linear_depth_fast(f2 depth)
{
    f2 r = f2{__builtin_amdgcn_rcpf(depth.x), __builtin_amdgcn_rcpf(depth.y)};
    r = fma2(fma2(-depth, r, splat(1.0f)), r, r);
    const f2 x = r - (1.0f / kNear);
    const f2 q = x * kDepthRC;
    return fma2(fma2(splat(-kDepthC), q, x), splat(kDepthRC), q);
}

// pixelToRay (compute.wgsl:150-164) for pixels (x0, y) and (x0 + 1, y), jitter (0.5, 0.5).
RWR_DEV v3 pixel_pair_ray_dir(const rwr_camera_inv_uniform &cam, uint32_t x0, uint32_t y, uint32_t width, uint32_t height)
{
    const f2 fx = f2{(float)x0, (float)(x0 + 1u)} + 0.5f;
    const float fy = (float)y + 0.5f;
    const f2 x_nds = 2.0f * fx / (float)width - 1.0f;
    const f2 y_nds = splat(2.0f * fy / (float)height - 1.0f);  // same row: evaluated once
    const float(&p)[4][4] = cam.proj_inv;
    // view_vec = proj_inv * (x_nds, y_nds, 1, 1); only xyz are used (w is overwritten with 0)
    const f2 vx = p[0][0] * x_nds + p[1][0] * y_nds + p[2][0] * 1.0f + p[3][0] * 1.0f;
    const f2 vy = p[0][1] * x_nds + p[1][1] * y_nds + p[2][1] * 1.0f + p[3][1] * 1.0f;
    const f2 vz = p[0][2] * x_nds + p[1][2] * y_nds + p[2][2] * 1.0f + p[3][2] * 1.0f;
    const f2 vw = splat(0.0f);
    const float(&m)[4][4] = cam.viewmodel_inv;
    v3 w;
    w.x = m[0][0] * vx + m[1][0] * vy + m[2][0] * vz + m[3][0] * vw;
    w.y = m[0][1] * vx + m[1][1] * vy + m[2][1] * vz + m[3][1] * vw;
    w.z = m[0][2] * vx + m[1][2] * vy + m[2][2] * vz + m[3][2] * vw;
    return normalize3(w);
}

// The same ray from the per-frame tables of k_frame_setup (FrameParams::ray_colp / ray_row; x0 even):
// the table entries are the products proj_inv[0] * x_nds and proj_inv[1] * y_nds, so each sum below is
// the shader's  m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]*v.w  with its first two products already rounded.
RWR_DEV v3 pixel_pair_ray_dir_tab(const rwr_camera_inv_uniform &cam, const float4 *__restrict__ colp,
                                  const float4 *__restrict__ row, uint32_t x0, uint32_t y)
{
    const float4 ca = colp[x0], cb = colp[x0 + 1u];   // entry pair of column pair x0 / 2
    const float4 r = row[y];
    const float(&p)[4][4] = cam.proj_inv;
    const f2 vx = f2{ca.x, ca.y} + r.x + p[2][0] * 1.0f + p[3][0] * 1.0f;
    const f2 vy = f2{ca.z, ca.w} + r.y + p[2][1] * 1.0f + p[3][1] * 1.0f;
    const f2 vz = f2{cb.x, cb.y} + r.z + p[2][2] * 1.0f + p[3][2] * 1.0f;
    const f2 vw = splat(0.0f);
    const float(&m)[4][4] = cam.viewmodel_inv;
    v3 w;
    w.x = m[0][0] * vx + m[1][0] * vy + m[2][0] * vz + m[3][0] * vw;
    w.y = m[0][1] * vx + m[1][1] * vy + m[2][1] * vz + m[3][1] * vw;
    w.z = m[0][2] * vx + m[1][2] * vy + m[2][2] * vz + m[3][2] * vw;
    // same bits either way (rwr_device.h normalize3_fast); a wave with an axis-parallel or absurdly
    // scaled ray takes the compiler's division
    if (__all(normalize_fast_domain(w))) return normalize3_fast(w);
    return normalize3(w);
}

// sphereRayIntersect (sphere/compute.wgsl:63-85) for a pixel pair: mask of hits and t.
RWR_DEV i2 sphere_ray_intersect_t(f3 center, float radius, f3 O, v3 D, f2 &t_out)
{
    const f3 oc1 = sub3(O, center);
    const v3 oc = splat3(oc1);
    const f2 a = dot3(D, D);
    const f2 b = 2.0f * dot3(oc, D);
    const float c = dot3(oc1, oc1) - (radius * radius);
    const f2 discriminant = b * b - 4.0f * a * c;
    const i2 miss = discriminant < 0.0f;
    if (!any2(~miss)) return i2{0, 0};
    const f2 sq = sqrt2(discriminant);
    const f2 t1 = (-b - sq) / (2.0f * a);
    const f2 t2 = (-b + sq) / (2.0f * a);
    const i2 use1 = t1 >= 0.0f, use2 = t2 >= 0.0f;
    t_out = use1 ? t1 : t2;
    return ~miss & (use1 | use2);
}

// Running nearest-hit state of the face loop for a pixel pair.
struct MeshHit2 {
    i2 have;
    f2 t, u, v, ndotd;
    u2 idx;
};

// triangleRayIntersect + selection (compute.wgsl:82-148, 198-201), branch-free, for a pixel pair
// against a wave-uniform record (cf. intersect_and_select).
// tnum = -(dot(N, O) + d), the ray-independent numerator of :99-102 (k_frame_setup).
RWR_DEV void intersect_and_select(const TriRecord &T, float tnum, uint32_t idx, f3 O, v3 D, MeshHit2 &best)
{
    const f3 N1 = ld3(T.N);
    const v3 N = splat3(N1);
    const f2 ndotd = dot3(N, D);
    i2 hit = ~(abs2(ndotd) < kEpsilon);                       // :94
    const f2 t = tnum / ndotd;                                // :99-102
    hit &= ~(t < 0.0f);                                       // :105
    const v3 P = along(splat3(O), t, D);                      // :110
    v3 C = cross3(splat3(ld3(T.e0)), sub3(P, splat3(ld3(T.p0))));
    hit &= ~(dot3(N, C) < 0.0f);                              // :118
    C = cross3(splat3(ld3(T.e1)), sub3(P, splat3(ld3(T.p1))));
    const f2 u = dot3(N, C);
    hit &= ~(u < 0.0f);                                       // :127
    C = cross3(splat3(ld3(T.e2)), sub3(P, splat3(ld3(T.p2))));
    const f2 v = dot3(N, C);
    hit &= ~(v < 0.0f);                                       // :136
    const i2 take = hit & (~best.have | (t < best.t));        // :198; ascending face order keeps the lowest index on ties
    best.have |= take;
    best.t = take ? t : best.t;
    best.u = take ? u : best.u;
    best.v = take ? v : best.v;
    best.ndotd = take ? ndotd : best.ndotd;
    best.idx = take ? u2{idx, idx} : best.idx;
}

// Scene data read at wave-uniform addresses goes through the CONSTANT address space: in a kernel that also stores to global
// memory (ray queues, atomics) the compiler no longer dares to use scalar loads for plain global pointers (it cannot see that
// nobody writes the scene) — face records would come in through the vector memory pipe, 64 identical addresses per load.
template <typename T> using const_ptr = const __attribute__((address_space(4))) T *;
template <typename T> RWR_DEV const_ptr<T> to_const_space(const T *p) { return (const_ptr<T>)(p); }

// The fields of a face record the hit test reads, loaded from the constant address space (the loads merge into four
// s_load_dwordx8 when the index is wave-uniform).
RWR_DEV TriRecord load_tri_record(const_ptr<TriRecord> rec)
{
    const const_ptr<float> f = (const_ptr<float>)rec;
    TriRecord T;
    T.p0[0] = f[0]; T.p0[1] = f[1]; T.p0[2] = f[2]; T.d = f[3];
    T.p1[0] = f[4]; T.p1[1] = f[5]; T.p1[2] = f[6]; T.denom = f[7];
    T.p2[0] = f[8]; T.p2[1] = f[9]; T.p2[2] = f[10]; T.pad0 = 0.0f;
    T.N[0] = f[12]; T.N[1] = f[13]; T.N[2] = f[14]; T.pad1 = 0.0f;
    T.e0[0] = f[16]; T.e0[1] = f[17]; T.e0[2] = f[18]; T.pad2 = 0.0f;
    T.e1[0] = f[20]; T.e1[1] = f[21]; T.e1[2] = f[22]; T.pad3 = 0.0f;
    T.e2[0] = f[24]; T.e2[1] = f[25]; T.e2[2] = f[26]; T.pad4 = 0.0f;
    T.nhat[0] = T.nhat[1] = T.nhat[2] = 0.0f; T.pad5 = 0.0f;
    return T;
}
RWR_DEV ShadeRec load_shade_record(const_ptr<ShadeRec> rec)
{
    const const_ptr<float> f = (const_ptr<float>)rec;
    ShadeRec S;
    S.n[0] = f[0]; S.n[1] = f[1]; S.n[2] = f[2]; S.ndl0 = f[3];
    S.c0[0] = f[4]; S.c0[1] = f[5]; S.c1[0] = f[6]; S.c1[1] = f[7];
    S.c2[0] = f[8]; S.c2[1] = f[9]; S.material = ((const_ptr<uint32_t>)rec)[10]; S.pad = 0.0f;
    return S;
}

}  // namespace rwr
