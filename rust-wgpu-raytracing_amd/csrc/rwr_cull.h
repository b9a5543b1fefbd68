// Conservative face culling for rays that share one origin (primary rays).
//
// Nothing here decides what a pixel shows: these tests may only DROP faces that no
// ray of a tile can hit, and every comparison is written so that rounding, NaNs
// or degenerate inputs err on the side of keeping the face.  That is why this
// code, unlike the hit test, is free to use FMA and approximate constants.
//
// Once per frame k_frame_setup turns every face into a FrameTri: its conservative
// pixel-space bounding rectangle plus, per edge, the affine function
//     d_i(x, y) = dot(edge-plane normal_i, dir(x, y)) = ea_i + x*ex_i + y*ey_i ,
// signed so that a ray through pixel-space point (x, y) can reach the face at
// t >= 0 only if all three d_i >= 0 (rasteriser edge functions: the planes through
// the origin and an edge of the face).  The render kernel then rejects a face for
// a pixel rectangle with a handful of compares: rectangle overlap (good for small
// faces) and "whole rectangle outside one edge" (tight for faces much larger than
// the tile; also removes everything behind the camera — the reference camera
// sits INSIDE the mesh, lib.rs:352-360).
// Margins: 2e-5 relative on L1 norms (>100x the f32 rounding of the exact hit
// test) and 0.02 px + 1e-5 relative on the rectangle, on top of the half-pixel
// guard band between tile bounds and pixel centres.
#pragma once

#include "rwr_device.h"

namespace rwr {

RWR_DEV float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RWR_DEV float fdot(f3 a, f3 b) { return ffma(a.z, b.z, ffma(a.y, b.y, a.x * b.x)); }
RWR_DEV float fdot(const float *a, f3 b) { return ffma(a[2], b.z, ffma(a[1], b.y, a[0] * b.x)); }
RWR_DEV f3 fcross(f3 a, f3 b)
{
    return mk3(ffma(a.y, b.z, -(a.z * b.y)), ffma(a.z, b.x, -(a.x * b.z)), ffma(a.x, b.y, -(a.y * b.x)));
}
RWR_DEV float l1norm(f3 a) { return fabsf(a.x) + fabsf(a.y) + fabsf(a.z); }

constexpr float kCullRel = 2e-5f;  // relative margin (context.cpp folds it into CullConsts::corner_margin)
constexpr float kCullDegenerate = 1e-4f;

// One face -> FrameTri.  q_i = corner - ray origin.
RWR_DEV FrameTri make_frame_tri(const CullConsts &cc, const CullRec &R)
{
    const f3 O = ld3(cc.origin);
    const f3 q[3] = {sub3(ld3(R.p0), O), sub3(ld3(R.p1), O), sub3(ld3(R.p2), O)};
    const float inf = __builtin_inff();
    FrameTri T;

    // -- pixel-space bounding rectangle ---------------------------------------
    // A corner q = t*dir(x, y) has Vx.q = t*vxa and (Ux + x*Vx).q = 0, hence
    // t = (Vx.q)/vxa and x = -(Ux.q)/(Vx.q); likewise for y.
    float xs[3], ys[3];
    bool all_front = true, all_behind = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float vx = fdot(cc.Vx, q[i]), vy = fdot(cc.Vy, q[i]);
        const float t = vx / cc.vxa;
        const float tol = 1e-5f * l1norm(q[i]) * cc.Vx[3] / fabsf(cc.vxa);
        all_front &= t > tol;
        all_behind &= t < -tol;
        xs[i] = -fdot(cc.Ux, q[i]) / vx;
        ys[i] = -fdot(cc.Uy, q[i]) / vy;
    }
    // default: unbounded (also what NaNs fall into)
    T.bx0 = -inf; T.by0 = -inf; T.bx1 = inf; T.by1 = inf;
    if (all_front) {
        const float xmin = fminf(xs[0], fminf(xs[1], xs[2])), xmax = fmaxf(xs[0], fmaxf(xs[1], xs[2]));
        const float ymin = fminf(ys[0], fminf(ys[1], ys[2])), ymax = fmaxf(ys[0], fmaxf(ys[1], ys[2]));
        const float px = 0.02f + 1e-5f * fmaxf(fabsf(xmin), fabsf(xmax));
        const float py = 0.02f + 1e-5f * fmaxf(fabsf(ymin), fabsf(ymax));
        if (xmin <= xmax && ymin <= ymax) {  // false only with NaNs
            T.bx0 = xmin - px; T.bx1 = xmax + px; T.by0 = ymin - py; T.by1 = ymax + py;
        }
    } else if (all_behind) {
        T.bx0 = inf; T.by0 = inf; T.bx1 = -inf; T.by1 = -inf;  // empty: no t >= 0 hit possible
    }

    // -- edge functions ---------------------------------------------------------
    // D reaches the face at t > 0 iff s*dot(q_i x q_j, D) >= 0 for the three edges,
    // s = sign of the signed volume (q0 x q1).q2.
    const f3 e[3] = {fcross(q[0], q[1]), fcross(q[1], q[2]), fcross(q[2], q[0])};
    const float vol = fdot(e[0], q[2]);
    const bool reliable = fabsf(vol) > kCullDegenerate * l1norm(e[0]) * l1norm(q[2]);  // false when edge-on or NaN
    const float s = vol > 0.0f ? 1.0f : -1.0f;
    float me[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        T.ea[i] = reliable ? s * fdot(cc.A, e[i]) : inf;  // +inf: this edge never rejects
        T.ex[i] = reliable ? s * fdot(cc.Bx, e[i]) : 0.0f;
        T.ey[i] = reliable ? s * fdot(cc.By, e[i]) : 0.0f;
        me[i] = cc.corner_margin * l1norm(e[i]);
        if (!(T.ea[i] == T.ea[i]) || !(T.ex[i] == T.ex[i]) || !(T.ey[i] == T.ey[i]) || !(me[i] == me[i])) {
            T.ea[i] = inf; T.ex[i] = 0.0f; T.ey[i] = 0.0f; me[i] = 0.0f;
        }
    }
    T.me0 = me[0]; T.me1 = me[1]; T.me2 = me[2];
    return T;
}

// Pixel rectangle of a tile, in pixel units (uniform across the wave / block).
struct TileRect { float x0, y0, x1, y1; };

// True = no ray through the rectangle can reach the face at t >= 0.
RWR_DEV bool rect_culls(const FrameTri &T, const TileRect &r)
{
    bool c = (T.bx1 < r.x0) || (T.bx0 > r.x1) || (T.by1 < r.y0) || (T.by0 > r.y1);
    const float me[3] = {T.me0, T.me1, T.me2};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        // d_i is affine, so its maximum over the rectangle is at the corner its gradient points to
        // (a NaN anywhere makes the comparison false: the face is kept)
        const float xs = T.ex[i] >= 0.0f ? r.x1 : r.x0, ys = T.ey[i] >= 0.0f ? r.y1 : r.y0;
        const float dmax = ffma(xs, T.ex[i], ffma(ys, T.ey[i], T.ea[i]));
        c |= dmax < -me[i];
    }
    return c;
}

}  // namespace rwr
