// Per-lane BVH traversal for incoherent rays (the bounce stage of the wavefront
// integrator).  4-wide nodes ("nodelets", bvh.hpp) are staged in LDS by the
// workgroup; the per-lane traversal stack lives in LDS too, laid out
// [level][thread] so that a wave's pushes and pops are conflict-free.  Face
// records stay in global memory (L2-resident): a leaf test reads the 128-byte
// TriRecord of a different face in every lane.
//
// The traversal is result-preserving: boxes are padded at build time, the slab
// test is conservative (NaN-ignoring min/max, relative slack on the exit
// distance), a node is kept while its entry distance is <= the best hit so far,
// and the hit test / selection is the reference's with ties resolved by face
// index — so the winner equals that of the brute-force loop in face order
// (compute.wgsl:190-202) for ANY visiting order.
#pragma once

#include "bvh.hpp"
#include "rwr_device.h"

namespace rwr {

// triangleRayIntersect + selection with explicit lowest-index tie break (the brute-
// force loop gets it from ascending order; a BVH visits faces in any order).
RWR_DEV void intersect_and_select_any_order(const TriRecord &T, uint32_t idx, f3 O, f3 D, MeshHit &best)
{
    const f3 N = ld3(T.N);
    const float ndotd = dot3(N, D);
    bool hit = !(fabsf(ndotd) < kEpsilon);
    const float t = -(dot3(N, O) + T.d) / ndotd;
    hit &= !(t < 0.0f);
    const f3 P = along(O, t, D);
    f3 C = cross3(ld3(T.e0), sub3(P, ld3(T.p0)));
    hit &= !(dot3(N, C) < 0.0f);
    C = cross3(ld3(T.e1), sub3(P, ld3(T.p1)));
    const float u = dot3(N, C);
    hit &= !(u < 0.0f);
    C = cross3(ld3(T.e2), sub3(P, ld3(T.p2)));
    const float v = dot3(N, C);
    hit &= !(v < 0.0f);
    if (hit && (!best.have || t < best.t || (t == best.t && idx < best.idx))) {
        best.have = true;
        best.t = t;
        best.u = u;
        best.v = v;
        best.ndotd = ndotd;
        best.idx = idx;
    }
}

// nodes: LDS (or global) array of BvhNode4, root at 0.
// stack: LDS, (3 * tree depth + 2) * blockDim.x words (the host sizes it: a 4-wide
// node pushes at most 3 entries beyond the one it pops, so it cannot overflow).
template <typename NodePtr>
RWR_DEV void bvh_nearest(NodePtr nodes, const uint32_t *__restrict__ leaf_faces, const TriRecord *__restrict__ tris,
                         uint32_t *stack, f3 O, f3 D, MeshHit &best)
{
    const float ix = 1.0f / D.x, iy = 1.0f / D.y, iz = 1.0f / D.z;  // +-inf for axis-parallel rays is fine below
    const uint32_t tid = threadIdx.x, stride = blockDim.x;
    uint32_t sp = 0;
    uint32_t cur = 0;  // the root is always an inner node
    bool have_cur = true;
    while (have_cur) {
        if (cur & kBvhLeafBit) {
            const uint32_t first = (cur & ~kBvhLeafBit) >> 3, count = (cur & 7u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                const uint32_t idx = leaf_faces[first + k];
                intersect_and_select_any_order(tris[idx], idx, O, D, best);
            }
        } else {
            float tn[4];
            uint32_t ch[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float x0 = (nodes[cur].bmin_x[i] - O.x) * ix, x1 = (nodes[cur].bmax_x[i] - O.x) * ix;
                const float y0 = (nodes[cur].bmin_y[i] - O.y) * iy, y1 = (nodes[cur].bmax_y[i] - O.y) * iy;
                const float z0 = (nodes[cur].bmin_z[i] - O.z) * iz, z1 = (nodes[cur].bmax_z[i] - O.z) * iz;
                const float tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                const float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                const uint32_t c = nodes[cur].child[i];
                // conservative: relative + absolute slack on the exit distance; keep equal-distance nodes (<=)
                const float tf = tfar + 2e-5f * fabsf(tfar) + 1e-30f;
                bool ok = (c != kBvhEmpty) && (tnear <= tf) && (tf >= 0.0f);
                if (best.have) ok = ok && (tnear <= best.t);
                tn[i] = ok ? tnear : __builtin_inff();
                ch[i] = ok ? c : kBvhEmpty;
            }
            // sort the four by entry distance (5 compare-exchanges), nearest first
            auto cswap = [&](int a, int b) {
                if (tn[b] < tn[a]) {
                    const float tf = tn[a]; tn[a] = tn[b]; tn[b] = tf;
                    const uint32_t tc = ch[a]; ch[a] = ch[b]; ch[b] = tc;
                }
            };
            cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
            // push far ones first so the nearest is popped next
#pragma unroll
            for (int i = 3; i >= 0; i--)
                if (ch[i] != kBvhEmpty) stack[(sp++) * stride + tid] = ch[i];
        }
        have_cur = sp > 0;
        if (have_cur) cur = stack[(--sp) * stride + tid];
    }
}

}  // namespace rwr
