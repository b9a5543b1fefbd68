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
// Same operations as intersect_and_select, but staged: after each of the shader's
// early returns the wave checks whether ANY of its active lanes is still alive and
// leaves if not — in a BVH leaf most rays miss the face, so the three edge tests
// (2/3 of the arithmetic) are usually skipped for the whole wave.  Lanes that stay
// compute exactly what the branch-free form computes.
RWR_DEV void intersect_and_select_any_order(const TriRecord &T, uint32_t idx, f3 O, f3 D, MeshHit &best)
{
    const f3 N = ld3(T.N);
    const float ndotd = dot3(N, D);
    bool hit = !(fabsf(ndotd) < kEpsilon);              // :94
    const float t = -(dot3(N, O) + T.d) / ndotd;        // :99-102
    hit &= !(t < 0.0f);                                 // :105
    // a hit farther than the current best can never be selected (ties need t == best.t)
    hit &= !(best.have && t > best.t);
    if (!__any(hit)) return;
    const f3 P = along(O, t, D);                        // :110
    f3 C = cross3(ld3(T.e0), sub3(P, ld3(T.p0)));       // :115-117
    hit &= !(dot3(N, C) < 0.0f);                        // :118
    if (!__any(hit)) return;
    C = cross3(ld3(T.e1), sub3(P, ld3(T.p1)));          // :123-125
    const float u = dot3(N, C);
    hit &= !(u < 0.0f);                                 // :127
    if (!__any(hit)) return;
    C = cross3(ld3(T.e2), sub3(P, ld3(T.p2)));          // :132-134
    const float v = dot3(N, C);
    hit &= !(v < 0.0f);                                 // :136
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
// "while-while" shape: every lane first walks inner nodes until it holds a leaf (or
// runs dry), then the wave tests leaves together — inner-node and leaf work are not
// serialised against each other inside one iteration.
template <typename NodePtr>
RWR_DEV void bvh_nearest(NodePtr nodes, const uint32_t *__restrict__ leaf_faces, const TriRecord *__restrict__ tris,
                         uint32_t *stack, f3 O, f3 D, MeshHit &best)
{
    // slab planes through  t = b * inv - O * inv  (one FMA each; conservative code may fuse)
    const float ix = 1.0f / D.x, iy = 1.0f / D.y, iz = 1.0f / D.z;  // +-inf for axis-parallel rays is fine below
    const float ox = -O.x * ix, oy = -O.y * iy, oz = -O.z * iz;
    const uint32_t tid = threadIdx.x, stride = blockDim.x;
    uint32_t sp = 0;
    uint32_t cur = 0;  // the root is always an inner node
    bool have_cur = true;
    while (__any(have_cur)) {
        // phase 1: inner nodes
        while (have_cur && !(cur & kBvhLeafBit)) {
            float tn[4];
            uint32_t ch[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // NaN (0 * inf: origin exactly on a slab plane of an axis-parallel ray) is ignored by min/max
                const float x0 = __builtin_fmaf(nodes[cur].bmin_x[i], ix, ox), x1 = __builtin_fmaf(nodes[cur].bmax_x[i], ix, ox);
                const float y0 = __builtin_fmaf(nodes[cur].bmin_y[i], iy, oy), y1 = __builtin_fmaf(nodes[cur].bmax_y[i], iy, oy);
                const float z0 = __builtin_fmaf(nodes[cur].bmin_z[i], iz, oz), z1 = __builtin_fmaf(nodes[cur].bmax_z[i], iz, oz);
                const float tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                const float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                const uint32_t c = nodes[cur].child[i];
                // conservative: relative + absolute slack on both distances; keep equal-distance nodes (<=)
                const float slack = 4e-5f * fmaxf(fabsf(tnear), fabsf(tfar)) + 1e-30f;
                bool ok = (c != kBvhEmpty) && (tnear - slack <= tfar + slack) && (tfar + slack >= 0.0f);
                if (best.have) ok = ok && (tnear - slack <= best.t);
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
            // push the far ones, continue with the nearest
#pragma unroll
            for (int i = 3; i >= 1; i--)
                if (ch[i] != kBvhEmpty) stack[(sp++) * stride + tid] = ch[i];
            if (ch[0] != kBvhEmpty) {
                cur = ch[0];
            } else {
                have_cur = sp > 0;
                if (have_cur) cur = stack[(--sp) * stride + tid];
            }
        }
        // phase 2: the leaf this lane arrived at
        if (have_cur) {
            const uint32_t first = (cur & ~kBvhLeafBit) >> 3, count = (cur & 7u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                const uint32_t idx = leaf_faces[first + k];
                intersect_and_select_any_order(tris[idx], idx, O, D, best);
            }
            have_cur = sp > 0;
            if (have_cur) cur = stack[(--sp) * stride + tid];
        }
    }
}

}  // namespace rwr
