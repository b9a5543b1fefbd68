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

// Ray constants of the slab test:  t = b * inv - O * inv  (one FMA per plane; conservative
// code may fuse).  +-inf for axis-parallel rays is fine, see bvh_inner_step.  (nx, ny, nz): which of
// a node's six float4 planes {min x, min y, min z, max x, max y, max z} (bvh.hpp BvhNode4) is the
// NEAR one on each axis for this ray — the sign of the reciprocal picks it, so a node visit loads
// near and far planes directly instead of ordering them with six min/max per box.
struct SlabRay { float ix, iy, iz, ox, oy, oz; uint32_t nx, ny, nz; };
RWR_DEV SlabRay make_slab_ray(f3 O, f3 D)
{
    SlabRay r;
    r.ix = 1.0f / D.x; r.iy = 1.0f / D.y; r.iz = 1.0f / D.z;
    r.ox = -O.x * r.ix; r.oy = -O.y * r.iy; r.oz = -O.z * r.iz;
    r.nx = 3u * (__float_as_uint(r.ix) >> 31);        // sign bit, so that 1 / -0 = -inf counts as negative
    r.ny = 3u * (__float_as_uint(r.iy) >> 31) + 1u;
    r.nz = 3u * (__float_as_uint(r.iz) >> 31) + 2u;
    return r;
}
static_assert(offsetof(BvhNode4, bmin_y) == 16 && offsetof(BvhNode4, bmin_z) == 32 && offsetof(BvhNode4, bmax_x) == 48 &&
              offsetof(BvhNode4, bmax_y) == 64 && offsetof(BvhNode4, bmax_z) == 80, "plane order the traversal indexes by");

RWR_DEV float f4at(const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// One inner-node visit of one lane: tests the four child boxes, continues with the nearest
// survivor (returned in `cur`) and pushes the others; pops when nothing survives.  Returns
// false when the lane's traversal is finished (stack empty).
// The kernel is VALU-bound and most of a node visit is bookkeeping, so that is kept lean:
// "behind the origin" and "farther than the best hit" fold into the interval test
// max(tnear, 0) <= min(tfar, best_t); the nearest surviving child is found as the unsigned
// minimum of keys (entry distance bits with the slot in the low 2 bits) — no sorting network;
// the other survivors are pushed unordered.  (An empty slot's inverted box does NOT fail a
// min/max slab test by itself: the link is tested.)
// Stack entries: child links as they are (uint32_t), or squeezed into 16 bits (uint16_t) for scenes whose node indices
// and leaf links fit 15 bits — half the LDS, half again as many workgroups per CU for a traversal that waits on memory.
template <typename StackT> RWR_DEV StackT bvh_stack_enc(uint32_t link);
template <> RWR_DEV uint32_t bvh_stack_enc<uint32_t>(uint32_t link) { return link; }
template <> RWR_DEV uint16_t bvh_stack_enc<uint16_t>(uint32_t link) { return (uint16_t)((link & kBvhLeafBit) ? (0x8000u | (link & 0x7fffu)) : link); }
RWR_DEV uint32_t bvh_stack_dec(uint32_t e) { return e; }
RWR_DEV uint32_t bvh_stack_dec(uint16_t e) { return (e & 0x8000u) ? (kBvhLeafBit | (uint32_t)(e & 0x7fffu)) : (uint32_t)e; }

template <typename NodePtr, typename StackT>
RWR_DEV bool bvh_inner_step(NodePtr nodes, const SlabRay &sr, float tbest, uint32_t &cur, StackT *&sp, StackT *sp0,
                            uint32_t stride)
{
    uint32_t key[4], child[4];
    const auto *planes = reinterpret_cast<const float4 *>(&nodes[cur].bmin_x[0]);  // keeps NodePtr's address space
    const float4 nearx = planes[sr.nx], farx = planes[3u - sr.nx];
    const float4 neary = planes[sr.ny], fary = planes[5u - sr.ny];
    const float4 nearz = planes[sr.nz], farz = planes[7u - sr.nz];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        child[i] = nodes[cur].child[i];
        // NaN (0 * inf: origin exactly on a slab plane of an axis-parallel ray) is ignored by min/max
        const float x0 = __builtin_fmaf(f4at(nearx, i), sr.ix, sr.ox), x1 = __builtin_fmaf(f4at(farx, i), sr.ix, sr.ox);
        const float y0 = __builtin_fmaf(f4at(neary, i), sr.iy, sr.oy), y1 = __builtin_fmaf(f4at(fary, i), sr.iy, sr.oy);
        const float z0 = __builtin_fmaf(f4at(nearz, i), sr.iz, sr.oz), z1 = __builtin_fmaf(f4at(farz, i), sr.iz, sr.oz);
        const float tnear = fmaxf(fmaxf(x0, y0), z0);
        const float tfar = fminf(fminf(x1, y1), z1);
        // conservative: relative slack on both distances; equal-distance nodes are kept (<=)
        const float lo = fmaxf(tnear - 4e-5f * fabsf(tnear), 0.0f);
        const float hi = fminf(tfar + 4e-5f * fabsf(tfar) + 1e-30f, tbest);
        // lo >= 0, so its bit pattern orders like the float; low 2 bits carry the slot
        key[i] = (lo <= hi && child[i] != kBvhEmpty) ? ((__float_as_uint(lo) & ~3u) | (uint32_t)i) : 0xffffffffu;
    }
    const uint32_t kmin = min(min(key[0], key[1]), min(key[2], key[3]));
    if (kmin == 0xffffffffu) {  // nothing survived: pop
        if (sp == sp0) return false;
        sp -= stride;
        cur = bvh_stack_dec(*sp);
        return true;
    }
    const uint32_t near_slot = kmin & 3u;
    uint32_t next = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if ((uint32_t)i == near_slot) next = child[i];
        else if (key[i] != 0xffffffffu) { *sp = bvh_stack_enc<StackT>(child[i]); sp += stride; }
    }
    cur = next;
    return true;
}

// One leaf visit of one lane: exact tests of the leaf's faces, then pop.  Returns false when
// the lane's traversal is finished.
template <typename StackT>
RWR_DEV bool bvh_leaf_step(const uint32_t *__restrict__ leaf_faces, const TriRecord *__restrict__ tris, uint32_t n_faces,
                           f3 O, f3 D, MeshHit &best, uint32_t &cur, StackT *&sp, StackT *sp0, uint32_t stride)
{
    const uint32_t first = (cur & ~kBvhLeafBit) >> 3, count = (cur & 7u) + 1u;
    for (uint32_t k = 0; k < count; k++) {
        if (first + k >= n_faces) break;  // cannot happen with a well-formed tree; keeps a bad link from faulting
        const uint32_t idx = leaf_faces[first + k];
        if (idx < n_faces) intersect_and_select_any_order(tris[idx], idx, O, D, best);
    }
    if (sp == sp0) return false;
    sp -= stride;
    cur = bvh_stack_dec(*sp);
    return true;
}

// Nearest hit of one ray per lane.  nodes: LDS (or global) array of BvhNode4, root at 0.
// stack: LDS, (3 * tree depth + 2) * stride words, this lane's column at stack + lane
// (the host sizes it: a 4-wide node pushes at most 3 entries beyond the one it pops).
// "while-while" shape: every lane first walks inner nodes until it holds a leaf (or runs
// dry), then the wave tests leaves together.
template <typename NodePtr, typename StackT>
RWR_DEV void bvh_nearest(NodePtr nodes, const uint32_t *__restrict__ leaf_faces, const TriRecord *__restrict__ tris,
                         uint32_t n_faces, StackT *stack, f3 O, f3 D, MeshHit &best)
{
    const SlabRay sr = make_slab_ray(O, D);
    const uint32_t stride = blockDim.x;
    StackT *sp = stack + threadIdx.x;  // next free slot of this lane's stack
    StackT *const sp0 = sp;
    uint32_t cur = 0;  // the root is always an inner node
    bool have_cur = true;
    while (__any(have_cur)) {
        while (have_cur && !(cur & kBvhLeafBit))
            have_cur = bvh_inner_step(nodes, sr, best.have ? best.t : __builtin_inff(), cur, sp, sp0, stride);
        if (have_cur) have_cur = bvh_leaf_step(leaf_faces, tris, n_faces, O, D, best, cur, sp, sp0, stride);
    }
}

}  // namespace rwr
