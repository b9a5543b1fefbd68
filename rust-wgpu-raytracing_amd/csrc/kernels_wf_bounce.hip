// Wavefront integrator, second stage: the bounce rays of one launch group (kernels_wf_primary.hip).
//
//   k_wf_bounce    one workgroup per TILE of the primary stage (64x8 pixels); its ray POOL is everything those
//                  pixels emitted in the group's samples (up to 32 x 512 rays, origins within one small patch of
//                  surface, directions all over the hemisphere).
//                  1. compaction + sort, in one: the ballots the primary stage published say which fixed queue
//                     slots hold a ray; every ray's direction is binned (octant x 8x8 cells of the octahedral map,
//                     Morton order inside an octant), a histogram / prefix sum / scatter through LDS atomics
//                     leaves the pool's live slots in direction order (wf.sorted, 2 B per ray).
//                     Rays of one wave then start from almost the same point in almost the same direction:
//                     they walk the BVH together (measured on the simulator, tools/sim: 62 -> 36 wave
//                     instructions per ray at cfg3 — neighbouring queue entries share an origin, not a direction).
//                  2. traversal: per-lane, 4-wide BVH with nodelets staged in LDS (rwr_bvh.h), the reference's
//                     exact hit test with ties broken by face index — the winner equals the brute-force
//                     winner (oracle) for any visiting order.
//                  3. shading of the second hit; the contribution albedo(h0) * E(h1) is added to the pool's
//                     per-pixel sums in LDS as 64-bit FIXED-POINT atomics: integer sums do not depend on the
//                     order in which rays retire, so the frame is bit-reproducible whatever the sort did.
//                  4. one read-modify-write of the RGBA32F accumulator per pixel and group.
//
// Nothing here decides what the first hit shows; bounce rays are the oracle's rays bit for bit (first stage) and
// their nearest hits are exact, so the stage's output differs from the oracle's only by float summation order
// (tolerance 1e-4).
#include "rwr_bvh.h"
#include "rwr_device_p2.h"
#include "rwr_primary.h"

namespace rwr {

constexpr float kWfFixedScale = 67108864.0f;  // 2^26: contributions < 64 per ray, 32 samples of them < 2^37

// Direction bin: 3 bits of octant (Gray-coded so that neighbours share two signs) and 6 bits of position inside
// the octant's triangle of the octahedral map (Morton order of an 8x8 grid).
RWR_DEV uint32_t direction_bin(f3 D)
{
    const uint32_t sx = __float_as_uint(D.x) >> 31, sy = __float_as_uint(D.y) >> 31, sz = __float_as_uint(D.z) >> 31;
    const uint32_t oct = sz * 4u + (sy ^ sz) * 2u + (sx ^ sy);  // reflected Gray code of (sz, sy, sx)
    const float ax = fabsf(D.x), ay = fabsf(D.y), az = fabsf(D.z);
    const float inv = __builtin_amdgcn_rcpf(ax + ay + az + 1e-30f);
    const uint32_t iu = min(7u, (uint32_t)(ax * inv * 8.0f)), iv = min(7u, (uint32_t)(ay * inv * 8.0f));
    const uint32_t mu = (iu & 1u) | ((iu & 2u) << 1) | ((iu & 4u) << 2), mv = (iv & 1u) | ((iv & 2u) << 1) | ((iv & 4u) << 2);
    return oct * 64u + (mu | (mv << 1));
}

struct BouncePoolShared {
    unsigned long long masks[kWfMaxGroup * 8u];
    unsigned long long acc[kWfTilePixels * 3u];   // fixed-point sums of albedo * E(h1) per pixel of the tile
    uint32_t hist[kWfDirBins];
    uint32_t offs[kWfDirBins];
    uint32_t oct_begin[9];   // sorted position where each octant's rays begin; [8] = ray count
    uint32_t total;
    uint32_t next_packet;
};

// Steps 0 and 1 for one pool: the published ballots -> live slot count; direction sort -> wf.sorted[pool] holds
// the live slots octant by octant (octant o: sh.oct_begin[o] .. sh.oct_begin[o + 1]), fine bins in Morton order
// inside.  Returns the number of rays (uniform over the workgroup).  Contains barriers.
RWR_DEV uint32_t prepare_pool(BouncePoolShared &sh, const WfBuffers &wf, uint32_t tile, uint32_t sample_count, uint32_t min_rays,
                              uint32_t max_rays)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t n_masks = sample_count * 8u, n_slots = sample_count * kWfTilePixels;
    if (tid == 0u) sh.total = 0u;
    __syncthreads();
    if (tid < n_masks) {
        const unsigned long long m = wf.masks[(size_t)tile * wf.group * 8u + tid];
        sh.masks[tid] = m;
        if (m) atomicAdd(&sh.total, (uint32_t)__popcll(m));
    }
    for (uint32_t i = tid; i < kWfDirBins; i += 256u) sh.hist[i] = 0u;
    for (uint32_t i = tid; i < kWfTilePixels * 3u; i += 256u) sh.acc[i] = 0ull;
    __syncthreads();
    const uint32_t n_rays = sh.total;
    if (n_rays < min_rays || n_rays >= max_rays) return 0u;  // nothing here, or the other kernel's pool (uniform)

    const size_t pool_base = (size_t)tile * wf.group * kWfTilePixels;
    for (uint32_t e = tid; e < n_slots; e += 256u) {  // e = sample * 512 + wave * 128 + k * 64 + lane
        if ((sh.masks[e >> 6] >> (e & 63u)) & 1ull) {
            const float4 d = wf.q1[pool_base + e];
            atomicAdd(&sh.hist[direction_bin(mk3(d.x, d.y, d.z))], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = tid; b < kWfDirBins; b += 256u) {  // exclusive prefix (512 bins: two per thread, broadcast reads)
        uint32_t sum = 0;
        for (uint32_t j = 0; j < b; j++) sum += sh.hist[j];
        sh.offs[b] = sum;
        if ((b & 63u) == 0u) sh.oct_begin[b >> 6] = sum;
    }
    if (tid == 0u) sh.oct_begin[8] = n_rays;
    __syncthreads();
    uint16_t *__restrict__ sorted = wf.sorted + pool_base;
    for (uint32_t e = tid; e < n_slots; e += 256u) {
        if ((sh.masks[e >> 6] >> (e & 63u)) & 1ull) {
            const float4 d = wf.q1[pool_base + e];
            sorted[atomicAdd(&sh.offs[direction_bin(mk3(d.x, d.y, d.z))], 1u)] = (uint16_t)e;
        }
    }
    __threadfence_block();
    __syncthreads();
    return n_rays;
}

// Adds one ray's contribution albedo(h0) * E(h1) to its pixel's fixed-point sums.  e: the ray's pool slot.
RWR_DEV void add_contribution(BouncePoolShared &sh, uint32_t e, float cr, float cg, float cb)
{
    const uint32_t r = e & (kWfTilePixels - 1u), w = r >> 7, k = (r >> 6) & 1u, l = r & 63u;
    const uint32_t lx = (w & 1u) * 32u + 2u * (l & 15u) + k, ly = (w >> 1) * 4u + (l >> 4);
    unsigned long long *dst = &sh.acc[(ly * kWfTileW + lx) * 3u];
    // float -> u32 conversion saturates and sends NaN / negatives to 0
    atomicAdd(dst + 0, (unsigned long long)(uint32_t)(cr * kWfFixedScale));
    atomicAdd(dst + 1, (unsigned long long)(uint32_t)(cg * kWfFixedScale));
    atomicAdd(dst + 2, (unsigned long long)(uint32_t)(cb * kWfFixedScale));
}

// Step 4: the pool's sums -> one read-modify-write of the RGBA32F accumulator per pixel.  Call after a barrier.
RWR_DEV void flush_pool(BouncePoolShared &sh, const FrameParams &p, const WfBuffers &wf, uint32_t tile)
{
    const uint32_t tile_x0 = (tile % wf.tiles_x) * kWfTileW, tile_y0 = p.row_begin + (tile / wf.tiles_x) * kWfTileH;
    for (uint32_t q = threadIdx.x; q < kWfTilePixels; q += 256u) {
        const uint32_t px = tile_x0 + (q & (kWfTileW - 1u)), py = tile_y0 + q / kWfTileW;
        const unsigned long long sr = sh.acc[q * 3u], sg = sh.acc[q * 3u + 1u], sb = sh.acc[q * 3u + 2u];
        if ((sr | sg | sb) != 0ull && px < p.width && py < p.row_end) {
            const uint32_t pixel = py * p.width + px;
            float4 acc = wf.accum[pixel];
            acc.x += (float)sr * (1.0f / kWfFixedScale);
            acc.y += (float)sg * (1.0f / kWfFixedScale);
            acc.z += (float)sb * (1.0f / kWfFixedScale);
            wf.accum[pixel] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Sparse pools (fewer than `max_rays` rays: silhouette tiles, small instances): one ray per lane, per-lane BVH
// traversal with the nodelets and the traversal stacks in LDS (rwr_bvh.h).
template <bool NODES_IN_LDS>
__global__ void __launch_bounds__(256)
k_wf_bounce(const FrameParams p, const TriRecord *__restrict__ tris, const ShadeRec *__restrict__ shade,
            const BvhDevice bvh, const float4 *__restrict__ tex, const WfBuffers wf, uint32_t sample_count, uint32_t max_rays)
{
    __shared__ BouncePoolShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t n_rays = prepare_pool(sh, wf, tile, sample_count, 1u, max_rays);
    if (n_rays == 0u) return;

    // LDS carve of the dynamic part: [nodelets][traversal stack]
    BvhNode4 *s_nodes = reinterpret_cast<BvhNode4 *>(s_dyn);
    const uint32_t node_bytes = NODES_IN_LDS ? bvh.n_nodes * (uint32_t)sizeof(BvhNode4) : 0u;
    uint32_t *s_stack = reinterpret_cast<uint32_t *>(s_dyn + node_bytes);
    if (NODES_IN_LDS) {
        const float4 *src = reinterpret_cast<const float4 *>(bvh.nodes);
        float4 *dst = reinterpret_cast<float4 *>(s_nodes);
        for (uint32_t i = tid; i < bvh.n_nodes * 8u; i += 256u) dst[i] = src[i];
        __syncthreads();
    }
    const size_t pool_base = (size_t)tile * wf.group * kWfTilePixels;
    const uint16_t *__restrict__ sorted = wf.sorted + pool_base;
    for (uint32_t i = tid; i < n_rays; i += 256u) {
        const uint32_t e = sorted[i];
        const float4 a = wf.q0[pool_base + e], b = wf.q1[pool_base + e];
        const float c = wf.q2[pool_base + e];
        const f3 O = mk3(a.x, a.y, a.z), D = mk3(b.x, b.y, b.z);
        const f3 thr = mk3(a.w, b.w, c);

        // nearest over spheres (in order), then the mesh; strict '<' keeps the earlier candidate on ties
        bool have = false;
        float best_t = 0.0f;
        int32_t obj = -1;
        for (uint32_t s = 0; s < p.n_spheres; s++) {
            float t;
            if (sphere_ray_intersect_t(ld3(p.spheres[s].center), p.spheres[s].radius, O, D, t)) {
                if (!have || t < best_t) { have = true; best_t = t; obj = -2 - (int32_t)s; }
            }
        }
        MeshHit mh;
        mh.have = false; mh.t = 0.0f; mh.u = 0.0f; mh.v = 0.0f; mh.ndotd = 0.0f; mh.idx = 0u;
        if (p.n_tris) {
            if (NODES_IN_LDS) bvh_nearest(s_nodes, bvh.leaf_faces, tris, p.n_tris, s_stack, O, D, mh);
            else bvh_nearest(bvh.nodes, bvh.leaf_faces, tris, p.n_tris, s_stack, O, D, mh);
            if (mh.have && (!have || mh.t < best_t)) { have = true; best_t = mh.t; obj = (int32_t)mh.idx; }
        }
        if (have) {
            const f3 e1 = shade_winner(p, obj, best_t, mh.u, mh.v, mh.ndotd, shade, tex, O, D).colour;
            add_contribution(sh, e, thr.x * e1.x, thr.y * e1.y, thr.z * e1.z);
        }
    }
    __syncthreads();
    flush_pool(sh, p, wf, tile);
}

// ---------------------------------------------------------------------------------------------------------------
// Dense pools: PACKET traversal.  A wave takes 128 consecutive rays of one octant of the sorted pool, two per
// lane (v_pk_* arithmetic), and walks the BVH ONCE for all of them: the node index is wave-uniform, so the
// nodelet's planes and child links arrive by scalar loads and sit in SGPRs, the one traversal stack is a VGPR
// addressed by lane (entry i in lane i, popped with v_readlane), children are ordered by the entry distance of the first ray that
// hits them, and a leaf's faces are tested like the frame kernel tests them — record in scalar registers, all rays
// in lock-step, branch-free.  No LDS in the loop, no per-lane gathers.  A child is entered when ANY ray of the
// packet can reach it; rays the box test excluded still run the exact test (it can only find true hits).
// Same conservative slab test, same exact hit test and (t, face index) selection as the per-lane traversal, so the
// winner of every ray is the brute-force winner.
struct PairRays {
    v3 O, D;             // the two rays of a lane
    f2 ix, iy, iz, ox, oy, oz;   // slab constants (rwr_bvh.h make_slab_ray)
    i2 valid;
};

RWR_DEV f2 max2(f2 a, f2 b) { return f2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
RWR_DEV f2 min2(f2 a, f2 b) { return f2{fminf(a.x, b.x), fminf(a.y, b.y)}; }

// triangleRayIntersect + selection with the explicit lowest-index tie break, for a ray pair with separate origins
// against a wave-uniform record (cf. rwr_bvh.h intersect_and_select_any_order, rwr_device_p2.h intersect_and_select).
RWR_DEV void intersect_pair_any_order(const TriRecord &T, uint32_t idx, const PairRays &R, MeshHit2 &best)
{
    const v3 N = splat3(ld3(T.N));
    const f2 ndotd = dot3(N, R.D);
    i2 hit = R.valid & ~(abs2(ndotd) < kEpsilon);             // :94
    const f2 t = -(dot3(N, R.O) + T.d) / ndotd;               // :99-102
    hit &= ~(t < 0.0f);                                       // :105
    // a hit farther than the ray's best can never be selected (ties need t == best.t); when no ray of the packet is
    // left after a stage, the rest of the test is skipped for the whole wave (lanes that stay compute what the
    // branch-free form computes)
    hit &= ~(best.have & (t > best.t));
    if (!__any(any2(hit))) return;
    const v3 P = along(R.O, t, R.D);                          // :110
    v3 C = cross3(splat3(ld3(T.e0)), sub3(P, splat3(ld3(T.p0))));
    hit &= ~(dot3(N, C) < 0.0f);                              // :118
    if (!__any(any2(hit))) return;
    C = cross3(splat3(ld3(T.e1)), sub3(P, splat3(ld3(T.p1))));
    const f2 u = dot3(N, C);
    hit &= ~(u < 0.0f);                                       // :127
    if (!__any(any2(hit))) return;
    C = cross3(splat3(ld3(T.e2)), sub3(P, splat3(ld3(T.p2))));
    const f2 v = dot3(N, C);
    hit &= ~(v < 0.0f);                                       // :136
    const i2 lower = i2{idx < best.idx.x ? -1 : 0, idx < best.idx.y ? -1 : 0};
    const i2 take = hit & (~best.have | (t < best.t) | ((t == best.t) & lower));
    best.have |= take;
    best.t = take ? t : best.t;
    best.u = take ? u : best.u;
    best.v = take ? v : best.v;
    best.ndotd = take ? ndotd : best.ndotd;
    best.idx = take ? u2{idx, idx} : best.idx;
}

// sphereRayIntersect (sphere/compute.wgsl:63-85) for a ray pair with separate origins.
RWR_DEV i2 sphere_pair_intersect_t(f3 center, float radius, v3 O, v3 D, f2 &t_out)
{
    const v3 oc = sub3(O, splat3(center));
    const f2 a = dot3(D, D);
    const f2 b = 2.0f * dot3(oc, D);
    const f2 c = dot3(oc, oc) - (radius * radius);
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

template <int DUMMY>
__global__ void __launch_bounds__(256)
k_wf_bounce_packet(const FrameParams p, const TriRecord *__restrict__ tris, const ShadeRec *__restrict__ shade,
                   const BvhDevice bvh, const float4 *__restrict__ tex, const WfBuffers wf, uint32_t sample_count, uint32_t min_rays)
{
    __shared__ BouncePoolShared sh;
    const uint32_t tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t n_rays = prepare_pool(sh, wf, tile, sample_count, max(min_rays, 1u), 0xffffffffu);
    if (n_rays == 0u) return;
    if (tid == 0u) sh.next_packet = 0u;
    __syncthreads();
    // packets never straddle an octant: packet q of octant o covers sorted[oct_begin[o] + 128 q ...)
    uint32_t pk_begin[9];
    pk_begin[0] = 0u;
#pragma unroll
    for (int o = 0; o < 8; o++) pk_begin[o + 1] = pk_begin[o] + (sh.oct_begin[o + 1] - sh.oct_begin[o] + 127u) / 128u;
    const uint32_t n_packets = pk_begin[8];
    const size_t pool_base = (size_t)tile * wf.group * kWfTilePixels;
    const uint16_t *__restrict__ sorted = wf.sorted + pool_base;
    const BvhNode4 *__restrict__ nodes = bvh.nodes;

    for (;;) {
        uint32_t pk = 0;
        if (lane == 0u) pk = atomicAdd(&sh.next_packet, 1u);
        pk = (uint32_t)__builtin_amdgcn_readfirstlane((int)pk);
        if (pk >= n_packets) break;
        uint32_t oct = 0;
#pragma unroll
        for (int o = 1; o < 8; o++) oct += (pk >= pk_begin[o]) ? 1u : 0u;
        const uint32_t first = sh.oct_begin[oct] + (pk - pk_begin[oct]) * 128u, last = sh.oct_begin[oct + 1];
        // Gray-coded octant -> sign bits (kernels: direction_bin): sz = bit 2, sy = bit 1 ^ sz, sx = bit 0 ^ sy
        const uint32_t sz = oct >> 2, sy = ((oct >> 1) & 1u) ^ sz, sx = (oct & 1u) ^ sy;

        // -- the lane's two rays ---------------------------------------------------------------------------------
        const uint32_t i0 = first + lane, i1 = first + 64u + lane;
        PairRays R;
        R.valid = i2{i0 < last ? -1 : 0, i1 < last ? -1 : 0};
        const uint32_t e0 = R.valid.x ? sorted[i0] : sorted[first], e1 = R.valid.y ? sorted[i1] : sorted[first];
        const float4 a0 = wf.q0[pool_base + e0], b0 = wf.q1[pool_base + e0], a1 = wf.q0[pool_base + e1], b1 = wf.q1[pool_base + e1];
        const float c0 = wf.q2[pool_base + e0], c1 = wf.q2[pool_base + e1];
        R.O = v3{f2{a0.x, a1.x}, f2{a0.y, a1.y}, f2{a0.z, a1.z}};
        R.D = v3{f2{b0.x, b1.x}, f2{b0.y, b1.y}, f2{b0.z, b1.z}};
        const v3 thr = v3{f2{a0.w, a1.w}, f2{b0.w, b1.w}, f2{c0, c1}};
        // slab constants by v_rcp_f32 (1 ulp): the box test is conservative by 4e-5 relative on either side
        R.ix = f2{__builtin_amdgcn_rcpf(R.D.x.x), __builtin_amdgcn_rcpf(R.D.x.y)};
        R.iy = f2{__builtin_amdgcn_rcpf(R.D.y.x), __builtin_amdgcn_rcpf(R.D.y.y)};
        R.iz = f2{__builtin_amdgcn_rcpf(R.D.z.x), __builtin_amdgcn_rcpf(R.D.z.y)};
        R.ox = -R.O.x * R.ix; R.oy = -R.O.y * R.iy; R.oz = -R.O.z * R.iz;

        // -- nearest mesh hit: packet traversal ----------------------------------------------------------------------
        MeshHit2 best;
        best.have = i2{0, 0};
        best.t = best.u = best.v = best.ndotd = splat(0.0f);
        best.idx = u2{0u, 0u};
        if (p.n_tris) {
            uint32_t stk = 0;   // the packet's traversal stack: lane i holds entry i
            uint32_t sp = 0;    // wave-uniform
            uint32_t cur = 0;   // the root is always an inner node
            for (;;) {
                if (!(cur & kBvhLeafBit)) {
                    const BvhNode4 &nd = nodes[cur];   // wave-uniform index: scalar loads
                    const float *nearx = sx ? nd.bmax_x : nd.bmin_x, *farx = sx ? nd.bmin_x : nd.bmax_x;
                    const float *neary = sy ? nd.bmax_y : nd.bmin_y, *fary = sy ? nd.bmin_y : nd.bmax_y;
                    const float *nearz = sz ? nd.bmax_z : nd.bmin_z, *farz = sz ? nd.bmin_z : nd.bmax_z;
                    // rays that are out of the running: invalid ones, and no distance beyond a ray's best hit matters
                    const f2 tb = f2{R.valid.x ? (best.have.x ? best.t.x : __builtin_inff()) : -1.0f,
                                     R.valid.y ? (best.have.y ? best.t.y : __builtin_inff()) : -1.0f};
                    // per child slot: entry-distance key (0xffffffff: nobody reaches it) and link — wave-uniform values
                    uint32_t k0 = 0xffffffffu, k1 = 0xffffffffu, k2 = 0xffffffffu, k3 = 0xffffffffu;
                    uint32_t c0 = nd.child[0], c1 = nd.child[1], c2 = nd.child[2], c3 = nd.child[3];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t child = i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : c3;
                        if (child == kBvhEmpty) continue;   // uniform
                        const f2 x0 = fma2(splat(nearx[i]), R.ix, R.ox), x1 = fma2(splat(farx[i]), R.ix, R.ox);
                        const f2 y0 = fma2(splat(neary[i]), R.iy, R.oy), y1 = fma2(splat(fary[i]), R.iy, R.oy);
                        const f2 z0 = fma2(splat(nearz[i]), R.iz, R.oz), z1 = fma2(splat(farz[i]), R.iz, R.oz);
                        const f2 tnear = max2(max2(x0, y0), z0), tfar = min2(min2(x1, y1), z1);
                        // conservative, exactly as rwr_bvh.h bvh_inner_step: relative slack on both distances, <=
                        const f2 lo = max2(tnear - 4e-5f * abs2(tnear), splat(0.0f));
                        const f2 hi = min2(tfar + 4e-5f * abs2(tfar) + 1e-30f, tb);
                        const i2 in = lo <= hi;
                        const unsigned long long m = __ballot(any2(in));
                        if (m) {   // uniform: some ray of the packet can reach this child
                            // order key: entry distance of the first lane that reaches it (lo >= 0: bits order like floats)
                            const float lo1 = in.x ? lo.x : lo.y;
                            const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(lo1), (int)__builtin_ctzll(m)) & ~3u;
                            if (i == 0) k0 = kk; else if (i == 1) k1 = kk | 1u; else if (i == 2) k2 = kk | 2u; else k3 = kk | 3u;
                        }
                    }
                    // sort the four (key, link) pairs ascending by key: a network of five compare-exchanges on scalars
#define RWR_CE(ka, ca, kb, cb) { const bool sw = kb < ka; const uint32_t tk = sw ? kb : ka, tc = sw ? cb : ca; kb = sw ? ka : kb; cb = sw ? ca : cb; ka = tk; ca = tc; }
                    RWR_CE(k0, c0, k1, c1) RWR_CE(k2, c2, k3, c3) RWR_CE(k0, c0, k2, c2) RWR_CE(k1, c1, k3, c3) RWR_CE(k1, c1, k2, c2)
#undef RWR_CE
                    if (k0 == 0xffffffffu) {   // nobody reaches any child: pop
                        if (sp == 0u) break;
                        sp--;
                        cur = (uint32_t)__builtin_amdgcn_readlane((int)stk, (int)sp);
                        continue;
                    }
                    // nearest child next; the others go on the stack, farthest first (entry sp lives in lane sp)
                    if (k3 != 0xffffffffu) { stk = (lane == sp) ? c3 : stk; sp++; }
                    if (k2 != 0xffffffffu) { stk = (lane == sp) ? c2 : stk; sp++; }
                    if (k1 != 0xffffffffu) { stk = (lane == sp) ? c1 : stk; sp++; }
                    cur = c0;
                } else {
                    const uint32_t lf = (cur & ~kBvhLeafBit) >> 3, count = (cur & 7u) + 1u;
                    for (uint32_t k = 0; k < count; k++) {
                        const uint32_t idx = bvh.leaf_faces[lf + k];   // uniform: scalar load
                        if (idx < p.n_tris) intersect_pair_any_order(tris[idx], idx, R, best);
                    }
                    if (sp == 0u) break;
                    sp--;
                    cur = (uint32_t)__builtin_amdgcn_readlane((int)stk, (int)sp);
                }
            }
        }

        // -- spheres (in order), then the mesh; strict '<' keeps the earlier candidate on ties -----------------------
        i2 have = i2{0, 0}, obj = i2{-1, -1};
        f2 best_t = splat(0.0f);
        for (uint32_t s = 0; s < p.n_spheres; s++) {
            f2 t = splat(0.0f);
            const i2 hit = sphere_pair_intersect_t(ld3(p.spheres[s].center), p.spheres[s].radius, R.O, R.D, t);
            const i2 take = hit & R.valid & (~have | (t < best_t));
            have |= take;
            best_t = take ? t : best_t;
            obj = take ? i2{-2 - (int)s, -2 - (int)s} : obj;
        }
        {
            const i2 take = best.have & (~have | (best.t < best_t));
            have |= take;
            best_t = take ? best.t : best_t;
            obj = take ? i2{(int)best.idx.x, (int)best.idx.y} : obj;
        }

        // -- shade the second hit, add albedo(h0) * E(h1) to the pixel's sums --------------------------------------
        if (__any(any2(have))) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (k ? have.y : have.x) {
                    const f3 Ok = lane3(R.O, k), Dk = lane3(R.D, k);
                    const f3 ek = shade_winner(p, k ? obj.y : obj.x, k ? best_t.y : best_t.x, k ? best.u.y : best.u.x,
                                               k ? best.v.y : best.v.x, k ? best.ndotd.y : best.ndotd.x, shade, tex, Ok, Dk).colour;
                    add_contribution(sh, k ? e1 : e0, (k ? thr.x.y : thr.x.x) * ek.x, (k ? thr.y.y : thr.y.x) * ek.y,
                                     (k ? thr.z.y : thr.z.x) * ek.z);
                }
            }
        }
    }
    __syncthreads();
    flush_pool(sh, p, wf, tile);
}

hipError_t launch_wf_bounce(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                            const BvhDevice &bvh, const float4 *tex, const WfBuffers &wf, uint32_t n_tiles,
                            uint32_t sample_count, uint32_t packet_min_rays)
{
    if (n_tiles == 0 || sample_count == 0) return hipSuccess;
    const dim3 grid(n_tiles);
    // pools of at least packet_min_rays rays: packet traversal (its one stack is a VGPR of 64 entries)
    const bool packets = bvh.stack_depth <= 64u && packet_min_rays <= sample_count * kWfTilePixels;
    const uint32_t split = packets ? packet_min_rays : 0xffffffffu;
    if (packets)
        hipLaunchKernelGGL((k_wf_bounce_packet<0>), grid, dim3(256), 0, s, fp, tris, shade, bvh, tex, wf, sample_count, split);
    if (split > 1u) {
        // the rest: per-lane traversal
        const size_t fixed = (size_t)bvh.stack_depth * 256u * 4u;
        const size_t node_bytes = (size_t)bvh.n_nodes * sizeof(BvhNode4);
        // nodelets go to LDS when the workgroup then still fits a CU at least three times (160 KiB LDS, ~19 KiB static)
        if (node_bytes + fixed <= 28u * 1024u)
            hipLaunchKernelGGL((k_wf_bounce<true>), grid, dim3(256), node_bytes + fixed, s, fp, tris, shade, bvh, tex, wf, sample_count, split);
        else
            hipLaunchKernelGGL((k_wf_bounce<false>), grid, dim3(256), fixed, s, fp, tris, shade, bvh, tex, wf, sample_count, split);
    }
    return hipGetLastError();
}

hipError_t preload_kernels_wf_bounce()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>((&k_wf_bounce<true>)));
}

}  // namespace rwr
