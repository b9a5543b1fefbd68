// Wavefront integrator, second stage: the bounce rays of one launch group (kernels_wf_primary.hip).
//
// The unit of work is a POOL: everything the 64x8 pixels of one tile of the primary stage emitted in the group's
// samples (up to 32 x 512 rays: origins within one small patch of surface, directions all over the hemisphere).
// Three kernels per launch group:
//   k_wf_sort          (one workgroup per pool) steps 1 below + how the pool is traced;
//   k_wf_trace_packet  well-filled compact pools: 128 rays per wave walk the BVH together (see there);
//   k_wf_trace_lane    the other pools: one ray per lane, rwr_bvh.h.
// The trace kernels are persistent (2048 workgroups pulling work items from a device counter); into how many work
// items a pool is cut is decided ON THE DEVICE from the number of live pools the sort counted (pool_split): one
// when the frame fills the chip by itself, up to 32 when only a few tiles see the mesh — otherwise four waves
// would chew through thousands of rays while 250 CUs idle.
//                  1. compaction + sort, in one: the ballots the primary stage published say which fixed queue
//                     slots hold a ray; every ray's direction is binned (octant x 16x16 cells of the octahedral map,
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
//                  4. the work item's sums go to the frame's fixed-point planes (WfBuffers::fix) by integer atomics:
//                     another share of the pool, the primary stage of the other launch group in flight or nobody may
//                     be adding to the same pixels at the same time — the sums are the same bits.
// On frames that show little the sort strides over the live tiles k_wf_classify listed instead of visiting every pool.
//
// Nothing here decides what the first hit shows; bounce rays are the oracle's rays bit for bit (first stage) and
// their nearest hits are exact, so the stage's output differs from the oracle's only by the fixed-point rounding
// of the terms and of the throughput (tolerance 1e-4).
#include <atomic>
#include <algorithm>
#include <type_traits>

#include "rwr_bvh.h"
#include "rwr_primary.h"
#include "rwr_shade_p2.h"

namespace rwr {

// Per pool, written by k_wf_sort and read by the two trace kernels.
struct PoolInfo {
    uint32_t n_rays;         // 0: nothing to trace
    uint32_t packets;        // 1: packet traversal, 0: per-lane traversal
    uint32_t oct_begin[9];   // sorted position where each octant's rays begin; [8] = n_rays
    uint32_t pad;
};
static_assert(sizeof(PoolInfo) == 48, "PoolInfo is 48 B");

constexpr uint32_t kWfMaxSplit = 32;       // at most this many work items share one pool
constexpr uint32_t kWfWholePools = 1024;   // this many live pools keep the chip busy by themselves
constexpr uint32_t kWfTargetItemsDense = 4096;
constexpr uint32_t kWfTraceGroups = 2048;  // workgroups of a trace launch (persistent: they pull work items)
// Fewer packet pools than `min_packet_pools` (default 64, BvhDevice) in a launch group: the per-lane kernel takes them
// (a packet is one long chain of dependent scalar loads; a handful of them on an otherwise idle chip take longer
// than everything else in the frame).
// device counters of one launch group (one set per ray queue, WfBuffers::counters): pools of each class, work items handed out
enum { kLivePackets = 0, kLiveLane = 1, kWorkPackets = 2, kWorkLane = 3, kCountersPerQueue = 4 };

// float -> uint32 key whose unsigned order is the float order
RWR_DEV uint32_t float_key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
RWR_DEV float key_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// Into how many work items a pool of a class with `live` pools in this launch group is cut: one when the frame
// keeps the chip busy by itself, up to kWfMaxSplit when only a few tiles see anything (a small mesh on an empty
// screen: otherwise four waves would walk through a pool of thousands of rays one after the other while 250 CUs idle).
RWR_DEV uint32_t pool_split(uint32_t live, bool packets, uint32_t lane_items)
{
    if (live == 0u) return 1u;
    // just enough pieces that every workgroup gets a few items and the last round is short: every share zeroes and flushes
    // 12 KiB of sums (measured at configs[2], 4 050 packet pools: two shares beat one by 8 % and five by 6 %; a 135-row band
    // of the same frame, 510 packet pools: 8 shares, not 32 — 2.9 -> 1.9 ms).  Few pools of the per-lane class (a small mesh on
    // an empty screen: their rays are few, the traversals long): one 256-ray chunk per item.
    const uint32_t target = (packets || live >= kWfWholePools) ? kWfTargetItemsDense : lane_items;
    return min(kWfMaxSplit, max(1u, (target + live - 1u) / live));
}

struct SortShared {
    unsigned long long masks[kWfMaxGroup * 8u];
    uint32_t hist[kWfDirBins];   // rays per direction bin, then (in place) where the bin's next ray goes in the sorted list
    uint32_t lo[3], hi[3];   // bounds of the ray origins of the group's first samples (order-preserving keys)
    uint32_t wave_sum[4];
    uint32_t total;
};

// Steps 0 and 1 for one pool: the published ballots -> live slot count; how it will be traced; direction sort ->
// wf.sorted[pool] holds the live slots octant by octant, fine bins in Morton order inside; the pool is appended to
// its class's list.  Queue reads are issued eight at a time per thread (the kernel is all memory latency).
// A pool is traced as PACKETS when it is well filled (min_fill rays) and its rays start close together compared
// with the geometry (origins within bvh.packet_extent): then the rays of one direction bin really travel
// together.  A tile that spans many small faces (a distant instance) is not such a pool.
template <uint32_t NT>
RWR_DEV void sort_pool(SortShared &sh, uint16_t *s_bins, const uint32_t tile, const WfBuffers &wf, PoolInfo *__restrict__ info,
                       uint32_t *__restrict__ counters, uint32_t *__restrict__ pool_list, uint32_t n_tiles,
                       uint32_t sample_count, uint32_t min_fill, float packet_extent, uint32_t dense_rays)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t n_masks = sample_count * 8u, n_slots = sample_count * kWfTilePixels;
    if (tid == 0u) sh.total = 0u;
    if (tid < 3u) { sh.lo[tid] = 0xffffffffu; sh.hi[tid] = 0u; }
    __syncthreads();
    if (tid < n_masks) {
        const unsigned long long m = wf.masks[(size_t)tile * wf.group * 8u + tid];
        sh.masks[tid] = m;
        if (m) atomicAdd(&sh.total, (uint32_t)__popcll(m));
    }
    for (uint32_t i = tid; i < kWfDirBins; i += NT) sh.hist[i] = 0u;
    __syncthreads();
    const uint32_t n_rays = sh.total;
    if (n_rays == 0u) {  // uniform
        if (tid == 0u) info[tile].n_rays = 0u;
        return;
    }
    const size_t pool_base = (size_t)tile * wf.group * kWfTilePixels;
    // (a pool of very many rays sorts into packets that are tight in DIRECTION whatever patch of surface they start from: a
    // packet covers 128 / n_rays of the sphere)
    const bool dense = dense_rays != 0u && n_rays >= dense_rays;
    bool packets = dense || n_rays >= min_fill;
    if (packets && !dense) {   // uniform: how far apart do the rays start?  (the first two samples' slots)
        constexpr int kPerThread = 1024 / (int)NT;
        float4 o[kPerThread];
        bool live_slot[kPerThread];
#pragma unroll
        for (int u = 0; u < kPerThread; u++) {
            const uint32_t e = tid + NT * (uint32_t)u;
            live_slot[u] = e < n_slots && ((sh.masks[e >> 6] >> (e & 63u)) & 1ull);
            o[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (live_slot[u]) o[u] = wf.rays[2u * (pool_base + e)];
        }
        uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < kPerThread; u++)
            if (live_slot[u]) {
                const uint32_t k[3] = {float_key(o[u].x), float_key(o[u].y), float_key(o[u].z)};
#pragma unroll
                for (int c = 0; c < 3; c++) { lo[c] = min(lo[c], k[c]); hi[c] = max(hi[c], k[c]); }
            }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                lo[c] = min(lo[c], (uint32_t)__shfl_xor((int)lo[c], d));
                hi[c] = max(hi[c], (uint32_t)__shfl_xor((int)hi[c], d));
            }
        if ((tid & 63u) == 0u)
            for (int c = 0; c < 3; c++) { atomicMin(&sh.lo[c], lo[c]); atomicMax(&sh.hi[c], hi[c]); }
        __syncthreads();
        float ext = 0.0f;
        if (sh.hi[0] >= sh.lo[0])
            ext = fmaxf(fmaxf(key_float(sh.hi[0]) - key_float(sh.lo[0]), key_float(sh.hi[1]) - key_float(sh.lo[1])),
                        key_float(sh.hi[2]) - key_float(sh.lo[2]));
        packets = ext <= packet_extent;   // a NaN extent: no packets
    }
    // histogram of the direction bins; e = sample * 512 + wave * 128 + k * 64 + lane
    for (uint32_t e0 = tid; e0 < n_slots; e0 += NT * 8u) {
        uint32_t bin[8];   // the primary stage stored every ray's bin beside it: 2 B to read instead of the 32-byte record
        bool live_slot[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t e = e0 + NT * (uint32_t)u;
            live_slot[u] = e < n_slots && ((sh.masks[e >> 6] >> (e & 63u)) & 1ull);
            bin[u] = 0xffffu;
            if (live_slot[u]) bin[u] = min((uint32_t)wf.bins[pool_base + e], kWfDirBins - 1u);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t e = e0 + NT * (uint32_t)u;
            if (e < n_slots) {
                s_bins[e] = (uint16_t)bin[u];
                if (live_slot[u]) atomicAdd(&sh.hist[bin[u]], 1u);
            }
        }
    }
    __syncthreads();
    {   // exclusive prefix over the bins by the first 256 threads: kPer consecutive bins each, shuffle scan inside a wave, four
        // wave totals through LDS
        constexpr uint32_t kPer = kWfDirBins / 256u;
        static_assert(kWfDirBins % 2048u == 0u || kWfDirBins == 512u, "whole bins per thread, an octant starts at a thread's first bin");
        const bool scan = tid < 256u;   // (whole waves)
        uint32_t c[kPer], sum = 0, incl = 0;
        if (scan) {
#pragma unroll
            for (uint32_t k = 0; k < kPer; k++) { c[k] = sh.hist[kPer * tid + k]; sum += c[k]; }
            incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if ((int)(tid & 63u) >= d) incl += up;
            }
            if ((tid & 63u) == 63u) sh.wave_sum[tid >> 6] = incl;
        }
        __syncthreads();
        if (scan) {
            uint32_t before = incl - sum;
            for (uint32_t w = 0; w < (tid >> 6); w++) before += sh.wave_sum[w];
            if ((tid & 31u) == 0u) info[tile].oct_begin[tid >> 5] = before;   // octant o begins at bin (kWfDirBins / 8) o = kPer * (32 o)
#pragma unroll
            for (uint32_t k = 0; k < kPer; k++) { sh.hist[kPer * tid + k] = before; before += c[k]; }   // (its own bins: in place)
        }
    }
    if (tid == 0u) {
        info[tile].oct_begin[8] = n_rays;
        info[tile].n_rays = n_rays;
        info[tile].packets = packets ? 1u : 0u;
        // append to the class's pool list (order of arrival: any order gives the same frame)
        const uint32_t pos = atomicAdd(&counters[(packets ? kLivePackets : kLiveLane)], 1u);
        pool_list[(packets ? 0u : n_tiles) + pos] = tile;
        if (wf.dbg) {
            atomicAdd(&wf.dbg[packets ? 0 : 2], 1ull);
            atomicAdd(&wf.dbg[packets ? 1 : 3], (unsigned long long)n_rays);
        }
    }
    __syncthreads();
    // scatter inside LDS (64 two-byte stores to 64 different places would cost the memory pipe a cycle each), then the
    // sorted list goes out in whole 16-byte pieces
    uint16_t *s_sorted = s_bins + n_slots;
    for (uint32_t e = tid; e < n_slots; e += NT) {
        const uint32_t bin = s_bins[e];
        if (bin != 0xffffu) s_sorted[atomicAdd(&sh.hist[bin], 1u)] = (uint16_t)e;
    }
    __syncthreads();
    uint4 *__restrict__ out = reinterpret_cast<uint4 *>(wf.sorted + pool_base);   // pool_base is a multiple of 512
    const uint4 *src = reinterpret_cast<const uint4 *>(s_sorted);
    for (uint32_t i = tid; i < (n_rays + 7u) / 8u; i += NT) out[i] = src[i];
}

// One workgroup per pool — or, on a frame that shows little, workgroups striding over the live tiles of k_wf_classify's list
// (the other tiles emitted nothing and nobody looks at their pools).
#ifndef RWR_SORT_THREADS
#define RWR_SORT_THREADS 1024
#endif
constexpr uint32_t kWfSortThreads = RWR_SORT_THREADS;   // (a pool's sort is all latency; on whole frames 256 and 1024 measure the same — the
                                                        // sort hides behind the other queue — on a band of 510 pools 1024 is 3x faster)
static_assert(kWfSortThreads >= kWfMaxGroup * 8u && kWfSortThreads >= 256u, "one thread per ballot word of a pool; the prefix sum needs four waves");
template <bool LIST>
__global__ void __launch_bounds__(kWfSortThreads)
k_wf_sort(const WfBuffers wf, PoolInfo *__restrict__ info, uint32_t *__restrict__ counters, uint32_t *__restrict__ pool_list,
          uint32_t n_tiles, uint32_t sample_count, uint32_t min_fill, float packet_extent, uint32_t dense_rays)
{
    __shared__ SortShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint16_t *s_bins = reinterpret_cast<uint16_t *>(s_dyn);   // direction bin of every slot (0xffff: no ray), then the sorted list: 2 x 2 B x sample_count x 512
    if (!LIST) {
        sort_pool<kWfSortThreads>(sh, s_bins, blockIdx.x, wf, info, counters, pool_list, n_tiles, sample_count, min_fill, packet_extent, dense_rays);
        return;
    }
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)*wf.live_count);
    for (uint32_t t = blockIdx.x; t < n; t += gridDim.x) {
        const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)wf.live_list[t]);
        sort_pool<kWfSortThreads>(sh, s_bins, tile, wf, info, counters, pool_list, n_tiles, sample_count, min_fill, packet_extent, dense_rays);
        __syncthreads();   // (the next pool reuses the LDS)
    }
}

// What a trace workgroup keeps in LDS: fixed-point sums of albedo * E(h1) per pixel of the tile.
struct TraceShared {
    unsigned long long acc[kWfTilePixels * 3u];
    uint32_t oct_begin[9], pk_begin[9];   // packet kernel: where each octant's rays / packets begin
    uint32_t next_packet;
    uint32_t item;                        // the work item the workgroup pulled
};

// Adds one ray's contribution albedo(h0) * E(h1) to its pixel's fixed-point sums.  e: the ray's pool slot.
RWR_DEV void add_contribution(TraceShared &sh, uint32_t e, float cr, float cg, float cb)
{
    const uint32_t r = e & (kWfTilePixels - 1u), w = r >> 7, k = (r >> 6) & 1u, l = r & 63u;
    const uint32_t lx = (w & 1u) * 32u + 2u * (l & 15u) + k, ly = (w >> 1) * 4u + (l >> 4);
    unsigned long long *dst = &sh.acc[(ly * kWfTileW + lx) * 3u];
    // float -> u32 conversion saturates and sends NaN / negatives to 0
    atomicAdd(dst + 0, (unsigned long long)(uint32_t)(cr * kWfFixedScale));
    atomicAdd(dst + 1, (unsigned long long)(uint32_t)(cg * kWfFixedScale));
    atomicAdd(dst + 2, (unsigned long long)(uint32_t)(cb * kWfFixedScale));
}

// Step 4: the workgroup's sums -> the frame's fixed-point bounce planes (integer atomics: whichever workgroups
// share the pool, in whatever order, the sums are the same bits).  Call after a barrier.
RWR_DEV void flush_pool(TraceShared &sh, const FrameParams &p, const WfBuffers &wf, uint32_t tile)
{
    const uint32_t tile_x0 = (tile % wf.tiles_x) * kWfTileW, tile_y0 = p.row_begin + (tile / wf.tiles_x) * p.row_pitch;
    const size_t plane = (size_t)p.width * p.height;   // (planes 0..2: red, green, blue; plane 3 is the primary stage's alpha)
    for (uint32_t q = threadIdx.x; q < kWfTilePixels; q += blockDim.x) {
        const uint32_t px = tile_x0 + (q & (kWfTileW - 1u)), py = tile_y0 + q / kWfTileW;
        const unsigned long long sr = sh.acc[q * 3u], sg = sh.acc[q * 3u + 1u], sb = sh.acc[q * 3u + 2u];
        if ((sr | sg | sb) != 0ull && px < p.width && py < p.row_end) {
            const size_t pixel = (size_t)py * p.width + px;
            if (sr) atomicAdd(&wf.fix[pixel], sr);
            if (sg) atomicAdd(&wf.fix[plane + pixel], sg);
            if (sb) atomicAdd(&wf.fix[2u * plane + pixel], sb);
        }
    }
}

// The trace kernels are PERSISTENT: kWfTraceGroups workgroups pull work items — (pool of the class, share of it)
// — from a device counter until the class's live x split items are handed out.  Returns false when none are left.
// Contains barriers; uniform over the workgroup.
RWR_DEV bool next_item(TraceShared &sh, const PoolInfo *__restrict__ info, uint32_t *__restrict__ counters, const uint32_t *__restrict__ pool_list,
                       uint32_t n_tiles, uint32_t want_packets, uint32_t min_packet_pools, uint32_t lane_items, uint32_t &tile, uint32_t &share,
                       uint32_t &n_shares, PoolInfo &pi)
{
    const uint32_t live_p = counters[kLivePackets], live_l = counters[kLiveLane];
    const bool demote = live_p < min_packet_pools;
    if (want_packets && demote) return false;
    const uint32_t live = want_packets ? live_p : live_l + (demote ? live_p : 0u);
    n_shares = pool_split(live, want_packets != 0u, lane_items);
    __syncthreads();   // everybody is done with the previous item (sh.item, sh.acc)
    if (threadIdx.x == 0u) sh.item = atomicAdd(&counters[(want_packets ? kWorkPackets : kWorkLane)], 1u);
    __syncthreads();
    // (values read back from LDS are uniform, but only we know that: readfirstlane keeps everything derived from
    // them — and with it the traversal's node and face loads — on the scalar unit)
    const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.item);
    if (item >= live * n_shares) return false;
    // item -> (share, pool): consecutive items are DIFFERENT pools, so that early items spread over the pools
    share = item / live;
    const uint32_t k = item % live;
    const uint32_t list_pos = want_packets ? k : (k < live_l ? n_tiles + k : k - live_l);
    tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)pool_list[list_pos]);
    pi.n_rays = (uint32_t)__builtin_amdgcn_readfirstlane((int)info[tile].n_rays);
#pragma unroll
    for (int o = 0; o < 9; o++) pi.oct_begin[o] = (uint32_t)__builtin_amdgcn_readfirstlane((int)info[tile].oct_begin[o]);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Pools that are sparse or spread out (silhouette tiles, distant instances): one ray per lane, per-lane BVH
// traversal with the nodelets and the traversal stacks in LDS (rwr_bvh.h).
// STACK16: 16-bit traversal stack entries (rwr_bvh.h) for scenes of at most 4 095 faces and 32 767 nodes.
#ifdef RWR_LANE_OCC   // (6 waves per SIMD: 80 VGPRs + 92 B of scratch, measured below)
#define RWR_LANE_BOUNDS __launch_bounds__(256, RWR_LANE_OCC)
#else
#define RWR_LANE_BOUNDS __launch_bounds__(256)
#endif
// WIDE: a workgroup of 1 024 threads that shares ONE copy of the nodelets in LDS (a BVH too large for four 256-thread
// workgroups per CU to hold a copy each — 508 nodes = 65 KB at configs[3] — but small enough for one per CU): every node fetch
// of the traversal then comes from LDS instead of through the vector memory pipe, 64 different 16-byte pieces per load.
template <bool NODES_IN_LDS, bool NMAP, bool STACK16, bool WIDE = false>
__global__ void __launch_bounds__(WIDE ? 1024 : 256)
k_wf_trace_lane(const FrameParams p, const TriRecord *__restrict__ tris, const ShadeRec *__restrict__ shade,
                const BvhDevice bvh, const float4 *__restrict__ tex, const WfBuffers wf, const PoolInfo *__restrict__ info,
                uint32_t *__restrict__ counters, const uint32_t *__restrict__ pool_list, uint32_t n_tiles)
{
    __shared__ TraceShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const uint32_t tid = threadIdx.x;
    // LDS carve of the dynamic part: [nodelets][traversal stack]
    BvhNode4 *s_nodes = reinterpret_cast<BvhNode4 *>(s_dyn);
    const uint32_t node_bytes = NODES_IN_LDS ? bvh.n_nodes * (uint32_t)sizeof(BvhNode4) : 0u;
    typedef typename std::conditional<STACK16, uint16_t, uint32_t>::type StackT;
    StackT *s_stack = reinterpret_cast<StackT *>(s_dyn + node_bytes);
    bool staged = false;
    uint32_t tile, share, n_shares;
    PoolInfo pi;
    while (next_item(sh, info, counters, pool_list, n_tiles, 0u, bvh.min_packet_pools, bvh.lane_items, tile, share, n_shares, pi)) {
        const uint32_t n_rays = pi.n_rays;
        if (share * 256u >= n_rays) continue;   // uniform
        if (NODES_IN_LDS && !staged) {
            const float4 *src = reinterpret_cast<const float4 *>(bvh.nodes);
            float4 *dst = reinterpret_cast<float4 *>(s_nodes);
            for (uint32_t i = tid; i < bvh.n_nodes * 8u; i += blockDim.x) dst[i] = src[i];
            staged = true;
        }
        for (uint32_t i = tid; i < kWfTilePixels * 3u; i += blockDim.x) sh.acc[i] = 0ull;
        if (tid == 0u) sh.next_packet = 0u;
        __syncthreads();
        const size_t pool_base = (size_t)tile * wf.group * kWfTilePixels;
        const uint16_t *__restrict__ sorted = wf.sorted + pool_base;
        // The item's rays: chunks share, share + n_shares, ... of 256 sorted rays.  Its waves take them 64 rays at a time from a
        // counter in LDS: a wave whose rays had short traversals goes on with the next 64 instead of waiting for the others at
        // a barrier (traversal lengths differ several-fold between waves of sparse pools).
        const uint32_t n_chunks = (n_rays + 255u) / 256u;
        const uint32_t n_sub = share < n_chunks ? 4u * ((n_chunks - share + n_shares - 1u) / n_shares) : 0u;
        for (;;) {
            uint32_t j = 0u;
            if ((tid & 63u) == 0u) j = atomicAdd(&sh.next_packet, 1u);
            j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
            if (j >= n_sub) break;
            const uint32_t i = (share + (j >> 2) * n_shares) * 256u + (j & 3u) * 64u + (tid & 63u);
            if (i >= n_rays) continue;
            const uint32_t e = sorted[i];
            const float4 a = wf.rays[2u * (pool_base + e)], b = wf.rays[2u * (pool_base + e) + 1u];   // one 32-byte record
            const f3 O = mk3(a.x, a.y, a.z), D = mk3(b.x, b.y, b.z);
            const f3 thr = mk3(wf_unorm16_lo(a.w), wf_unorm16_hi(a.w), wf_unorm16_lo(b.w));

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
                const f3 e1 = shade_winner<NMAP>(p, obj, best_t, mh.u, mh.v, mh.ndotd, shade, tex, O, D).colour;
                add_contribution(sh, e, thr.x * e1.x, thr.y * e1.y, thr.z * e1.z);
            }
        }
        __syncthreads();
        flush_pool(sh, p, wf, tile);
    }
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
// Scene data the packet kernel reads at wave-uniform addresses, through the CONSTANT address space: the kernel is a
// persistent loop with global atomics in it, and behind those the compiler would no longer dare to use scalar
// loads for plain global pointers (it cannot see that nobody writes the scene) — node and face records would come
// in through the vector memory pipe, 64 identical addresses per load.
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

#ifndef RWR_PACKET_OCC
#define RWR_PACKET_OCC 4
#endif
template <bool NMAP>
__global__ void __launch_bounds__(256, RWR_PACKET_OCC)
k_wf_trace_packet(const FrameParams p, const TriRecord *__restrict__ tris, const ShadeRec *__restrict__ shade,
                  const BvhDevice bvh, const float4 *__restrict__ tex, const WfBuffers wf, const PoolInfo *__restrict__ info,
                  uint32_t *__restrict__ counters, const uint32_t *__restrict__ pool_list, uint32_t n_tiles)
{
    __shared__ TraceShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const const_ptr<BvhNode4> nodes = to_const_space(bvh.nodes);
    const const_ptr<uint32_t> leaf_faces = to_const_space(bvh.leaf_faces);
    const const_ptr<TriRecord> tris_c = to_const_space(tris);
    uint32_t tile, share, n_shares;
    PoolInfo pi;
    while (next_item(sh, info, counters, pool_list, n_tiles, 1u, bvh.min_packet_pools, bvh.lane_items, tile, share, n_shares, pi)) {
    // packets never straddle an octant: packet q of octant o covers sorted[oct_begin[o] + 128 q ...)
    uint32_t n_packets = 0;
#pragma unroll
    for (int o = 0; o < 8; o++) {
        if (tid == 0u) { sh.oct_begin[o] = pi.oct_begin[o]; sh.pk_begin[o] = n_packets; }
        n_packets += (pi.oct_begin[o + 1] - pi.oct_begin[o] + 127u) / 128u;
    }
    if (share >= n_packets) continue;   // uniform
    if (tid == 0u) { sh.oct_begin[8] = pi.oct_begin[8]; sh.pk_begin[8] = n_packets; sh.next_packet = 0u; }
    for (uint32_t i = tid; i < kWfTilePixels * 3u; i += 256u) sh.acc[i] = 0ull;
    __syncthreads();
    const size_t pool_base = (size_t)tile * wf.group * kWfTilePixels;
    const uint16_t *__restrict__ sorted = wf.sorted + pool_base;

    // A wave always holds its NEXT packet too: which one it is, and the slots of its rays — requested while the current packet
    // is traversed, so that a packet begins with one memory round trip (the ray records), not three in a row.
    struct Packet { uint32_t pk, oct, first, last, e0, e1; };
    auto grab = [&]() {
        Packet q;
        uint32_t pk = 0;
        if (lane == 0u) pk = atomicAdd(&sh.next_packet, 1u);
        q.pk = (uint32_t)__builtin_amdgcn_readfirstlane((int)pk) * n_shares + share;   // this workgroup's share of the packets
        q.oct = 0; q.first = 0; q.last = 0; q.e0 = 0; q.e1 = 0;
        if (q.pk < n_packets) {   // uniform
            uint32_t oct = 0;
#pragma unroll
            for (int o = 1; o < 8; o++) oct += (q.pk >= sh.pk_begin[o]) ? 1u : 0u;
            q.oct = (uint32_t)__builtin_amdgcn_readfirstlane((int)oct);
            q.first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(sh.oct_begin[q.oct] + (q.pk - sh.pk_begin[q.oct]) * 128u));
            q.last = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.oct_begin[q.oct + 1]);
            const uint32_t i0 = q.first + lane, i1 = q.first + 64u + lane;
            q.e0 = sorted[i0 < q.last ? i0 : q.first];
            q.e1 = sorted[i1 < q.last ? i1 : q.first];
        }
        return q;
    };
    Packet next = grab();
    for (;;) {
        const Packet cur_pk = next;
        if (cur_pk.pk >= n_packets) break;
        const uint32_t oct = cur_pk.oct, first = cur_pk.first, last = cur_pk.last;
        // Gray-coded octant -> sign bits (rwr_device.h wf_direction_bin): sz = bit 2, sy = bit 1 ^ sz, sx = bit 0 ^ sy
        const uint32_t sz = oct >> 2, sy = ((oct >> 1) & 1u) ^ sz, sx = (oct & 1u) ^ sy;

        // -- the lane's two rays ---------------------------------------------------------------------------------
        const uint32_t i0 = first + lane, i1 = first + 64u + lane;
        PairRays R;
        R.valid = i2{i0 < last ? -1 : 0, i1 < last ? -1 : 0};
        const uint32_t e0 = cur_pk.e0, e1 = cur_pk.e1;
        const float4 a0 = wf.rays[2u * (pool_base + e0)], b0 = wf.rays[2u * (pool_base + e0) + 1u];   // one 32-byte record per ray
        const float4 a1 = wf.rays[2u * (pool_base + e1)], b1 = wf.rays[2u * (pool_base + e1) + 1u];
        next = grab();
        R.O = v3{f2{a0.x, a1.x}, f2{a0.y, a1.y}, f2{a0.z, a1.z}};
        R.D = v3{f2{b0.x, b1.x}, f2{b0.y, b1.y}, f2{b0.z, b1.z}};
        const v3 thr = v3{f2{wf_unorm16_lo(a0.w), wf_unorm16_lo(a1.w)}, f2{wf_unorm16_hi(a0.w), wf_unorm16_hi(a1.w)},
                          f2{wf_unorm16_lo(b0.w), wf_unorm16_lo(b1.w)}};
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
                    const const_ptr<BvhNode4> nd = nodes + cur;   // wave-uniform index: scalar loads
                    const const_ptr<float> nearx = sx ? nd->bmax_x : nd->bmin_x, farx = sx ? nd->bmin_x : nd->bmax_x;
                    const const_ptr<float> neary = sy ? nd->bmax_y : nd->bmin_y, fary = sy ? nd->bmin_y : nd->bmax_y;
                    const const_ptr<float> nearz = sz ? nd->bmax_z : nd->bmin_z, farz = sz ? nd->bmin_z : nd->bmax_z;
                    // rays that are out of the running: invalid ones, and no distance beyond a ray's best hit matters
                    const f2 tb = f2{R.valid.x ? (best.have.x ? best.t.x : __builtin_inff()) : -1.0f,
                                     R.valid.y ? (best.have.y ? best.t.y : __builtin_inff()) : -1.0f};
                    // per child slot: entry-distance key (0xffffffff: nobody reaches it) and link — wave-uniform values
                    uint32_t k0 = 0xffffffffu, k1 = 0xffffffffu, k2 = 0xffffffffu, k3 = 0xffffffffu;
                    uint32_t c0 = nd->child[0], c1 = nd->child[1], c2 = nd->child[2], c3 = nd->child[3];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t child = i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : c3;
                        if (child == kBvhEmpty) continue;   // uniform
                        const f2 x0 = fma2(splat(nearx[i]), R.ix, R.ox), x1 = fma2(splat(farx[i]), R.ix, R.ox);
                        const f2 y0 = fma2(splat(neary[i]), R.iy, R.oy), y1 = fma2(splat(fary[i]), R.iy, R.oy);
                        const f2 z0 = fma2(splat(nearz[i]), R.iz, R.oz), z1 = fma2(splat(farz[i]), R.iz, R.oz);
                        const f2 tnear = max2(max2(x0, y0), z0), tfar = min2(min2(x1, y1), z1);
                        // conservative like rwr_bvh.h bvh_inner_step (relative slack on both distances, <=), in fewer
                        // instructions: entry distances below 0 clamp to 0 whatever their slack, and a box whose exit
                        // distance is negative lies behind the ray with or without slack — so the slack is a scale
                        const f2 lo = max2(tnear * 0.99996f, splat(0.0f));
                        const f2 hi = min2(fma2(tfar, splat(1.00004f), splat(1e-30f)), tb);
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
                        const uint32_t idx = leaf_faces[lf + k];   // uniform: scalar load
                        if (idx < p.n_tris) {
                            const TriRecord T = load_tri_record(tris_c + idx);   // into scalar registers
                            intersect_pair_any_order(T, idx, R, best);
                        }
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
            f2 er = splat(0.0f), eg = splat(0.0f), eb = splat(0.0f);
            const i2 is_mesh = have & (obj >= 0);
            if (__any(any2(is_mesh))) {   // mesh winners: both rays of a lane at once (packed arithmetic, rwr_shade_p2.h)
                const ShadeRec none = {};
                if (NMAP) shade_mesh_pair<true, false, true>(p, shade, tex, obj, none, best, R.D, er, eg, eb);
                else if (p.n_materials > 1u) shade_mesh_pair<true, false>(p, shade, tex, obj, none, best, R.D, er, eg, eb);
                else shade_mesh_pair<false, false>(p, shade, tex, obj, none, best, R.D, er, eg, eb);
            }
            if (__any(any2(have & (obj < -1)))) {   // sphere winners (rare): one ray at a time
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int o = k ? obj.y : obj.x;
                    if ((k ? have.y : have.x) && o < -1) {
                        const f3 Ok = lane3(R.O, k), Dk = lane3(R.D, k);
                        const f3 ek = shade_winner<false>(p, o, k ? best_t.y : best_t.x, 0.0f, 0.0f, 0.0f, shade, tex, Ok, Dk).colour;
                        if (k) { er.y = ek.x; eg.y = ek.y; eb.y = ek.z; } else { er.x = ek.x; eg.x = ek.y; eb.x = ek.z; }
                    }
                }
            }
            const f2 cr = thr.x * er, cg = thr.y * eg, cb = thr.z * eb;
            if (have.x) add_contribution(sh, e0, cr.x, cg.x, cb.x);
            if (have.y) add_contribution(sh, e1, cr.y, cg.y, cb.y);
        }
    }
    __syncthreads();
    flush_pool(sh, p, wf, tile);
    }   // next work item
}

hipError_t launch_wf_bounce(hipStream_t s, const FrameParams &fp, const TriRecord *tris, const ShadeRec *shade,
                            const BvhDevice &bvh, const float4 *tex, const WfBuffers &wf, uint32_t n_tiles,
                            uint32_t sample_count, uint32_t packet_min_rays, void *pool_info, uint32_t *pool_list)
{
    if (n_tiles == 0 || sample_count == 0) return hipSuccess;
    uint32_t *counters = wf.counters;   // this queue's set, zeroed by the primary stage that filled the queue
    PoolInfo *info = static_cast<PoolInfo *>(pool_info);
    // well-filled, compact pools: packet traversal (its one stack is a VGPR of 64 entries)
    const bool packets = bvh.stack_depth <= 64u && packet_min_rays <= sample_count * kWfTilePixels && bvh.packet_extent > 0.0f;
    const size_t sort_lds = 2u * (size_t)sample_count * kWfTilePixels * sizeof(uint16_t);
    if (sort_lds > 64u * 1024u) {   // beyond the default limit of dynamic LDS (groups of more than 32 samples)
        // a function attribute belongs to the function ON ONE DEVICE: raised once per device a context renders on
        static std::atomic<uint64_t> raised_on{0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        if (!(raised_on.load(std::memory_order_acquire) & bit)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wf_sort<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wf_sort<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (e != hipSuccess) return e;
            raised_on.fetch_or(bit, std::memory_order_release);
        }
    }
    if (wf.live_list)
        hipLaunchKernelGGL(k_wf_sort<true>, dim3(std::min(n_tiles, 2048u)), dim3(kWfSortThreads), sort_lds, s, wf, info, counters, pool_list, n_tiles,
                           sample_count, packets ? packet_min_rays : 0xffffffffu, bvh.packet_extent, bvh.packet_dense_rays);
    else
        hipLaunchKernelGGL(k_wf_sort<false>, dim3(n_tiles), dim3(kWfSortThreads), sort_lds, s, wf, info, counters, pool_list, n_tiles,
                           sample_count, packets ? packet_min_rays : 0xffffffffu, bvh.packet_extent, bvh.packet_dense_rays);
    const dim3 grid(std::min(kWfTraceGroups, n_tiles * kWfMaxSplit));
    const bool nmap = (fp.flags & RWR_FLAG_NORMAL_MAP) != 0;
    if (packets) {
        if (nmap) hipLaunchKernelGGL((k_wf_trace_packet<true>), grid, dim3(256), 0, s, fp, tris, shade, bvh, tex, wf, info, counters, pool_list, n_tiles);
        else hipLaunchKernelGGL((k_wf_trace_packet<false>), grid, dim3(256), 0, s, fp, tris, shade, bvh, tex, wf, info, counters, pool_list, n_tiles);
    }
    const bool stack16 = bvh.n_nodes <= 0x7fffu && fp.n_tris <= 4095u;   // node indices and leaf links (first << 3 | count - 1) in 15 bits
    const size_t fixed = (size_t)bvh.stack_depth * 256u * (stack16 ? 2u : 4u);
    const size_t node_bytes = (size_t)bvh.n_nodes * sizeof(BvhNode4);
    // nodelets go to LDS when the workgroup then still fits a CU at least four times (160 KiB LDS, 12 KiB static)
#define RWR_LANE_LAUNCH(L, N, S16, BYTES) hipLaunchKernelGGL((k_wf_trace_lane<L, N, S16>), grid, dim3(256), BYTES, s, fp, tris, shade, bvh, tex, wf, info, counters, pool_list, n_tiles)
#define RWR_LANE_LAUNCH2(L, BYTES) \
    if (nmap) { if (stack16) RWR_LANE_LAUNCH(L, true, true, BYTES); else RWR_LANE_LAUNCH(L, true, false, BYTES); } \
    else { if (stack16) RWR_LANE_LAUNCH(L, false, true, BYTES); else RWR_LANE_LAUNCH(L, false, false, BYTES); }
    const size_t fixed_wide = 4u * fixed, wide_bytes = node_bytes + fixed_wide;
    if (node_bytes + fixed <= 28u * 1024u) { RWR_LANE_LAUNCH2(true, node_bytes + fixed) }
    else if (bvh.wide_lane && !nmap && stack16 && wide_bytes + 14u * 1024u <= 160u * 1024u) {
        // a BVH too large for a copy per 256-thread workgroup, small enough for one copy per CU: 1 024-thread workgroups
        static std::atomic<uint64_t> wide_raised_on{0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        if (!(wide_raised_on.load(std::memory_order_acquire) & bit)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wf_trace_lane<true, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 146 * 1024);
            if (e != hipSuccess) return e;
            wide_raised_on.fetch_or(bit, std::memory_order_release);
        }
        hipLaunchKernelGGL((k_wf_trace_lane<true, false, true, true>), dim3(std::min(512u, n_tiles * kWfMaxSplit)), dim3(1024), wide_bytes, s, fp, tris, shade, bvh, tex,
                           wf, info, counters, pool_list, n_tiles);
    }
    else { RWR_LANE_LAUNCH2(false, fixed) }
#undef RWR_LANE_LAUNCH2
#undef RWR_LANE_LAUNCH
    return hipGetLastError();
}

size_t wf_pool_info_bytes() { return sizeof(PoolInfo); }

hipError_t preload_kernels_wf_bounce()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>((&k_wf_trace_packet<false>)));
}

}  // namespace rwr
