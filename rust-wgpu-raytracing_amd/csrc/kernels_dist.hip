// Pack and deal-out of the interleaved partition's gather (rwr_dist_gather_strips_rgba8; layout: rwr_strips.h).
// No reference counterpart: the reference is single-device (/root/reference/src/lib.rs:266-303, one submit :1226).
//
//   pack     (every rank):  frame strips r, r + world, ...  ->  the rank's message, strips back to back
//   deal-out (root):        receive buffer (messages of ranks 0, 1, ... side by side)  ->  the frame
//
// One launch each (a strided hipMemcpy2DAsync per rank cost the root up to 2 x world copy commands per frame, each a
// few microseconds of a frame that renders in ten).  HBM-bound byte work: every strip is one contiguous span on both
// sides, 16-byte aligned at both ends' starts for any width (a strip is 32 x width bytes), moved as dwordx4 with a dword
// tail for a short last strip of an odd width.
#include <hip/hip_runtime.h>

#include <cstring>

#include "rwr_internal.h"
#include "rwr_strips.h"

namespace rwr {
static_assert(kStripRowsLayout == kStripRows, "one strip height");

namespace {
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kCopyThreads = 256;
constexpr uint32_t kCopyChunk = kCopyThreads * 16u * 4u;   // bytes a workgroup moves: 4 dwordx4 per thread

// DEAL = false: blockIdx.y = j, the j-th strip of rank `rank`'s message; src = the rank's frame, dst = its message.
// DEAL = true:  blockIdx.y = s, strip s of the frame; src = the root's receive buffer, dst = the assembled frame.
template <bool DEAL>
__global__ __launch_bounds__(kCopyThreads) void k_strips_copy(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                              StripLayout L, uint32_t rank, uint32_t row_bytes)
{
    uint32_t s;
    size_t src_off, dst_off;
    const size_t strip_bytes = (size_t)row_bytes * kStripRows;
    if (DEAL) {
        s = blockIdx.y;
        src_off = (size_t)L.recv_row(L.owner(s)) * row_bytes + (size_t)L.index_in_message(s) * strip_bytes;
        dst_off = (size_t)s * strip_bytes;
    } else {
        s = L.frame_strip(rank, blockIdx.y);
        src_off = (size_t)s * strip_bytes;
        dst_off = (size_t)blockIdx.y * strip_bytes;
    }
    const uint32_t bytes = L.strip_rows(s) * row_bytes;   // a multiple of 4
    const uint32_t begin = blockIdx.x * kCopyChunk;
    if (begin >= bytes) return;
    const uint32_t end = min(bytes, begin + kCopyChunk);
    const uint8_t *sp = src + src_off;
    uint8_t *dp = dst + dst_off;
    const uint32_t vec_end = begin + ((end - begin) & ~15u);
    for (uint32_t at = begin + threadIdx.x * 16u; at < vec_end; at += kCopyThreads * 16u)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const u4 *>(sp + at)), reinterpret_cast<u4 *>(dp + at));
    const uint32_t at = vec_end + threadIdx.x * 4u;   // < 16 bytes are left
    if (at < end) *reinterpret_cast<uint32_t *>(dp + at) = *reinterpret_cast<const uint32_t *>(sp + at);
}
}  // namespace

hipError_t launch_strips_pack(hipStream_t s, const StripLayout &L, uint32_t rank, uint32_t row_bytes, const uint8_t *frame, uint8_t *message)
{
    const uint32_t n = L.strips_of(rank);
    if (n == 0u || row_bytes == 0u) return hipSuccess;
    const dim3 grid((row_bytes * kStripRows + kCopyChunk - 1u) / kCopyChunk, n);
    hipLaunchKernelGGL(k_strips_copy<false>, grid, dim3(kCopyThreads), 0, s, message, frame, L, rank, row_bytes);
    return hipGetLastError();
}

hipError_t launch_strips_deal(hipStream_t s, const StripLayout &L, uint32_t row_bytes, const uint8_t *recv, uint8_t *frame)
{
    if (L.n_strips == 0u || row_bytes == 0u) return hipSuccess;
    const dim3 grid((row_bytes * kStripRows + kCopyChunk - 1u) / kCopyChunk, L.n_strips);
    hipLaunchKernelGGL(k_strips_copy<true>, grid, dim3(kCopyThreads), 0, s, frame, recv, L, 0u, row_bytes);
    return hipGetLastError();
}

// The same two steps on HOST memory, by the same layout functions (rwr_dist_host_pack_strips / _deal_strips: hosts that
// stage a gather through CPU memory, and the world-size-2/3 tests over gloo).  Byte moves only — nothing is rendered here.
void strips_pack_host(const StripLayout &L, uint32_t rank, size_t row_bytes, const uint8_t *frame, uint8_t *message)
{
    const size_t strip_bytes = row_bytes * kStripRows;
    for (uint32_t j = 0; j < L.strips_of(rank); j++) {
        const uint32_t s = L.frame_strip(rank, j);
        std::memcpy(message + (size_t)j * strip_bytes, frame + (size_t)s * strip_bytes, (size_t)L.strip_rows(s) * row_bytes);
    }
}

void strips_deal_host(const StripLayout &L, size_t row_bytes, const uint8_t *recv, uint8_t *frame)
{
    const size_t strip_bytes = row_bytes * kStripRows;
    for (uint32_t s = 0; s < L.n_strips; s++)
        std::memcpy(frame + (size_t)s * strip_bytes,
                    recv + (size_t)L.recv_row(L.owner(s)) * row_bytes + (size_t)L.index_in_message(s) * strip_bytes,
                    (size_t)L.strip_rows(s) * row_bytes);
}

hipError_t preload_kernels_dist()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_strips_copy<true>));
}

}  // namespace rwr
