// The interleaved partition of a frame across ranks and the layout of its ONE gather (include/rwr_hip.h, "multi-GPU
// frames").  The reference is single-device (/root/reference/src/lib.rs:266-303, one submit :1226), so there
// is no reference line this follows: north_star asks for a tile partition with a single RCCL gather of the finished tiles.
//
// A *strip* is kStripRows = 8 framebuffer rows (the height of every render kernel's workgroup tile); rank r of `world`
// renders strips r, r + world, ...  Its *message* is those strips back to back in ascending order (a short last strip
// of the frame, if it is the rank's, is therefore the end of its message).  The root's *receive buffer* holds the
// messages of ranks 0, 1, ... one after the other, each starting on a whole-strip boundary (so every strip in it is
// 32 * width bytes from the buffer's start: 16-byte aligned for any width).
//
// ONE definition for everything that has to agree: the sizes of the RCCL sends and receives and the receive offsets
// (context.cpp), the pack and deal-out kernels (kernels_dist.hip), the host-memory forms and rwr_dist_strip_layout
// (what the tests and a host binding see).
#pragma once
#include <cstdint>

#if defined(__HIPCC__) || defined(__HIP__)
#define RWR_HD __host__ __device__
#else
#define RWR_HD
#endif

namespace rwr {

constexpr uint32_t kStripRowsLayout = 8;   // == kStripRows (rwr_internal.h), == RWR_STRIP_ROWS (rwr_hip.h)

struct StripLayout {
    uint32_t height, world;
    uint32_t n_strips;    // strips of the frame, the last one possibly short
    uint32_t tail_rows;   // rows of a short last strip, 0 when the height is a multiple of 8

    RWR_HD static StripLayout make(uint32_t height, uint32_t world)
    {
        StripLayout L;
        L.height = height;
        L.world = world;
        L.n_strips = (height + kStripRowsLayout - 1u) / kStripRowsLayout;
        L.tail_rows = height % kStripRowsLayout;
        return L;
    }
    // rows of strip s of the FRAME
    RWR_HD uint32_t strip_rows(uint32_t s) const
    {
        const uint32_t y0 = s * kStripRowsLayout;
        return y0 >= height ? 0u : (height - y0 < kStripRowsLayout ? height - y0 : kStripRowsLayout);
    }
    // strips rank r owns: r, r + world, ... below n_strips
    RWR_HD uint32_t strips_of(uint32_t r) const { return r < n_strips ? (n_strips - r + world - 1u) / world : 0u; }
    RWR_HD bool owns_tail(uint32_t r) const { return tail_rows != 0u && (n_strips - 1u) % world == r; }
    // rows of rank r's message (= what it sends)
    RWR_HD uint32_t rows_of(uint32_t r) const
    {
        return strips_of(r) * kStripRowsLayout - (owns_tail(r) ? kStripRowsLayout - tail_rows : 0u);
    }
    // row of the root's receive buffer at which rank r's message starts: whole strips of the ranks before it
    RWR_HD uint32_t recv_row(uint32_t r) const
    {
        uint32_t rows = 0;
        for (uint32_t q = 0; q < r && q < world; q++) rows += strips_of(q) * kStripRowsLayout;
        return rows;
    }
    RWR_HD uint32_t recv_rows_total() const { return n_strips * kStripRowsLayout; }
    // strip s of the frame: whose it is, and which strip of that rank's message
    RWR_HD uint32_t owner(uint32_t s) const { return s % world; }
    RWR_HD uint32_t index_in_message(uint32_t s) const { return s / world; }
    // j-th strip of rank r's message: which strip of the frame
    RWR_HD uint32_t frame_strip(uint32_t r, uint32_t j) const { return r + j * world; }
};

}  // namespace rwr
