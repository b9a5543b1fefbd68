// Host-side BVH construction for incoherent (bounce) rays: binned-SAH BVH2,
// collapsed to 4-wide nodes ("nodelets") that the traversal kernel stages in LDS.
// The reference has no acceleration structure (brute-force loop,
// /root/reference/src/models/triangle_list/compute.wgsl:190-202); the BVH is
// exact-result-preserving: boxes are padded, traversal keeps every node whose
// entry distance is <= the best hit so far, and the hit test + (t, face index)
// selection rule are the reference's, so the winner equals the brute-force
// winner (tests/test_gpu_path.py checks it face for face).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace rwr {

// 128-byte node: SoA boxes of up to four children + child links.
//   child[i] == kBvhEmpty                      : unused slot (box inverted)
//   child[i] &  kBvhLeafBit                    : leaf, faces leaf_faces[first .. first+count)
//                                                first = (child & ~kBvhLeafBit) >> 3, count = (child & 7) + 1
//   otherwise                                  : index of an inner node
struct alignas(16) BvhNode4 {
    float bmin_x[4], bmin_y[4], bmin_z[4];
    float bmax_x[4], bmax_y[4], bmax_z[4];
    uint32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(BvhNode4) == 128, "BvhNode4 is 128 B");

constexpr uint32_t kBvhEmpty = 0xffffffffu;
constexpr uint32_t kBvhLeafBit = 0x80000000u;
constexpr uint32_t kBvhMaxLeafDefault = 2;  // the exact hit test costs ~4x a box test: prefer small leaves
// Deepest 4-wide tree the traversal kernels are sized for: a lane's stack holds up to 3 * depth + 2 entries
// (38 KiB of LDS per 256-thread workgroup at this depth; the packet traversal's stack is one 64-lane register).
// A binned-SAH tree over a spatially skewed scene (a dense cluster plus far outliers) can be far deeper than
// log4(n); such a scene is rebuilt with object-median splits, which bound the depth by ~log2(n) / 2 + 1
// (12 covers millions of faces).
constexpr uint32_t kBvhMaxDepth = 12;

struct Bvh {
    std::vector<BvhNode4> nodes;       // nodes[0] is the root (present even for 1 face)
    std::vector<uint32_t> leaf_faces;  // face indices, grouped per leaf, ascending inside a leaf
    uint32_t max_depth = 0;            // of the 4-wide tree
    float mean_leaf_extent = 0.0f;     // mean over the leaves of the largest box dimension (the scale of the geometry)
};

namespace bvh_detail {

struct Box {
    float lo[3], hi[3];
    void reset()
    {
        for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const float *p)
    {
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
    }
    void grow(const Box &b)
    {
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
    }
    float half_area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.0f : dx * dy + dy * dz + dz * dx;
    }
};

struct Node2 {
    Box box;
    int left = -1, right = -1;    // inner
    uint32_t first = 0, count = 0;  // leaf (count > 0)
};

struct Builder {
    const float *tri;  // n x 9 floats (p0, p1, p2)
    uint32_t max_leaf = kBvhMaxLeafDefault;
    bool median_only = false;  // object-median splits instead of binned SAH (bounded depth)
    std::vector<Box> tbox;
    std::vector<float> cen;  // n x 3
    std::vector<uint32_t> order;
    std::vector<Node2> nodes;

    int build(uint32_t first, uint32_t count)
    {
        Node2 node;
        node.box.reset();
        Box cbox;
        cbox.reset();
        for (uint32_t i = first; i < first + count; i++) {
            node.box.grow(tbox[order[i]]);
            cbox.grow(&cen[3 * order[i]]);
        }
        const int self = (int)nodes.size();
        nodes.push_back(node);
        if (count <= max_leaf) {
            nodes[self].first = first;
            nodes[self].count = count;
            return self;
        }
        // binned SAH over the widest centroid axis (fall back to a median split)
        int axis = 0;
        float ext = -1.0f;
        for (int k = 0; k < 3; k++)
            if (cbox.hi[k] - cbox.lo[k] > ext) { ext = cbox.hi[k] - cbox.lo[k]; axis = k; }
        uint32_t mid = first + count / 2;
        if (median_only) {
            if (ext > 0.0f && std::isfinite(ext))
                std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](uint32_t a, uint32_t b) {
                    const float ca = cen[3 * (size_t)a + axis], cb = cen[3 * (size_t)b + axis];
                    return ca < cb || (ca == cb && a < b);
                });
        } else if (ext > 0.0f && std::isfinite(ext)) {
            constexpr int kBins = 16;
            Box bb[kBins];
            uint32_t bc[kBins] = {};
            for (auto &b : bb) b.reset();
            const float scale = kBins / ext;
            auto bin_of = [&](uint32_t t) {
                int b = (int)((cen[3 * t + axis] - cbox.lo[axis]) * scale);
                return std::min(std::max(b, 0), kBins - 1);
            };
            for (uint32_t i = first; i < first + count; i++) {
                const int b = bin_of(order[i]);
                bb[b].grow(tbox[order[i]]);
                bc[b]++;
            }
            float right_area[kBins];
            uint32_t right_cnt[kBins];
            Box acc;
            acc.reset();
            uint32_t cnt = 0;
            for (int b = kBins - 1; b > 0; b--) {
                acc.grow(bb[b]);
                cnt += bc[b];
                right_area[b] = acc.half_area();
                right_cnt[b] = cnt;
            }
            acc.reset();
            cnt = 0;
            float best = std::numeric_limits<float>::infinity();
            int best_split = -1;
            for (int b = 0; b < kBins - 1; b++) {
                acc.grow(bb[b]);
                cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                const float cost = acc.half_area() * cnt + right_area[b + 1] * right_cnt[b + 1];
                if (cost < best) { best = cost; best_split = b; }
            }
            if (best_split >= 0) {
                auto it = std::stable_partition(order.begin() + first, order.begin() + first + count,
                                                [&](uint32_t t) { return bin_of(t) <= best_split; });
                mid = (uint32_t)(it - order.begin());
                if (mid == first || mid == first + count) mid = first + count / 2;
            }
        }
        if (mid == first + count / 2 && !(ext > 0.0f)) {
            // coincident centroids: split by face index to keep it deterministic
        }
        const int l = build(first, mid - first);
        const int r = build(mid, first + count - mid);
        nodes[self].left = l;
        nodes[self].right = r;
        return self;
    }
};

}  // namespace bvh_detail

// tri: n faces x 9 floats (world-space p0, p1, p2 as the device holds them).
inline Bvh build_bvh_with(const float *tri, uint32_t n, uint32_t max_leaf, bool median_only)
{
    using namespace bvh_detail;
    Bvh out;
    Builder b;
    b.tri = tri;
    b.median_only = median_only;
    b.max_leaf = std::min(std::max(max_leaf, 1u), 8u);
    b.tbox.resize(n);
    b.cen.resize((size_t)3 * n);
    b.order.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        b.order[i] = i;
        b.tbox[i].reset();
        for (int v = 0; v < 3; v++) b.tbox[i].grow(tri + 9 * (size_t)i + 3 * v);
        for (int k = 0; k < 3; k++) b.cen[3 * (size_t)i + k] = 0.5f * (b.tbox[i].lo[k] + b.tbox[i].hi[k]);
    }
    if (n == 0) {
        BvhNode4 root;
        std::memset(&root, 0, sizeof root);
        for (int i = 0; i < 4; i++) {
            root.bmin_x[i] = root.bmin_y[i] = root.bmin_z[i] = std::numeric_limits<float>::infinity();
            root.bmax_x[i] = root.bmax_y[i] = root.bmax_z[i] = -std::numeric_limits<float>::infinity();
            root.child[i] = kBvhEmpty;
        }
        out.nodes.push_back(root);
        return out;
    }
    b.nodes.reserve(2 * (size_t)n);
    const int root2 = b.build(0, n);

    {
        double sum = 0.0;
        size_t n_leaves = 0;
        for (const Node2 &nd : b.nodes)
            if (nd.count) {
                const float dx = nd.box.hi[0] - nd.box.lo[0], dy = nd.box.hi[1] - nd.box.lo[1], dz = nd.box.hi[2] - nd.box.lo[2];
                const float ext = std::max(dx, std::max(dy, dz));
                if (std::isfinite(ext)) { sum += ext; n_leaves++; }
            }
        out.mean_leaf_extent = n_leaves ? (float)(sum / (double)n_leaves) : 0.0f;
    }
    // leaves: faces ascending inside a leaf (the tie rule is decided by index, order is cosmetic)
    out.leaf_faces = b.order;
    for (const Node2 &nd : b.nodes)
        if (nd.count) std::sort(out.leaf_faces.begin() + nd.first, out.leaf_faces.begin() + nd.first + nd.count);

    auto pad_box = [](const Box &bx, BvhNode4 &dst, int slot) {
        // padding: 1e-5 of the largest coordinate magnitude + 1e-6, well above the f32 slop of the hit test
        float m = 0.0f;
        for (int k = 0; k < 3; k++) m = std::max(m, std::max(std::fabs(bx.lo[k]), std::fabs(bx.hi[k])));
        const float e = 1e-5f * m + 1e-6f;
        dst.bmin_x[slot] = bx.lo[0] - e; dst.bmin_y[slot] = bx.lo[1] - e; dst.bmin_z[slot] = bx.lo[2] - e;
        dst.bmax_x[slot] = bx.hi[0] + e; dst.bmax_y[slot] = bx.hi[1] + e; dst.bmax_z[slot] = bx.hi[2] + e;
    };
    auto leaf_link = [](const Node2 &nd) { return kBvhLeafBit | (nd.first << 3) | (nd.count - 1u); };

    // collapse: a 4-wide node adopts its BVH2 children, then repeatedly splits the adopted
    // inner child with the largest area until four slots are filled
    struct Work { int node2; uint32_t node4; uint32_t depth; };
    std::vector<Work> stack;
    out.nodes.emplace_back();
    stack.push_back({root2, 0u, 1u});
    while (!stack.empty()) {
        const Work w = stack.back();
        stack.pop_back();
        out.max_depth = std::max(out.max_depth, w.depth);
        std::vector<int> kids;
        const Node2 &top = b.nodes[w.node2];
        if (top.count) kids.push_back(w.node2);
        else { kids.push_back(top.left); kids.push_back(top.right); }
        while (kids.size() < 4) {
            int pick = -1;
            float area = -1.0f;
            for (size_t i = 0; i < kids.size(); i++) {
                const Node2 &k = b.nodes[kids[i]];
                if (!k.count && k.box.half_area() > area) { area = k.box.half_area(); pick = (int)i; }
            }
            if (pick < 0) break;
            const Node2 k = b.nodes[kids[pick]];
            kids[pick] = k.left;
            kids.push_back(k.right);
        }
        BvhNode4 n4;
        std::memset(&n4, 0, sizeof n4);
        for (int i = 0; i < 4; i++) {
            n4.bmin_x[i] = n4.bmin_y[i] = n4.bmin_z[i] = std::numeric_limits<float>::infinity();
            n4.bmax_x[i] = n4.bmax_y[i] = n4.bmax_z[i] = -std::numeric_limits<float>::infinity();
            n4.child[i] = kBvhEmpty;
        }
        for (size_t i = 0; i < kids.size(); i++) {
            const Node2 &k = b.nodes[kids[i]];
            pad_box(k.box, n4, (int)i);
            if (k.count) {
                n4.child[i] = leaf_link(k);
            } else {
                const uint32_t id = (uint32_t)out.nodes.size();
                out.nodes.emplace_back();
                n4.child[i] = id;
                stack.push_back({kids[i], id, w.depth + 1});
            }
        }
        out.nodes[w.node4] = n4;
    }
    return out;
}

// Binned SAH; rebuilt with object-median splits when that tree is deeper than kBvhMaxDepth (callers check
// max_depth again: a scene beyond millions of faces could still exceed it).
inline Bvh build_bvh(const float *tri, uint32_t n, uint32_t max_leaf = kBvhMaxLeafDefault)
{
    Bvh bvh = build_bvh_with(tri, n, max_leaf, false);
    if (bvh.max_depth > kBvhMaxDepth) bvh = build_bvh_with(tri, n, max_leaf, true);
    return bvh;
}

}  // namespace rwr
