// Traversal-divergence simulator for the bounce stage (design aid, not product code).
// Replays csrc/rwr_bvh.h's per-lane traversal on the CPU for the rays written by make_rays.py, records every
// ray's sequence of steps, and prices scheduling policies in wave instructions per ray:
//   A  one ray per lane, 64 consecutive queue entries per wave, while-while (the round-1 kernel)
//   B  k rays per lane in sequence (lane j takes entries j, j+64, ...), while-while
//   C  persistent wave with refill: lanes that finished fetch the next ray when >= T lanes are idle
// g++ -O2 -std=c++17 -I rust-wgpu-raytracing_amd/csrc tools/sim/sim_bvh.cpp -o /tmp/sim_bvh
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <cmath>
#include <vector>
#include "bvh.hpp"
using namespace rwr;

struct Step { uint8_t kind; uint8_t stages[8]; uint8_t nf; uint32_t id; };  // kind 0 inner, 1 leaf; stages[f] = how far face f's test got (1..4)
struct RayTrace { std::vector<Step> steps; };

static int g_cost_inner = 100, g_cost_stage[5] = {0, 28, 22, 22, 30};  // plane+div, edge0, edge1, edge2+select

struct Tri { float p0[3], p1[3], p2[3], N[3], d; };
static std::vector<Tri> g_tris;

static inline float dot(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross(const float *a, const float *b, float *o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }

// returns stage reached (1 = rejected at plane, 2 = at edge0, 3 = edge1, 4 = ran to the end); updates best
static int test_face(uint32_t idx, const float *O, const float *D, bool &have, float &best_t, uint32_t &best_idx)
{
    const Tri &T = g_tris[idx];
    const float nd = dot(T.N, D);
    bool hit = !(std::fabs(nd) < 1e-6f);
    const float t = -(dot(T.N, O) + T.d) / nd;
    hit &= !(t < 0.0f);
    hit &= !(have && t > best_t);
    if (!hit) return 1;
    float P[3] = {O[0] + t * D[0], O[1] + t * D[1], O[2] + t * D[2]}, e[3], q[3], C[3];
    for (int k = 0; k < 3; k++) { e[k] = T.p1[k] - T.p0[k]; q[k] = P[k] - T.p0[k]; }
    cross(e, q, C);
    if (dot(T.N, C) < 0.0f) return 2;
    for (int k = 0; k < 3; k++) { e[k] = T.p2[k] - T.p1[k]; q[k] = P[k] - T.p1[k]; }
    cross(e, q, C);
    if (dot(T.N, C) < 0.0f) return 3;
    for (int k = 0; k < 3; k++) { e[k] = T.p0[k] - T.p2[k]; q[k] = P[k] - T.p2[k]; }
    cross(e, q, C);
    if (!(dot(T.N, C) < 0.0f) && (!have || t < best_t || (t == best_t && idx < best_idx))) { have = true; best_t = t; best_idx = idx; }
    return 4;
}

static RayTrace trace(const Bvh &bvh, const float *O, const float *D)
{
    RayTrace rt;
    float inv[3], o[3];
    int nearp[3];
    for (int k = 0; k < 3; k++) { inv[k] = 1.0f / D[k]; o[k] = -O[k] * inv[k]; nearp[k] = std::signbit(inv[k]) ? 1 : 0; }
    std::vector<uint32_t> stack;
    uint32_t cur = 0;
    bool have = false; float best_t = 0; uint32_t best_idx = 0;
    for (;;) {
        if (!(cur & kBvhLeafBit)) {
            Step s{}; s.kind = 0; s.id = cur; rt.steps.push_back(s);
            const BvhNode4 &n = bvh.nodes[cur];
            uint32_t key[4];
            const float tb = have ? best_t : INFINITY;
            for (int i = 0; i < 4; i++) {
                const float lo3[3] = {n.bmin_x[i], n.bmin_y[i], n.bmin_z[i]}, hi3[3] = {n.bmax_x[i], n.bmax_y[i], n.bmax_z[i]};
                float tn = -INFINITY, tf = INFINITY;
                for (int k = 0; k < 3; k++) {
                    const float a = std::fma(nearp[k] ? hi3[k] : lo3[k], inv[k], o[k]), b = std::fma(nearp[k] ? lo3[k] : hi3[k], inv[k], o[k]);
                    tn = std::fmax(tn, a); tf = std::fmin(tf, b);
                }
                const float lo = std::fmax(tn - 4e-5f * std::fabs(tn), 0.0f), hi = std::fmin(tf + 4e-5f * std::fabs(tf) + 1e-30f, tb);
                uint32_t lob; std::memcpy(&lob, &lo, 4);
                key[i] = (lo <= hi && n.child[i] != kBvhEmpty) ? ((lob & ~3u) | i) : 0xffffffffu;
            }
            const uint32_t kmin = std::min(std::min(key[0], key[1]), std::min(key[2], key[3]));
            if (kmin == 0xffffffffu) { if (stack.empty()) break; cur = stack.back(); stack.pop_back(); continue; }
            uint32_t next = 0;
            for (int i = 0; i < 4; i++) {
                if ((uint32_t)i == (kmin & 3u)) next = n.child[i];
                else if (key[i] != 0xffffffffu) stack.push_back(n.child[i]);
            }
            cur = next;
        } else {
            Step s{}; s.kind = 1;
            const uint32_t first = (cur & ~kBvhLeafBit) >> 3, count = (cur & 7u) + 1u;
            s.nf = (uint8_t)count; s.id = first;
            for (uint32_t k = 0; k < count; k++) s.stages[k] = (uint8_t)test_face(bvh.leaf_faces[first + k], O, D, have, best_t, best_idx);
            rt.steps.push_back(s);
            if (stack.empty()) break;
            cur = stack.back(); stack.pop_back();
        }
    }
    return rt;
}

// wave-level replay.  Each lane has a cursor into its current ray's step list.
struct Lane { const RayTrace *r = nullptr; size_t pc = 0; bool active() const { return r && pc < r->steps.size(); } };

static long leaf_phase_cost(std::vector<Lane> &L)
{
    // lanes at a leaf run it together: per face slot, stage costs are paid while any lane is still in that stage
    int maxf = 0;
    for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 1) maxf = std::max(maxf, (int)l.r->steps[l.pc].nf);
    long c = 0;
    for (int f = 0; f < maxf; f++) {
        int deepest = 0;
        for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 1 && f < l.r->steps[l.pc].nf) deepest = std::max(deepest, (int)l.r->steps[l.pc].stages[f]);
        for (int s = 1; s <= deepest; s++) c += g_cost_stage[s];
        c += 6;  // index load, loop
    }
    for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 1) l.pc++;
    return c + 8;  // pop
}

// while-while until every lane is done (no refill); returns wave instructions
static long run_wave_static(std::vector<Lane> &L, long *lane_steps = nullptr)
{
    long cost = 0;
    auto any_active = [&] { for (auto &l : L) if (l.active()) return true; return false; };
    while (any_active()) {
        for (;;) {  // inner phase: as long as some lane sits on an inner node
            bool any_inner = false;
            for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 0) { any_inner = true; l.pc++; if (lane_steps) (*lane_steps)++; }
            if (!any_inner) break;
            cost += g_cost_inner;
        }
        bool any_leaf = false;
        for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 1) any_leaf = true;
        if (any_leaf) cost += leaf_phase_cost(L);
    }
    return cost;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: sim_bvh prefix [leaf] [layout: 0 = 8x8 tiles in 32x8 blocks, 1 = 32x4 p2 tiles]\n"); return 1; }
    const std::string pre = argv[1];
    const uint32_t max_leaf = argc > 2 ? atoi(argv[2]) : 2;
    const int layout = argc > 3 ? atoi(argv[3]) : 0;
    FILE *f = fopen((pre + "_rays.bin").c_str(), "rb");
    int32_t hdr[4];
    if (!f || fread(hdr, 4, 4, f) != 4) return 1;
    const int rows = hdr[0], w = hdr[1], ntri = hdr[2], S = hdr[3];
    std::vector<uint8_t> hit((size_t)rows * w);
    std::vector<float> rays((size_t)S * rows * w * 6);
    if (fread(hit.data(), 1, hit.size(), f) != hit.size() || fread(rays.data(), 4, rays.size(), f) != rays.size()) return 1;
    fclose(f);
    std::vector<float> tris((size_t)ntri * 9);
    f = fopen((pre + "_tris.bin").c_str(), "rb");
    if (!f || fread(tris.data(), 4, tris.size(), f) != tris.size()) return 1;
    fclose(f);
    g_tris.resize(ntri);
    for (int i = 0; i < ntri; i++) {
        Tri &T = g_tris[i];
        memcpy(T.p0, &tris[9 * i], 12); memcpy(T.p1, &tris[9 * i + 3], 12); memcpy(T.p2, &tris[9 * i + 6], 12);
        float a[3], b[3];
        for (int k = 0; k < 3; k++) { a[k] = T.p1[k] - T.p0[k]; b[k] = T.p2[k] - T.p0[k]; }
        cross(a, b, T.N);
        T.d = -dot(T.N, T.p0);
    }
    const Bvh bvh = build_bvh(tris.data(), ntri, max_leaf);
    printf("%s: %d x %d px, %d tris, leaf %u: %zu nodes, depth %u\n", pre.c_str(), w, rows, ntri, max_leaf, bvh.nodes.size(), bvh.max_depth);

    // queue order: workgroup by workgroup (compacted: only hits), as the primary stage emits them
    std::vector<std::vector<RayTrace>> segs;  // per workgroup
    const int bw = layout ? 64 : 32, bh = 8;
    for (int by = 0; by < rows; by += bh)
        for (int bx = 0; bx < w; bx += bw) {
            std::vector<RayTrace> seg;
            auto add = [&](int x, int y) {
                if (x >= w || y >= rows) return;
                const size_t i = (size_t)y * w + x;
                if (hit[i]) seg.push_back(trace(bvh, &rays[6 * i], &rays[6 * i + 3]));
            };
            if (!layout) {
                for (int wave = 0; wave < 4; wave++)
                    for (int lane = 0; lane < 64; lane++) add(bx + wave * 8 + (lane & 7), by + (lane >> 3));
            } else {
                for (int wave = 0; wave < 4; wave++)
                    for (int k = 0; k < 2; k++)  // pixel 0 of every lane, then pixel 1
                        for (int lane = 0; lane < 64; lane++) add(bx + (wave & 1) * 32 + 2 * (lane & 15) + k, by + (wave >> 1) * 4 + (lane >> 4));
            }
            segs.push_back(std::move(seg));
        }
    size_t n_rays = 0; long inner = 0, leafv = 0, faces = 0;
    std::vector<int> hist(64, 0);
    for (auto &s : segs) for (auto &r : s) {
        n_rays++;
        int ni = 0;
        for (auto &st : r.steps) { if (st.kind == 0) { inner++; ni++; } else { leafv++; faces += st.nf; } }
        hist[std::min(ni, 63)]++;
    }
    printf("rays %zu: inner visits/ray %.2f, leaf visits/ray %.2f, face tests/ray %.2f\n", n_rays, (double)inner / n_rays, (double)leafv / n_rays, (double)faces / n_rays);
    printf("inner-visit histogram:"); for (int i = 0; i < 40; i++) printf(" %d:%.1f%%", i, 100.0 * hist[i] / n_rays); printf("\n");
    // ideal cost (100 % lane utilisation): each ray alone, divided by 64
    {
        long c = 0;
        for (auto &s : segs) for (auto &r : s) { std::vector<Lane> L(1); L[0].r = &r; c += run_wave_static(L); }
        printf("ideal (no divergence)          : %8.1f wave-instr per ray\n", (double)c / n_rays / 64.0);
    }
    // policy A / B: k rays per lane, in sequence
    for (int k : {1, 2, 4, 8, 16}) {
        long c = 0, inner_iters = 0;
        for (auto &s : segs) {
            for (size_t base = 0; base < s.size(); base += 64 * (size_t)k) {
                // lanes process their rays one after another: the wave re-converges only at ray boundaries in the real kernel
                // if written as a per-lane loop; here: each lane walks its k rays as one concatenated step list
                std::vector<RayTrace> cat(64);
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < k; j++) {
                        const size_t e = base + (size_t)j * 64 + lane;
                        if (e < s.size()) { cat[lane].steps.insert(cat[lane].steps.end(), s[e].steps.begin(), s[e].steps.end()); }
                    }
                std::vector<Lane> L(64);
                for (int lane = 0; lane < 64; lane++) L[lane].r = &cat[lane];
                c += run_wave_static(L, &inner_iters) + (long)k * 40;  // per-ray setup
            }
        }
        printf("static, %2d ray(s) per lane       : %8.1f wave-instr per ray\n", k, (double)c / n_rays);
    }
    // policy M: each iteration the wave executes ONE kind of step — the kind with more weighted pending lanes
    for (int bias : {100, 150, 200, 300}) {
        long c = 0; double act_in = 0, it_in = 0, act_lf = 0, it_lf = 0;
        for (auto &s : segs)
            for (size_t base = 0; base < s.size(); base += 64) {
                std::vector<Lane> L(64);
                for (int lane = 0; lane < 64 && base + lane < s.size(); lane++) L[lane].r = &s[base + lane];
                for (;;) {
                    int a = 0, b = 0;
                    for (auto &l : L) if (l.active()) { if (l.r->steps[l.pc].kind == 0) a++; else b++; }
                    if (!a && !b) break;
                    if (b == 0 || (a && a * bias >= b * 100)) {   // inner step (bias: an inner lane counts bias/100 leaf lanes)
                        for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 0) l.pc++;
                        c += g_cost_inner; act_in += a; it_in++;
                    } else { c += leaf_phase_cost(L); act_lf += b; it_lf++; }
                }
            }
        printf("majority policy, bias %3d        : %8.1f wave-instr per ray (inner iters: %.1f lanes active; leaf phases: %.1f lanes)\n", bias,
               (double)c / n_rays, act_in / it_in, act_lf / it_lf);
    }
    {   // breakdown of the while-while baseline
        long c_in = 0, c_lf = 0; double act_in = 0, it_in = 0, act_lf = 0, it_lf = 0;
        for (auto &s : segs)
            for (size_t base = 0; base < s.size(); base += 64) {
                std::vector<Lane> L(64);
                for (int lane = 0; lane < 64 && base + lane < s.size(); lane++) L[lane].r = &s[base + lane];
                auto any_active = [&] { for (auto &l : L) if (l.active()) return true; return false; };
                while (any_active()) {
                    for (;;) {
                        int a = 0;
                        for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 0) { a++; l.pc++; }
                        if (!a) break;
                        c_in += g_cost_inner; act_in += a; it_in++;
                    }
                    int b = 0;
                    for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 1) b++;
                    if (b) { c_lf += leaf_phase_cost(L); act_lf += b; it_lf++; }
                }
            }
        printf("while-while breakdown: inner %.1f + leaf %.1f wave-instr per ray; inner iterations run %.1f lanes, leaf phases %.1f lanes\n",
               (double)c_in / n_rays, (double)c_lf / n_rays, act_in / it_in, act_lf / it_lf);
    }
    // policy C: persistent wave over a pool (the workgroup's segment x S samples: emulate by concatenating S segments), refill at threshold
    for (int pool_segs : {1, 4, 8}) for (int thresh : {8, 16, 32}) {
        long c = 0;
        const int fetch_cost = 45;
        for (size_t sb = 0; sb < segs.size(); sb += pool_segs) {
            std::vector<const RayTrace *> pool;
            for (size_t s = sb; s < std::min(segs.size(), sb + pool_segs); s++) for (auto &r : segs[s]) pool.push_back(&r);
            // 4 waves share the pool: emulate one wave taking every 4th chunk -> simply run one wave over pool/4 rays, 4 times
            const size_t per_wave = (pool.size() + 3) / 4;
            for (int wv = 0; wv < 4; wv++) {
                size_t next = wv * per_wave; const size_t end = std::min(pool.size(), next + per_wave);
                std::vector<Lane> L(64);
                auto idle = [&] { int n = 0; for (auto &l : L) n += !l.active(); return n; };
                for (;;) {
                    if (next < end && idle() >= thresh) { for (auto &l : L) if (!l.active() && next < end) { l.r = pool[next++]; l.pc = 0; } c += fetch_cost; }
                    bool any = false; for (auto &l : L) any |= l.active();
                    if (!any) { if (next >= end) break; for (auto &l : L) if (!l.active() && next < end) { l.r = pool[next++]; l.pc = 0; } c += fetch_cost; continue; }
                    // one while-while round, leaving the inner loop when too many lanes have gone idle
                    for (;;) {
                        bool any_inner = false;
                        for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 0) { any_inner = true; l.pc++; }
                        if (!any_inner) break;
                        c += g_cost_inner + 2;
                        if (next < end && idle() >= thresh) break;
                    }
                    bool any_leaf = false;
                    for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 1) any_leaf = true;
                    if (any_leaf) c += leaf_phase_cost(L);
                }
            }
        }
        printf("refill, pool %d segs, threshold %2d : %8.1f wave-instr per ray\n", pool_segs, thresh, (double)c / n_rays);
    }
    // policy S: pool = all S samples of a 64x8-pixel tile; rays sorted by direction bin (nb x nb octahedral map), then 64 per wave
    if (S > 1) {
        const int tw = argc > 4 ? atoi(argv[4]) : 64, th = argc > 5 ? atoi(argv[5]) : 8;
        for (int nb : {1, 4, 8, 16, 32}) {
            long c = 0; size_t nr = 0; long pk_cost = 0, pk_n = 0, pk_f = 0, pk_w = 0; long pk_stage[5] = {0, 0, 0, 0, 0};
            for (int by = 0; by < rows; by += th)
                for (int bx = 0; bx < w; bx += tw) {
                    struct E { int bin; RayTrace r; };
                    std::vector<E> pool;
                    for (int sidx = 0; sidx < S; sidx++)
                        for (int y = by; y < std::min(rows, by + th); y++)
                            for (int x = bx; x < std::min(w, bx + tw); x++) {
                                const size_t i = (size_t)y * w + x;
                                if (!hit[i]) continue;
                                const float *R = &rays[((size_t)sidx * rows * w + i) * 6];
                                const float *D = R + 3;
                                const float l1 = std::fabs(D[0]) + std::fabs(D[1]) + std::fabs(D[2]);
                                float u = D[0] / l1, v = D[1] / l1;
                                if (D[2] < 0) { const float uu = (1 - std::fabs(v)) * (u >= 0 ? 1 : -1), vv = (1 - std::fabs(u)) * (v >= 0 ? 1 : -1); u = uu; v = vv; }
                                const int iu = std::min(nb - 1, (int)((u * 0.5f + 0.5f) * nb)), iv = std::min(nb - 1, (int)((v * 0.5f + 0.5f) * nb));
                                pool.push_back({iv * nb + iu, trace(bvh, R, D)});
                            }
                    std::stable_sort(pool.begin(), pool.end(), [](const E &a, const E &b) { return a.bin < b.bin; });
                    nr += pool.size();
                    for (size_t base = 0; base < pool.size(); base += 128) {   // packet of 128 rays (2 per lane): union of nodes / faces
                        std::vector<uint8_t> seen_n(bvh.nodes.size(), 0), seen_f(bvh.leaf_faces.size() + 8, 0);
                        for (size_t e = base; e < std::min(pool.size(), base + 128); e++)
                            for (auto &st : pool[e].r.steps) {
                                if (st.kind == 0) seen_n[st.id] = 1;
                                else for (int k = 0; k < st.nf; k++) seen_f[st.id + k] = std::max<uint8_t>(seen_f[st.id + k], st.stages[k]);
                            }
                        long un = 0, uf = 0;
                        for (auto v : seen_n) un += v;
                        for (auto v : seen_f) { uf += v ? 1 : 0; pk_stage[v]++; }
                        pk_cost += un * 130 + uf * 125 + 100; pk_n += un; pk_f += uf; pk_w++;
                    }
                    for (size_t base = 0; base < pool.size(); base += 64) {
                        std::vector<Lane> L(64);
                        for (int lane = 0; lane < 64 && base + lane < pool.size(); lane++) L[lane].r = &pool[base + lane].r;
                        c += run_wave_static(L) + 40;
                    }
                }
            printf("   faces per packet by the deepest stage any ray reached (1 plane .. 4 all edges): %.2f %.2f %.2f %.2f\n",
                   (double)pk_stage[1] / pk_w, (double)pk_stage[2] / pk_w, (double)pk_stage[3] / pk_w, (double)pk_stage[4] / pk_w);
            printf("pool of %d samples x tile sorted into %3d direction bins : %8.1f wave-instr per ray; packets of 128: %.1f nodes + %.1f faces -> %.1f wave-instr per ray\n",
                   S, nb * nb, (double)c / nr, (double)pk_n / pk_w, (double)pk_f / pk_w, (double)pk_cost / nr);
        }
    }
    // policy Q: pool = tile x S samples as in policy S, sort key = direction bin combined with the QUADRANT of the tile the ray
    // starts in (32x4 pixels: the wave that emitted it) — origin-major, direction-major, or direction alone
    if (S > 1) {
        const int tw = 64, th = 8;
        for (int mode = 0; mode < 3; mode++) for (int nb : {8, 16}) {
            long c = 0; size_t nr = 0;
            for (int by = 0; by < rows; by += th)
                for (int bx = 0; bx < w; bx += tw) {
                    struct E { int key; RayTrace r; };
                    std::vector<E> pool;
                    for (int sidx = 0; sidx < S; sidx++)
                        for (int y = by; y < std::min(rows, by + th); y++)
                            for (int x = bx; x < std::min(w, bx + tw); x++) {
                                const size_t i = (size_t)y * w + x;
                                if (!hit[i]) continue;
                                const float *R = &rays[((size_t)sidx * rows * w + i) * 6];
                                const float *D = R + 3;
                                const int o = (D[0] < 0) | ((D[1] < 0) << 1) | ((D[2] < 0) << 2);
                                const float l1 = std::fabs(D[0]) + std::fabs(D[1]) + std::fabs(D[2]);
                                const int iu = std::min(nb - 1, (int)(std::fabs(D[0]) / l1 * nb)), iv = std::min(nb - 1, (int)(std::fabs(D[1]) / l1 * nb));
                                const int dir = (o * nb + iv) * nb + iu, quad = ((x - bx) / 32) + 2 * ((y - by) / 4);
                                const int key = mode == 0 ? dir : mode == 1 ? quad * 8 * nb * nb + dir : dir * 4 + quad;
                                pool.push_back({key, trace(bvh, R, D)});
                            }
                    std::stable_sort(pool.begin(), pool.end(), [](const E &a, const E &b) { return a.key < b.key; });
                    nr += pool.size();
                    for (size_t base = 0; base < pool.size(); base += 64) {
                        std::vector<Lane> L(64);
                        for (int lane = 0; lane < 64 && base + lane < pool.size(); lane++) L[lane].r = &pool[base + lane].r;
                        c += run_wave_static(L) + 40;
                    }
                }
            printf("per-lane, 64x8 tile pools, key %s, %2dx%2d cells per octant: %8.1f wave-instr per ray\n",
                   mode == 0 ? "direction          " : mode == 1 ? "quadrant, direction" : "direction, quadrant", nb, nb, (double)c / nr);
        }
    }
    // policy R: the direction-sorted tile pool (policy Q, direction key) walked by a PERSISTENT wave that refills idle lanes from
    // the sorted list when at least T of them are idle; a refill event costs the fetch of the new rays AND the epilogue of the
    // finished ones (shading + accumulation: 210), the static schedule pays 190 per 64 rays
    if (S > 1) {
        const int tw = 64, th = 8, nb = 16;
        for (int thresh : {0, 8, 16, 24, 32, 48}) {
            long c = 0; size_t nr = 0;
            for (int by = 0; by < rows; by += th)
                for (int bx = 0; bx < w; bx += tw) {
                    struct E { int key; RayTrace r; };
                    std::vector<E> pool;
                    for (int sidx = 0; sidx < S; sidx++)
                        for (int y = by; y < std::min(rows, by + th); y++)
                            for (int x = bx; x < std::min(w, bx + tw); x++) {
                                const size_t i = (size_t)y * w + x;
                                if (!hit[i]) continue;
                                const float *R = &rays[((size_t)sidx * rows * w + i) * 6];
                                const float *D = R + 3;
                                const int o = (D[0] < 0) | ((D[1] < 0) << 1) | ((D[2] < 0) << 2);
                                const float l1 = std::fabs(D[0]) + std::fabs(D[1]) + std::fabs(D[2]);
                                const int iu = std::min(nb - 1, (int)(std::fabs(D[0]) / l1 * nb)), iv = std::min(nb - 1, (int)(std::fabs(D[1]) / l1 * nb));
                                pool.push_back({(o * nb + iv) * nb + iu, trace(bvh, R, D)});
                            }
                    std::stable_sort(pool.begin(), pool.end(), [](const E &a, const E &b) { return a.key < b.key; });
                    nr += pool.size();
                    if (thresh == 0) {   // static: 64 rays, all to the end
                        for (size_t base = 0; base < pool.size(); base += 64) {
                            std::vector<Lane> L(64);
                            for (int lane = 0; lane < 64 && base + lane < pool.size(); lane++) L[lane].r = &pool[base + lane].r;
                            c += run_wave_static(L) + 190;
                        }
                        continue;
                    }
                    std::vector<Lane> L(64);
                    size_t next = 0;
                    auto idle = [&] { int n = 0; for (auto &l : L) if (!l.active()) n++; return n; };
                    for (;;) {
                        if (next < pool.size() && idle() >= thresh) { for (auto &l : L) if (!l.active() && next < pool.size()) { l.r = &pool[next++].r; l.pc = 0; } c += 210; }
                        bool any = false; for (auto &l : L) if (l.active()) any = true;
                        if (!any) { if (next >= pool.size()) break; continue; }
                        // one while-while round: inner steps until no lane sits on an inner node or enough lanes went idle, then leaves
                        for (;;) {
                            bool any_inner = false;
                            for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 0) { any_inner = true; l.pc++; }
                            if (!any_inner) break;
                            c += g_cost_inner;
                            if (next < pool.size() && idle() >= thresh) break;
                        }
                        bool any_leaf = false;
                        for (auto &l : L) if (l.active() && l.r->steps[l.pc].kind == 1) any_leaf = true;
                        if (any_leaf) c += leaf_phase_cost(L);
                    }
                    c += 150;   // the last rays' epilogue
                }
            printf("sorted tile pools, per-lane, %s: %8.1f wave-instr per ray (with ray fetch and epilogue)\n",
                   thresh == 0 ? "static 64 rays per wave     " : (std::string("refill at ") + std::to_string(thresh) + " idle lanes     ").c_str(), (double)c / nr);
        }
    }
    // policy D: ray STREAM traversal — per pool (tile x S samples), octant and BATCH of B direction-sorted rays, every BVH node / leaf
    // face is visited ONCE with the list of the batch's rays that reach it (wave-uniform node, rays 64 at a time, compaction into
    // the children's lists); priced per chunk of 64 rays: node 80, face 100 wave instructions (+ list upkeep); rays visit what they
    // visit in their own traversal
    if (S > 1) {
        const int tw = argc > 4 ? atoi(argv[4]) : 64, th = argc > 5 ? atoi(argv[5]) : 8;
        for (int B : {256, 512, 1024, 1 << 20}) {
            long cost = 0, full = 0; size_t nr = 0; double fill = 0; size_t max_arena = 0;
            for (int by = 0; by < rows; by += th)
                for (int bx = 0; bx < w; bx += tw) {
                    struct E { int key; RayTrace r; };
                    std::vector<E> oct[8];
                    for (int sidx = 0; sidx < S; sidx++)
                        for (int y = by; y < std::min(rows, by + th); y++)
                            for (int x = bx; x < std::min(w, bx + tw); x++) {
                                const size_t i = (size_t)y * w + x;
                                if (!hit[i]) continue;
                                const float *R = &rays[((size_t)sidx * rows * w + i) * 6];
                                const float *D = R + 3;
                                const int o = (D[0] < 0) | ((D[1] < 0) << 1) | ((D[2] < 0) << 2);
                                const float l1 = std::fabs(D[0]) + std::fabs(D[1]) + std::fabs(D[2]);
                                const int iu = std::min(15, (int)(std::fabs(D[0]) / l1 * 16)), iv = std::min(15, (int)(std::fabs(D[1]) / l1 * 16));
                                oct[o].push_back({iv * 16 + iu, trace(bvh, R, D)});
                                nr++;
                            }
                    for (int o = 0; o < 8; o++) {
                        std::stable_sort(oct[o].begin(), oct[o].end(), [](const E &a, const E &b) { return a.key < b.key; });
                        for (size_t base = 0; base < oct[o].size(); base += (size_t)B) {
                            std::vector<uint32_t> cn(bvh.nodes.size(), 0), cf(bvh.leaf_faces.size() + 8, 0);
                            size_t visits = 0;
                            for (size_t e = base; e < std::min(oct[o].size(), base + (size_t)B); e++)
                                for (auto &st : oct[o][e].r.steps) {
                                    if (st.kind == 0) { cn[st.id]++; visits++; }
                                    else for (int k = 0; k < st.nf; k++) cf[st.id + k]++;
                                }
                            max_arena = std::max(max_arena, visits);
                            for (auto n : cn) if (n) { const long c = (n + 63) / 64; cost += c * 80; full += c; fill += n; }
                            for (auto n : cf) if (n) { const long c = (n + 63) / 64; cost += c * 100; full += c; fill += n; }
                        }
                    }
                }
            printf("ray stream, %dx%d tiles x %d samples, batches of %7d rays of an octant: %8.1f wave-instr per ray (+ set-up), mean chunk fill %.0f %%, most list entries a batch makes in all %zu\n",
                   tw, th, S, B, (double)cost / nr, 100.0 * fill / ((double)full * 64), max_arena);
        }
    }
    // policy F: pool = all rays of the band that START ON THE SAME FACE (whatever tile their pixel is in), sorted by direction bin
    if (S > 1) {
        std::vector<int32_t> objv((size_t)rows * w, -1);
        if (FILE *fo = fopen((pre + "_obj.bin").c_str(), "rb")) {
            if (fread(objv.data(), 4, objv.size(), fo) != objv.size()) fprintf(stderr, "short obj file\n");
            fclose(fo);
            const int group = argc > 6 ? atoi(argv[6]) : 1;   // faces per pool (consecutive face indices: same instance, roughly neighbours)
            for (int nb : {4, 8, 16}) {
                struct E { int bin; RayTrace r; };
                std::vector<std::vector<E>> pools((ntri + group - 1) / group);
                for (int sidx = 0; sidx < S; sidx++)
                    for (size_t i = 0; i < (size_t)rows * w; i++) {
                        if (!hit[i] || objv[i] < 0) continue;
                        const float *R = &rays[((size_t)sidx * rows * w + i) * 6];
                        const float *D = R + 3;
                        const float l1 = std::fabs(D[0]) + std::fabs(D[1]) + std::fabs(D[2]);
                        float u = D[0] / l1, v = D[1] / l1;
                        if (D[2] < 0) { const float uu = (1 - std::fabs(v)) * (u >= 0 ? 1 : -1), vv = (1 - std::fabs(u)) * (v >= 0 ? 1 : -1); u = uu; v = vv; }
                        const int iu = std::min(nb - 1, (int)((u * 0.5f + 0.5f) * nb)), iv = std::min(nb - 1, (int)((v * 0.5f + 0.5f) * nb));
                        pools[objv[i] / group].push_back({iv * nb + iu, trace(bvh, R, D)});
                    }
                long c = 0, pk_cost = 0, pk_n = 0, pk_f = 0, pk_w = 0; size_t nr = 0, npools = 0;
                for (auto &pool : pools) {
                    if (pool.empty()) continue;
                    npools++;
                    std::stable_sort(pool.begin(), pool.end(), [](const E &a, const E &b) { return a.bin < b.bin; });
                    nr += pool.size();
                    for (size_t base = 0; base < pool.size(); base += 128) {
                        std::vector<uint8_t> seen_n(bvh.nodes.size(), 0), seen_f(bvh.leaf_faces.size() + 8, 0);
                        for (size_t e = base; e < std::min(pool.size(), base + 128); e++)
                            for (auto &st : pool[e].r.steps) {
                                if (st.kind == 0) seen_n[st.id] = 1;
                                else for (int k = 0; k < st.nf; k++) seen_f[st.id + k] = std::max<uint8_t>(seen_f[st.id + k], st.stages[k]);
                            }
                        long un = 0, uf = 0;
                        for (auto v : seen_n) un += v;
                        for (auto v : seen_f) uf += v ? 1 : 0;
                        pk_cost += un * 130 + uf * 125 + 100; pk_n += un; pk_f += uf; pk_w++;
                    }
                    for (size_t base = 0; base < pool.size(); base += 64) {
                        std::vector<Lane> L(64);
                        for (int lane = 0; lane < 64 && base + lane < pool.size(); lane++) L[lane].r = &pool[base + lane].r;
                        c += run_wave_static(L) + 40;
                    }
                }
                printf("pools by origin face (%d faces per pool: %zu pools, %.0f rays each) in %3d direction bins: per-lane %8.1f wave-instr per ray; packets of 128: %.1f nodes + %.1f faces -> %.1f wave-instr per ray\n",
                       group, npools, (double)nr / npools, nb * nb, (double)c / nr, (double)pk_n / pk_w, (double)pk_f / pk_w, (double)pk_cost / nr);
            }
        }
    }
    return 0;
}
