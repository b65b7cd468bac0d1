// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_core.h header).
// extern "C" surface of the CPU oracle, loaded with ctypes by tests/, smoke()
// and bench.py's cpu_baseline leg.  Nothing under pbrt-rs_amd/ may use it.
#include <atomic>
#include <thread>

#include "ref_host.h"
#include "ref_integrator.h"
#include "ref_render.h"
#include "ref_scene.h"
#include "wide_walk.h"

using namespace ora;

struct OraRay { float o[3]; float tmax; float d[3]; float time; };
struct OraHit { uint32_t prim_id; float t, u, v; };

struct OraScene {
    Scene sc;
    uint32_t next_id = 0;
};

static M4 m4_from(const float* p) {
    M4 m;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m.m[r][c] = p[r * 4 + c];
    return m;
}

template <class F>
static void parallel_chunks(uint64_t n, int nthreads, F f) {
    if (nthreads <= 1 || n < 4096) { f(0, n, 0); return; }
    std::vector<std::thread> th;
    std::atomic<uint64_t> next(0);
    const uint64_t chunk = 4096;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            for (;;) {
                uint64_t b = next.fetch_add(chunk);
                if (b >= n) break;
                f(b, std::min(n, b + chunk), t);
            }
        });
    for (auto& t : th) t.join();
}

static void fill_hit(const Scene& sc, const Hit& h, OraHit* o) {
    o->prim_id = h.prim_id; o->t = h.t; o->u = h.u; o->v = h.v;
    const Prim& p = sc.prims[h.prim_slot];
    if (p.type == PRIM_SPHERE) {   // sphere.rs:382-385
        const Sphere& s = sc.spheres[p.index];
        o->u = h.phi / s.phi_max;
        float theta = safe_acos(h.p_obj.z / s.radius);
        o->v = (theta - s.theta_z_min) / (s.theta_z_max - s.theta_z_min);
    }
}

extern "C" {

OraScene* ora_scene_create() { return new OraScene(); }
void ora_scene_destroy(OraScene* s) { delete s; }

uint32_t ora_add_sphere(OraScene* s, float radius, float z_min, float z_max, float phi_max_deg,
                        const float* rfo, const float* ofr, int reverse) {
    s->sc.spheres.push_back(make_sphere(radius, z_min, z_max, phi_max_deg, m4_from(rfo), m4_from(ofr), reverse != 0));
    Prim p{PRIM_SPHERE, (uint32_t)s->sc.spheres.size() - 1, s->next_id++};
    s->sc.prims.push_back(p);
    return p.id;
}
// positions are already in render space (BilinearPatchMesh::new transforms them, bilinear_patch.rs:649-655)
uint32_t ora_add_patch_mesh_n(OraScene* s, const float* pos, uint32_t n_vert, const uint32_t* idx4, uint32_t n_patch,
                              const float* normals, int reverse_orientation, int transform_swaps_handedness);
uint32_t ora_add_patch_mesh_uv(OraScene* s, const float* pos, uint32_t n_vert, const uint32_t* idx4, uint32_t n_patch,
                               const float* normals, const float* uv, int reverse_orientation, int transform_swaps_handedness);
uint32_t ora_add_patch_mesh(OraScene* s, const float* pos, uint32_t n_vert, const uint32_t* idx4, uint32_t n_patch,
                            int reverse_orientation, int transform_swaps_handedness) {
    return ora_add_patch_mesh_n(s, pos, n_vert, idx4, n_patch, nullptr, reverse_orientation, transform_swaps_handedness);
}
// `normals`: one per vertex, already in render space and already negated for reverse_orientation, as
// BilinearPatchMesh::new leaves them (bilinear_patch.rs:660-671); may be null
uint32_t ora_add_patch_mesh_n(OraScene* s, const float* pos, uint32_t n_vert, const uint32_t* idx4, uint32_t n_patch,
                              const float* normals, int reverse_orientation, int transform_swaps_handedness) {
    return ora_add_patch_mesh_uv(s, pos, n_vert, idx4, n_patch, normals, nullptr, reverse_orientation, transform_swaps_handedness);
}
// ... with per-vertex uv (BilinearPatchMesh.uv, bilinear_patch.rs:621-622; two floats per vertex, may be null)
uint32_t ora_add_patch_mesh_uv(OraScene* s, const float* pos, uint32_t n_vert, const uint32_t* idx4, uint32_t n_patch,
                               const float* normals, const float* uv, int reverse_orientation, int transform_swaps_handedness) {
    (void)n_vert;
    uint32_t first = s->next_id;
    for (uint32_t i = 0; i < n_patch; ++i) {
        Patch pt;
        auto P = [&](uint32_t k) { return V3(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]); };
        pt.p00 = P(idx4[4 * i]); pt.p10 = P(idx4[4 * i + 1]); pt.p01 = P(idx4[4 * i + 2]); pt.p11 = P(idx4[4 * i + 3]);
        pt.flip_normal = (reverse_orientation != 0) ^ (transform_swaps_handedness != 0);
        if (normals) {
            auto N = [&](uint32_t k) { return V3(normals[3 * k], normals[3 * k + 1], normals[3 * k + 2]); };
            pt.has_normals = true;
            pt.n00 = N(idx4[4 * i]); pt.n10 = N(idx4[4 * i + 1]); pt.n01 = N(idx4[4 * i + 2]); pt.n11 = N(idx4[4 * i + 3]);
        }
        if (uv) {
            pt.has_uv = true;
            for (int k = 0; k < 2; ++k) {
                pt.uv00[k] = uv[2 * idx4[4 * i] + k]; pt.uv10[k] = uv[2 * idx4[4 * i + 1] + k];
                pt.uv01[k] = uv[2 * idx4[4 * i + 2] + k]; pt.uv11[k] = uv[2 * idx4[4 * i + 3] + k];
            }
        }
        s->sc.patches.push_back(pt);
        s->sc.prims.push_back(Prim{PRIM_PATCH, (uint32_t)s->sc.patches.size() - 1, s->next_id++});
    }
    return first;
}
// triangle (v0,v1,v2) == degenerate patch (v0,v1,v2,v2)
uint32_t ora_add_triangle_mesh(OraScene* s, const float* pos, uint32_t n_vert, const uint32_t* idx3, uint32_t n_tri) {
    (void)n_vert;
    uint32_t first = s->next_id;
    s->sc.patches.reserve(s->sc.patches.size() + n_tri);
    s->sc.prims.reserve(s->sc.prims.size() + n_tri);
    for (uint32_t i = 0; i < n_tri; ++i) {
        Patch pt;
        auto P = [&](uint32_t k) { return V3(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]); };
        pt.p00 = P(idx3[3 * i]); pt.p10 = P(idx3[3 * i + 1]); pt.p01 = P(idx3[3 * i + 2]); pt.p11 = pt.p01;
        pt.flip_normal = false;
        s->sc.patches.push_back(pt);
        s->sc.prims.push_back(Prim{PRIM_PATCH, (uint32_t)s->sc.patches.size() - 1, s->next_id++});
    }
    return first;
}
int ora_build(OraScene* s, int method, int max_prims_in_node) {
    s->sc.build((SplitMethod)method, max_prims_in_node);
    return (int)s->sc.nodes.size();
}
void ora_scene_bounds(OraScene* s, float* out6) {
    B3 b = s->sc.bounds();
    out6[0] = b.lo.x; out6[1] = b.lo.y; out6[2] = b.lo.z; out6[3] = b.hi.x; out6[4] = b.hi.y; out6[5] = b.hi.z;
}
// mode 0: reference BVH traversal; mode 1: brute force over all primitives
void ora_trace_closest(OraScene* s, const OraRay* rays, uint64_t n, OraHit* out, uint64_t* counters2, int nthreads,
                       int mode) {
    std::vector<Counters> ctr(std::max(1, nthreads));
    parallel_chunks(n, nthreads, [&](uint64_t b, uint64_t e, int tid) {
        for (uint64_t i = b; i < e; ++i) {
            const OraRay& r = rays[i];
            Hit h;
            V3 ro(r.o[0], r.o[1], r.o[2]), rd(r.d[0], r.d[1], r.d[2]);
            bool ok = mode == 0 ? s->sc.intersect(ro, rd, r.tmax, &h, &ctr[tid]) : s->sc.brute_force(ro, rd, r.tmax, &h);
            if (ok) fill_hit(s->sc, h, &out[i]);
            else { out[i].prim_id = 0xFFFFFFFFu; out[i].t = r.tmax; out[i].u = 0; out[i].v = 0; }
        }
    });
    if (counters2) {
        counters2[0] = counters2[1] = 0;
        for (auto& c : ctr) { counters2[0] += c.nodes_visited; counters2[1] += c.prims_tested; }
    }
}
void ora_trace_any(OraScene* s, const OraRay* rays, uint64_t n, uint8_t* out, uint64_t* counters2, int nthreads) {
    std::vector<Counters> ctr(std::max(1, nthreads));
    parallel_chunks(n, nthreads, [&](uint64_t b, uint64_t e, int tid) {
        for (uint64_t i = b; i < e; ++i) {
            const OraRay& r = rays[i];
            out[i] = s->sc.intersect_p(V3(r.o[0], r.o[1], r.o[2]), V3(r.d[0], r.d[1], r.d[2]), r.tmax, &ctr[tid]) ? 1 : 0;
        }
    });
    if (counters2) {
        counters2[0] = counters2[1] = 0;
        for (auto& c : ctr) { counters2[0] += c.nodes_visited; counters2[1] += c.prims_tested; }
    }
}

// Walk the GPU-layout wide BVH on the CPU (same order as the HIP kernel).  any_hit: out is uint8[n], else OraHit[n].
// flat != 0: the scene is swept without the tree, as the kernels do for PmAccelInfo.flat_prims != 0.
void ora_wide_trace(const void* nodes, const void* prims, const OraRay* rays, uint64_t n, void* out, int any_hit,
                    uint64_t* counters4, int nthreads, int flat) {
    WideWalker ww{reinterpret_cast<const WNode*>(nodes), reinterpret_cast<const uint8_t*>(prims)};
    WideWalker::FlatTable ft;
    if (flat) ft = ww.flat_table();
    std::vector<Counters> ctr(std::max(1, nthreads));
    parallel_chunks(n, nthreads, [&](uint64_t b, uint64_t e, int tid) {
        for (uint64_t i = b; i < e; ++i) {
            const OraRay& r = rays[i];
            uint32_t id; float t, u, v;
            const V3 ro(r.o[0], r.o[1], r.o[2]), rd(r.d[0], r.d[1], r.d[2]);
            bool f = flat ? ww.walk_flat(ft, ro, rd, r.tmax, any_hit != 0, &id, &t, &u, &v, &ctr[tid])
                          : ww.walk(ro, rd, r.tmax, any_hit != 0, &id, &t, &u, &v, &ctr[tid]);
            if (any_hit) reinterpret_cast<uint8_t*>(out)[i] = f ? 1 : 0;
            else { OraHit& h = reinterpret_cast<OraHit*>(out)[i]; h.prim_id = id; h.t = t; h.u = u; h.v = v; }
        }
    });
    if (counters4) {
        counters4[0] = counters4[1] = counters4[2] = counters4[3] = 0;
        for (auto& c : ctr) {
            counters4[0] += c.nodes_visited; counters4[1] += c.prims_tested; counters4[2] += c.exact_tests; counters4[3] += c.fallback_walks;
        }
    }
}

// Lock-step model of one 64-lane wave running the kernel's loop over 64 consecutive rays: iteration i of the wave costs a
// node step if any lane does one, plus max-over-lanes primitive (pre-filter) tests.  out6 = {waves, wave iterations,
// wave node steps, wave prim trips, lane node steps, lane prim tests}.  Analysis tool only.
void ora_wide_wave_model(const void* nodes, const void* prims, const OraRay* rays, uint64_t n, int any_hit, uint64_t* out6, int nthreads) {
    WideWalker ww{reinterpret_cast<const WNode*>(nodes), reinterpret_cast<const uint8_t*>(prims)};
    const uint64_t nw = (n + 63) / 64;
    std::vector<std::array<uint64_t, 6>> acc(std::max(1, nthreads), std::array<uint64_t, 6>{0, 0, 0, 0, 0, 0});
    parallel_chunks(nw, nthreads, [&](uint64_t b, uint64_t e, int tid) {
        std::vector<std::vector<uint8_t>> tr(64);
        for (uint64_t w = b; w < e; ++w) {
            size_t maxlen = 0;
            for (int l = 0; l < 64; ++l) {
                tr[l].clear();
                uint64_t i = w * 64 + l;
                if (i >= n) continue;
                const OraRay& r = rays[i];
                uint32_t id; float t, u, v;
                ww.walk(V3(r.o[0], r.o[1], r.o[2]), V3(r.d[0], r.d[1], r.d[2]), r.tmax, any_hit != 0, &id, &t, &u, &v, nullptr, &tr[l]);
                maxlen = std::max(maxlen, tr[l].size());
            }
            auto& a = acc[tid];
            a[0]++; a[1] += maxlen;
            for (size_t it = 0; it < maxlen; ++it) {
                bool node = false; uint32_t mp = 0;
                for (int l = 0; l < 64; ++l) {
                    if (it >= tr[l].size()) continue;
                    uint8_t x = tr[l][it];
                    if (x != 255) { node = true; mp = std::max<uint32_t>(mp, x); a[4]++; a[5] += x; }
                }
                if (node) a[2]++;
                a[3] += mp;
            }
        }
    });
    for (int k = 0; k < 6; ++k) { out6[k] = 0; for (auto& a : acc) out6[k] += a[k]; }
}

// Same model for PERSISTENT waves with batch refill: `n_waves` waves pull consecutive ray indices from one global counter;
// a wave refills its idle lanes when at least `thresh` of them are idle (or all are).  Pops are merged into the node step
// when merge_pop != 0.  out6 = {refills, wave iterations, wave node steps, wave prim trips, lane node steps, lane prim tests}.
void ora_wide_wave_model_persistent(const void* nodes, const void* prims, const OraRay* rays, uint64_t n, int any_hit,
                                    int n_waves, int thresh, int merge_pop, uint64_t* out6) {
    WideWalker ww{reinterpret_cast<const WNode*>(nodes), reinterpret_cast<const uint8_t*>(prims)};
    struct Lane { std::vector<uint8_t> tr; size_t pos = 0; bool active = false; };
    std::vector<std::vector<Lane>> waves(n_waves, std::vector<Lane>(64));
    uint64_t next = 0;
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    bool any_active = true;
    while (any_active) {
        any_active = false;
        for (auto& wv : waves) {
            int idle = 0;
            for (auto& l : wv) if (!l.active) idle++;
            if ((idle >= thresh || idle == 64) && next < n) {
                a[0]++;
                for (auto& l : wv) {
                    if (l.active || next >= n) continue;
                    const OraRay& r = rays[next++];
                    uint32_t id; float t, u, v;
                    l.tr.clear(); l.pos = 0;
                    ww.walk(V3(r.o[0], r.o[1], r.o[2]), V3(r.d[0], r.d[1], r.d[2]), r.tmax, any_hit != 0, &id, &t, &u, &v, nullptr, &l.tr);
                    if (merge_pop) { size_t k = 0; for (uint8_t x : l.tr) if (x != 255) l.tr[k++] = x; l.tr.resize(k); }
                    l.active = !l.tr.empty();
                }
            }
            bool node = false, act = false; uint32_t mp = 0;
            for (auto& l : wv) {
                if (!l.active) continue;
                act = true;
                uint8_t x = l.tr[l.pos++];
                if (x != 255) { node = true; mp = std::max<uint32_t>(mp, x); a[4]++; a[5] += x; }
                if (l.pos >= l.tr.size()) l.active = false;
            }
            if (act) { a[1]++; any_active = true; }
            if (node) a[2]++;
            a[3] += mp;
        }
        if (next < n) any_active = true;
    }
    for (int k = 0; k < 6; ++k) out6[k] = a[k];
}

// Model of a WAVE-PACKET walk (round 3 experiment): the 64 consecutive rays of a wave walk the tree
// TOGETHER - one wave-uniform stack of (node, lane mask) entries, the node read once, every lane of the mask running the slab
// test of the 8 children, a child entered when ANY lane hits it (the ballot), leaf triangles pre-filtered by the lanes that hit
// the leaf's box, children ordered by the octant of the wave's first ray.  Culling per lane as in walk() (certain hits of the
// triangle pre-filter shorten tcull).  Triangle-only trees.  out8 = {waves, packet node steps (the union of the lanes' node
// visits), packet prim trips, lane participations in node steps, lane prim tests, waves whose rays span more than one octant,
// deepest stack, packet node steps in which fewer than 16 lanes take part}.  Analysis tool only.
void ora_wide_packet_model(const void* nodes_, const void* prims_, const OraRay* rays, uint64_t n, uint64_t* out8, int nthreads) {
    const WNode* nodes = reinterpret_cast<const WNode*>(nodes_);
    const uint8_t* prims = reinterpret_cast<const uint8_t*>(prims_);
    const uint64_t nw = (n + 63) / 64;
    std::vector<std::array<uint64_t, 8>> acc(std::max(1, nthreads), std::array<uint64_t, 8>{0, 0, 0, 0, 0, 0, 0, 0});
    parallel_chunks(nw, nthreads, [&](uint64_t b, uint64_t e, int tid) {
        auto& a = acc[tid];
        struct LaneS { V3 ro, rd; float inv[3]; bool ng[3]; float R, D, tcull; bool live; };
        const float widen = 1.0f + 2.0f * gamma_n(3);
        const float MARGIN = 2.384185791015625e-07f;
        auto rcp = [](float d, float D) { return std::fabs(d) < 1e-18f * D ? std::copysign(1e18f / D, d) : 1.0f / d; };   // safe_rcp_dir
        std::vector<std::pair<uint32_t, uint64_t>> stack;
        for (uint64_t w = b; w < e; ++w) {
            LaneS L[64];
            uint64_t live = 0; uint32_t poct = 0; bool have = false, mixed = false;
            for (int l = 0; l < 64; ++l) {
                uint64_t i = w * 64 + l;
                L[l].live = false;
                if (i >= n) continue;
                const OraRay& r = rays[i];
                L[l].ro = V3(r.o[0], r.o[1], r.o[2]); L[l].rd = V3(r.d[0], r.d[1], r.d[2]);
                const float d3[3] = {r.d[0], r.d[1], r.d[2]};
                uint32_t oct = 0;
                const float Dl = WideWalker::maxabs(L[l].rd);
                for (int k = 0; k < 3; ++k) { L[l].inv[k] = rcp(d3[k], Dl); L[l].ng[k] = L[l].inv[k] < 0; oct |= L[l].ng[k] ? 1u << k : 0u; }
                L[l].R = WideWalker::maxabs(L[l].ro); L[l].D = WideWalker::maxabs(L[l].rd); L[l].tcull = r.tmax; L[l].live = true;
                live |= 1ull << l;
                if (!have) { poct = oct; have = true; } else if (oct != poct) mixed = true;
            }
            a[0]++; if (mixed) a[5]++;
            stack.clear();
            if (live) stack.emplace_back(0u, live);
            while (!stack.empty()) {
                a[6] = std::max<uint64_t>(a[6], stack.size());
                auto [nidx, mask] = stack.back(); stack.pop_back();
                const WNode& nd = nodes[nidx];
                a[1]++; a[3] += (uint64_t)__builtin_popcountll(mask);
                if (__builtin_popcountll(mask) < 16) a[7]++;
                uint64_t child_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int l = 0; l < 64; ++l) {
                    if (!(mask >> l & 1)) continue;
                    const LaneS& q = L[l];
                    const float o3[3] = {q.ro.x, q.ro.y, q.ro.z};
                    float si[3], an[3], af[3], sf[3];
                    for (int k = 0; k < 3; ++k) {
                        si[k] = u2f((uint32_t)nd.e[k] << 23) * q.inv[k];
                        float ad = (nd.o[k] - o3[k]) * q.inv[k];
                        float m = MARGIN * std::fmaf(256.0f, std::fabs(si[k]), std::fabs(ad));
                        an[k] = ad - m; af[k] = (ad + m) * widen; sf[k] = si[k] * widen;
                    }
                    for (int s = 0; s < 8; ++s) {
                        float tn[3], tf[3];
                        for (int k = 0; k < 3; ++k) {
                            float qn = (float)(q.ng[k] ? nd.qhi[k][s] : nd.qlo[k][s]), qf = (float)(q.ng[k] ? nd.qlo[k][s] : nd.qhi[k][s]);
                            tn[k] = std::fmaf(qn, si[k], an[k]);
                            tf[k] = std::fmaf(qf, sf[k], af[k]);
                        }
                        float nn = rmax(rmax(tn[0], tn[1]), rmax(tn[2], 0.0f));
                        float ff = rmin(rmin(tf[0], tf[1]), rmin(tf[2], q.tcull));
                        if (nn <= ff) child_mask[s] |= 1ull << l;
                    }
                }
                const uint32_t pbase = nd.prim_base & 0x3FFFFFFFu;
                // leaves of this node first (as the kernel does), nearest first by the packet's octant
                for (uint32_t k = 0; k < 8; ++k) {
                    uint32_t s = k ^ poct;
                    if ((nd.imask >> s & 1) || !child_mask[s]) continue;
                    uint32_t cnt = nd.meta[s] >> 5, off = nd.meta[s] & 31u;
                    for (uint32_t j = 0; j < cnt; ++j) {
                        a[2]++;
                        const float* f = reinterpret_cast<const float*>(prims + (size_t)(pbase + (off + j) * 3) * 16);
                        for (int l = 0; l < 64; ++l) {
                            if (!(child_mask[s] >> l & 1)) continue;
                            a[4]++;
                            float th, tl;
                            int cls = WideWalker::tri_prefilter(L[l].ro, L[l].rd, L[l].R, L[l].D, V3(f[0], f[1], f[2]), V3(f[3], f[4], f[5]),
                                                               V3(f[6], f[7], f[8]), L[l].tcull, &th, &tl);
                            if (cls == 2) L[l].tcull = th;
                        }
                    }
                }
                for (int k = 7; k >= 0; --k) {         // farthest pushed first
                    uint32_t s = (uint32_t)k ^ poct;
                    if (!(nd.imask >> s & 1) || !child_mask[s]) continue;
                    uint32_t cidx = nd.child_base + (uint32_t)__builtin_popcount(nd.imask & ((1u << s) - 1u));
                    stack.emplace_back(cidx, child_mask[s]);
                }
            }
        }
    });
    for (int k = 0; k < 8; ++k) {
        out8[k] = 0;
        for (auto& a : acc) { if (k == 6) out8[k] = std::max(out8[k], a[k]); else out8[k] += a[k]; }
    }
}

// ---- Integrate::render() on the CPU, from the flattened scene (same struct layouts as include/pbrt_mi355.h) ----
struct OraRenderer {
    Renderer r;
    std::vector<std::vector<float>> tables;   // owned copies of every table the description points to
    const float* own(const float* p, size_t n) { tables.emplace_back(p, p + n); return tables.back().data(); }
    uint64_t last_dead_final_rays = 0;
};
OraRenderer* ora_renderer_create(OraScene* s, const OMaterial* mats, uint32_t n_mats, const uint32_t* prim_mat, uint32_t n_prims,
                                 const OLight* lights, uint32_t n_lights, const OCamera* cam, const OFilm* film) {
    OraRenderer* o = new OraRenderer();
    o->tables.reserve(2 * n_mats + n_lights + 16);
    o->r.scene = &s->sc;
    o->r.materials.assign(mats, mats + n_mats);
    for (auto& m : o->r.materials) {
        if (m.spec.table) m.spec.table = o->own(m.spec.table, 471);
        if (m.texture != TEX_NONE && m.spec2.table) m.spec2.table = o->own(m.spec2.table, 471);
    }
    o->r.prim_material.assign(prim_mat, prim_mat + n_prims);
    o->r.lights.assign(lights, lights + n_lights);
    for (auto& l : o->r.lights) l.spectrum = o->own(l.spectrum, 471);
    o->r.cam = *cam;
    o->r.film = *film;
    o->r.film.sensor_bar = o->own(film->sensor_bar, 3 * 471);
    if (film->filter_vals) o->r.film.filter_vals = o->own(film->filter_vals, (size_t)film->filter_nx * film->filter_ny);
    o->r.prepare();
    return o;
}
void ora_renderer_destroy(OraRenderer* o) { delete o; }
// HomogeneousMedium (media/medium.rs:49-68) for INTEGRATOR_VOL_PATH; null tables remove it
void ora_renderer_set_medium(OraRenderer* o, const float* sigma_a471, const float* sigma_s471, float g, const float* bounds6) {
    o->r.have_medium = sigma_a471 && sigma_s471;
    if (!o->r.have_medium) return;
    o->r.medium.sigma_a = o->own(sigma_a471, 471); o->r.medium.sigma_s = o->own(sigma_s471, 471);
    o->r.medium.g = g;
    for (int k = 0; k < 6; ++k) o->r.medium.bounds[k] = bounds6[k];
}
void ora_renderer_render(OraRenderer* o, const ORender* rp, int nthreads, uint64_t* counters4) {
    RenderCounters rc = o->r.render(*rp, nthreads);
    if (counters4) { counters4[0] = rc.samples; counters4[1] = rc.closest_rays; counters4[2] = rc.shadow_rays; counters4[3] = rc.bounces; }
    o->last_dead_final_rays = rc.dead_final_rays;
}
uint64_t ora_renderer_dead_final_rays(OraRenderer* o) { return o->last_dead_final_rays; }   // of the last ora_renderer_render
void ora_renderer_clear(OraRenderer* o) { std::fill(o->r.pixels.begin(), o->r.pixels.end(), 0.0); }
void ora_renderer_film(OraRenderer* o, double* rgbw) { std::memcpy(rgbw, o->r.pixels.data(), o->r.pixels.size() * sizeof(double)); }
void ora_renderer_image(OraRenderer* o, float* rgb) { o->r.get_image(rgb); }
float ora_renderer_scene_radius(OraRenderer* o) { return o->r.scene_radius; }
// L[4], lambda[4], pdf[4], filter weight of one (pixel, sample) — no film add
void ora_renderer_sample(OraRenderer* o, const ORender* rp, int px, int py, int sample_index, float* out13) {
    RenderCounters rc; Spec4 L; Wavelengths w; float fw;
    o->r.eval_pixel_sample(px, py, sample_index, *rp, &rc, &L, &w, &fw);
    for (int i = 0; i < 4; ++i) { out13[i] = L.v[i]; out13[4 + i] = w.lambda[i]; out13[8 + i] = w.pdf[i]; }
    out13[12] = fw;
}
// debugging aid: the rays of ONE (pixel, sample) path, 10 floats each (ref_scene.h ray_log); returns the number of rays
int ora_renderer_path_log(OraRenderer* o, const ORender* rp, int px, int py, int sample_index, float* out, int max_rays) {
    std::vector<float> log;
    Scene::ray_log() = &log;
    RenderCounters rc; Spec4 L; Wavelengths w; float fw;
    o->r.eval_pixel_sample(px, py, sample_index, *rp, &rc, &L, &w, &fw);
    Scene::ray_log() = nullptr;
    const int n = (int)(log.size() / 10);
    for (int i = 0; i < n && i < max_rays; ++i) std::memcpy(out + 10 * i, log.data() + 10 * i, 40);
    return n;
}
// PixelSensor::to_sensor_rgb probe (camera/sensor.rs:114-121) for the golden vectors
void ora_to_sensor_rgb(const float* sensor_bar3x471, float imaging_ratio, const float* L4, const float* lambda4, const float* pdf4, float* rgb3) {
    for (int c = 0; c < 3; ++c) {
        float sum = 0.0f;
        for (int i = 0; i < 4; ++i) { float lp = pdf4[i] != 0.0f ? L4[i] / pdf4[i] : 0.0f; sum += dense_at(sensor_bar3x471 + 471 * c, lambda4[i]) * lp; }
        rgb3[c] = sum / 4.0f * imaging_ratio;
    }
}
float ora_dense_at(const float* table471, float lambda) { return dense_at(table471, lambda); }
float ora_sigmoid_poly(float c0, float c1, float c2, float lambda) { return sigmoid_poly(c0, c1, c2, lambda); }

// RGBToSpectrumTable::convert (color/color_space.rs:128-185) restated for the golden test of the regenerated table
void ora_rgb_table_convert(const float* z_nodes, const float* coeffs, const float* rgb_in, float* out3) {
    const int R = 64;
    float rgb[3];
    for (int i = 0; i < 3; ++i) rgb[i] = rgb_in[i] < 0.0f ? 0.0f : rgb_in[i];      // RGB::clamp_zero
    if (rgb[0] == rgb[1] && rgb[1] == rgb[2]) { out3[0] = 0; out3[1] = 0; out3[2] = (rgb[0] - 0.5f) / std::sqrt(rgb[0] * (1.0f - rgb[0])); return; }
    int max_i = rgb[0] > rgb[1] ? (rgb[0] > rgb[2] ? 0 : 2) : (rgb[1] > rgb[2] ? 1 : 2);
    float z = rgb[max_i];
    float x = rgb[(max_i + 1) % 3] * (float)(R - 1) / z, y = rgb[(max_i + 2) % 3] * (float)(R - 1) / z;
    int xi = std::min((int)x, R - 2), yi = std::min((int)y, R - 2);
    std::vector<float> zn(z_nodes, z_nodes + R);
    int left = 0, right = R - 1;
    while (left < right) { int mid = left + (right - left) / 2; if (zn[mid] < z) left = mid + 1; else right = mid; }
    int zi = left == 0 ? 0 : (left >= R - 1 ? R - 2 : left - 1);
    float dx = x - (float)xi, dy = y - (float)yi, dz = (z - zn[zi]) / (zn[zi + 1] - zn[zi]);
    for (int c = 0; c < 3; ++c) {
        auto co = [&](int a, int b, int d) { return coeffs[3 * ((((size_t)max_i * R + zi + d) * R + yi + b) * R + xi + a) + c]; };
        out3[c] = lerpf(lerpf(lerpf(co(0, 0, 0), co(1, 0, 0), dx), lerpf(co(0, 1, 0), co(1, 1, 0), dx), dy),
                        lerpf(lerpf(co(0, 0, 1), co(1, 0, 1), dx), lerpf(co(0, 1, 1), co(1, 1, 1), dx), dy), dz);
    }
}

// probes of the C5 extensions (media/phase_function.rs:57-80, shapes/bilinear_patch.rs:260-352)
float ora_henyey_greenstein(float cos_theta, float g) { return henyey_greenstein(cos_theta, g); }
void ora_sample_henyey_greenstein(const float* wo3, float g, const float* u2, float* wi3_pdf) {
    float pdf;
    V3 wi = sample_henyey_greenstein(V3(wo3[0], wo3[1], wo3[2]), g, P2{u2[0], u2[1]}, &pdf);
    wi3_pdf[0] = wi.x; wi3_pdf[1] = wi.y; wi3_pdf[2] = wi.z; wi3_pdf[3] = pdf;
}
// BilinearPatch::sample_with_context from ref_p: out7 = point, normal, solid-angle pdf; returns 0 when there is no sample
int ora_patch_sample_ctx(const float* p12, const float* ref_p3, const float* u2, float* out7) {
    Patch p; auto P = [&](int k) { return V3(p12[3 * k], p12[3 * k + 1], p12[3 * k + 2]); };
    p.p00 = P(0); p.p10 = P(1); p.p01 = P(2); p.p11 = P(3); p.flip_normal = false;
    ShapeSample ss;
    if (!patch_sample_ctx(p, V3(ref_p3[0], ref_p3[1], ref_p3[2]), P2{u2[0], u2[1]}, &ss)) return 0;
    V3 m = ss.pi.midpoints();
    out7[0] = m.x; out7[1] = m.y; out7[2] = m.z; out7[3] = ss.n.x; out7[4] = ss.n.y; out7[5] = ss.n.z; out7[6] = ss.pdf;
    return 1;
}
// Sphere::sample_with_context (ref_integrator.h sphere_sample_ctx): sphere = (radius, z_min, z_max, phi_max_deg, reverse) + the two
// 4 x 4 matrices; ctx_n3 may be null (a point in a medium: no offset).  The context's position error is zero here.
int ora_sphere_sample_ctx(const float* sph5, const float* rfo16, const float* ofr16, const float* ctx_p3, const float* ctx_n3,
                          const float* u2, float* out7) {
    M4 rfo, ofr;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { rfo.m[r][c] = rfo16[4 * r + c]; ofr.m[r][c] = ofr16[4 * r + c]; }
    Sphere sp = make_sphere(sph5[0], sph5[1], sph5[2], sph5[3], rfo, ofr, sph5[4] != 0.0f);
    V3 cp(ctx_p3[0], ctx_p3[1], ctx_p3[2]);
    V3 po = cp;
    if (ctx_n3) {
        V3fi pi(Interval(cp.x, cp.x), Interval(cp.y, cp.y), Interval(cp.z, cp.z));
        po = offset_ray_origin(pi, V3(ctx_n3[0], ctx_n3[1], ctx_n3[2]), sphere_center(sp) - cp);
    }
    ShapeSample ss;
    if (!sphere_sample_ctx(sp, cp, po, P2{u2[0], u2[1]}, &ss)) return 0;
    V3 m = ss.pi.midpoints();
    out7[0] = m.x; out7[1] = m.y; out7[2] = m.z; out7[3] = ss.n.x; out7[4] = ss.n.y; out7[5] = ss.n.z; out7[6] = ss.pdf;
    return 1;
}
// batch form for the device comparison (tests/test_sphere_light_gpu.py): item i: in8 = ctx_p[3], p_origin[3], u[2]; out8 = ok, p[3], n[3], pdf
void ora_sphere_sample_ctx_batch(const float* sph5, const float* rfo16, const float* ofr16, const float* in8, int n, float* out8) {
    M4 rfo, ofr;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { rfo.m[r][c] = rfo16[4 * r + c]; ofr.m[r][c] = ofr16[4 * r + c]; }
    Sphere sp = make_sphere(sph5[0], sph5[1], sph5[2], sph5[3], rfo, ofr, sph5[4] != 0.0f);
    for (int i = 0; i < n; ++i) {
        const float* q = in8 + 8 * (size_t)i; float* o = out8 + 8 * (size_t)i;
        ShapeSample ss; ss.n = V3(0.0f, 0.0f, 0.0f); ss.pdf = 0.0f;
        bool ok = sphere_sample_ctx(sp, V3(q[0], q[1], q[2]), V3(q[3], q[4], q[5]), P2{q[6], q[7]}, &ss);
        V3 m = ok ? ss.pi.midpoints() : V3(0.0f, 0.0f, 0.0f);
        o[0] = ok ? 1.0f : 0.0f; o[1] = m.x; o[2] = m.y; o[3] = m.z; o[4] = ss.n.x; o[5] = ss.n.y; o[6] = ss.n.z; o[7] = ss.pdf;
    }
}
int ora_patch_is_rectangle(const float* p12) {
    Patch p; auto P = [&](int k) { return V3(p12[3 * k], p12[3 * k + 1], p12[3 * k + 2]); };
    p.p00 = P(0); p.p10 = P(1); p.p01 = P(2); p.p11 = P(3);
    return patch_is_rectangle(p) ? 1 : 0;
}

// transcendental mode of the per-sample path (ref_core.h trans_mode): 0 faithful glibc f32, 1 f64-evaluated and rounded
void ora_set_transcendental_mode(int mode) { trans_mode() = mode ? 1 : 0; }
int ora_get_transcendental_mode() { return trans_mode(); }

// ---- single-function probes used by the golden-vector tests -----------------------------------
float ora_next_float_up(float v) { return next_float_up(v); }
float ora_next_float_down(float v) { return next_float_down(v); }
float ora_gamma(int n) { return gamma_n(n); }
int ora_exponent(float v) { return exponent_f32(v); }
float ora_fast_exp(float x) { return fast_exp(x); }
float ora_gaussian(float x, float mu, float sigma) { return gaussian(x, mu, sigma); }
float ora_erf_inv(float a) { return erf_inv(a); }
float ora_evaluate_polynomial(float t, const float* c, int n) { return evaluate_polynomial(t, c, n); }
float ora_difference_of_products(float a, float b, float c, float d) { return difference_of_products(a, b, c, d); }
int ora_solve_quadratic(float a, float b, float c, float* t01) { return solve_quadratic(a, b, c, &t01[0], &t01[1]) ? 1 : 0; }
int ora_mat4_inverse(const float* m16, float* out16) {
    M4 r;
    if (!m4_inverse(m4_from(m16), &r)) return 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out16[i * 4 + j] = r.m[i][j];
    return 1;
}
void ora_xf_bounds(const float* m16, const float* b6, float* out6) {
    B3 b; b.lo = V3(b6[0], b6[1], b6[2]); b.hi = V3(b6[3], b6[4], b6[5]);
    B3 r = xf_bounds(m4_from(m16), b);
    out6[0] = r.lo.x; out6[1] = r.lo.y; out6[2] = r.lo.z; out6[3] = r.hi.x; out6[4] = r.hi.y; out6[5] = r.hi.z;
}
float ora_visible_wavelengths_pdf(float l) { return visible_wavelengths_pdf(l); }
float ora_sample_visible_wavelengths(float u) { return sample_visible_wavelengths(u); }
int ora_sphere_basic_intersect(float radius, float z_min, float z_max, float phi_max_deg, const float* rfo,
                               const float* ofr, const float* o3, const float* d3, float t_max, float* out5) {
    Sphere s = make_sphere(radius, z_min, z_max, phi_max_deg, m4_from(rfo), m4_from(ofr), false);
    QuadricHit q;
    if (!sphere_basic_intersect(s, V3(o3[0], o3[1], o3[2]), V3(d3[0], d3[1], d3[2]), t_max, &q)) return 0;
    out5[0] = q.t_hit; out5[1] = q.p_obj.x; out5[2] = q.p_obj.y; out5[3] = q.p_obj.z; out5[4] = q.phi;
    return 1;
}
int ora_intersect_patch(const float* o3, const float* d3, float t_max, const float* p12, float* tuv) {
    PatchHit h;
    auto P = [&](int k) { return V3(p12[3 * k], p12[3 * k + 1], p12[3 * k + 2]); };
    if (!intersect_bilinear_patch(V3(o3[0], o3[1], o3[2]), V3(d3[0], d3[1], d3[2]), t_max, P(0), P(1), P(2), P(3), &h))
        return 0;
    tuv[0] = h.t; tuv[1] = h.u; tuv[2] = h.v;
    return 1;
}
uint64_t ora_hash_pixel(int32_t x, int32_t y, uint32_t seed) { return hash_pixel(x, y, seed); }
void ora_sampler_stream(int32_t x, int32_t y, uint32_t seed, uint64_t sample_index, uint64_t dim, int n, float* out) {
    Pcg32 r = start_pixel_sample(x, y, seed, sample_index, dim);
    for (int i = 0; i < n; ++i) out[i] = r.uniform_float();
}
uint32_t ora_pcg32_next(uint64_t* state_inc) {
    Pcg32 r; r.state = state_inc[0]; r.inc = state_inc[1];
    uint32_t v = r.next_u32();
    state_inc[0] = r.state;
    return v;
}
void ora_pcg32_new(uint64_t state, uint64_t stream, uint64_t* state_inc) {
    Pcg32 r = Pcg32::create(state, stream);
    state_inc[0] = r.state; state_inc[1] = r.inc;
}

// ---- the reference's scene-construction code (oracle/ref_host.h), one entry per constructor the product's host layer mirrors
static host::T4 t4_from(const float* m16, const float* inv16) {
    host::T4 t; std::memcpy(t.m.m, m16, 64); std::memcpy(t.inv.m, inv16, 64); return t;
}
int ora_host_look_at(const float* pos3, const float* look3, const float* up3, float* m16, float* inv16) {
    host::T4 t;
    if (!host::look_at(V3(pos3[0], pos3[1], pos3[2]), V3(look3[0], look3[1], look3[2]), V3(up3[0], up3[1], up3[2]), &t)) return 1;
    std::memcpy(m16, t.m.m, 64); std::memcpy(inv16, t.inv.m, 64);
    return 0;
}
int ora_host_perspective_camera(const float* wfc_m16, const float* wfc_inv16, float fov_deg, const float* screen_window4, int res_x,
                                int res_y, float lens_radius, float focal_distance, OCamera* out, float* world_from_render16,
                                float* render_from_world16) {
    host::CameraOut co;
    if (!host::perspective_camera(t4_from(wfc_m16, wfc_inv16), fov_deg, screen_window4, res_x, res_y, lens_radius, focal_distance, &co)) return 1;
    *out = co.cam;
    std::memcpy(world_from_render16, co.world_from_render.m.m, 64);
    std::memcpy(render_from_world16, co.world_from_render.inv.m, 64);
    return 0;
}
void ora_host_compose(const float* a_m, const float* a_inv, const float* b_m, const float* b_inv, float* m16, float* inv16) {
    const host::T4 t = host::compose(t4_from(a_m, a_inv), t4_from(b_m, b_inv));
    std::memcpy(m16, t.m.m, 64); std::memcpy(inv16, t.inv.m, 64);
}
void ora_host_translate(const float* d3, float* m16, float* inv16) {
    const host::T4 t = host::translate(V3(d3[0], d3[1], d3[2]));
    std::memcpy(m16, t.m.m, 64); std::memcpy(inv16, t.inv.m, 64);
}
int ora_host_frame_towards(const float* from3, const float* to3, float* m16, float* inv16) {
    host::T4 t;
    if (!host::frame_towards(V3(from3[0], from3[1], from3[2]), V3(to3[0], to3[1], to3[2]), &t)) return 1;
    std::memcpy(m16, t.m.m, 64); std::memcpy(inv16, t.inv.m, 64);
    return 0;
}
int ora_host_gaussian_filter_table(float rx, float ry, float sigma, float* vals, int cap, int* nx, int* ny) {
    std::vector<float> v;
    host::gaussian_filter_table(rx, ry, sigma, &v, nx, ny);
    if (vals) { if ((int)v.size() > cap) return 1; std::memcpy(vals, v.data(), v.size() * sizeof(float)); }
    return 0;
}
void ora_host_piecewise_linear_table(const float* il, int n, int normalize, const float* cie_y471, float* out471) { host::piecewise_linear_dense(il, n, normalize != 0, cie_y471, out471); }
void ora_host_d65_table(const float* cie_y471, const float* d65_interleaved, int n, float* out471) { host::d65_dense(cie_y471, d65_interleaved, n, out471); }
float ora_host_photometric(const float* cie_y471, const float* table471) { return host::photometric(cie_y471, table471); }
float ora_host_blackbody_table(float temp, float* out471) { return host::blackbody_dense(temp, out471); }
void ora_host_rgb_spectrum_table(const float* c3, float scale, const float* illum471_or_null, float* out471) {
    host::rgb_spectrum_dense(c3, scale, illum471_or_null, out471);
}
int ora_host_srgb_matrices(const float* cie_x, const float* cie_y, const float* cie_z, const float* illum471, float* xyz_from_rgb9,
                           float* rgb_from_xyz9) {
    float a[3][3], b[3][3];
    if (!host::srgb_matrices(cie_x, cie_y, cie_z, illum471, a, b)) return 1;
    std::memcpy(xyz_from_rgb9, a, 36); std::memcpy(rgb_from_xyz9, b, 36);
    return 0;
}

// BilinearPatchMesh::new / TriangleMesh::new transform their vertices with render_from_object * p (transform.rs:320-335)
void ora_host_xf_points(const float* m16, const float* pts, int n, float* out) {
    M4 m; std::memcpy(m.m, m16, 64);
    for (int i = 0; i < n; ++i) { const V3 q = xf_point(m, V3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2])); out[3 * i] = q.x; out[3 * i + 1] = q.y; out[3 * i + 2] = q.z; }
}

}  // extern "C"
