// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_core.h header).
//
// CPU walk of the GPU-layout wide BVH (80-byte nodes + leaf records exported by
// pm_get_node_buffer / pm_get_prim_buffer), in the same order as the HIP kernel, using the
// oracle's own primitive tests.  Purpose: the deterministic per-ray counters
// N_node / N_prim that define the algorithmic bytes, and a device-independent check of the
// traversal order logic.  The layout is re-declared here on purpose (no product header is included).
#pragma once
#include <cstring>
#include <vector>

#include "ref_scene.h"

namespace ora {

struct WNode {            // must match pbrt-rs_amd/csrc/pm_layout.h PmWideNode (checked by size + tests)
    float o[3];
    uint8_t e[3];
    uint8_t imask;
    uint32_t child_base;
    uint32_t prim_base;
    uint8_t meta[8];
    uint8_t qlo[3][8];
    uint8_t qhi[3][8];
};
static_assert(sizeof(WNode) == 80, "wide node is 80 bytes");

static inline uint32_t perm8(uint32_t m, uint32_t oct) {
    uint32_t r = 0;
    for (uint32_t k = 0; k < 8; ++k) if (m & (1u << (k ^ oct))) r |= 1u << k;
    return r;
}

struct WideWalker {
    const WNode* nodes;
    const uint8_t* prims;   // 16-byte units
    bool defer_spheres = true;   // GPU contexts defer spheres that pass the float pre-filter to a second pass

    bool test_prim(uint32_t type, uint32_t unit, V3 ro, V3 rd, float t_max, uint32_t* id, float* t, float* u, float* v) const {
        const float* f = reinterpret_cast<const float*>(prims + (size_t)unit * 16);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(f);
        if (type == 2) {          // triangle record: v0 v1 v2 id
            V3 v0(f[0], f[1], f[2]), v1(f[3], f[4], f[5]), v2(f[6], f[7], f[8]);
            PatchHit h;
            if (!intersect_bilinear_patch(ro, rd, t_max, v0, v1, v2, v2, &h)) return false;
            *id = w[9]; *t = h.t; *u = h.u; *v = h.v; return true;
        } else if (type == 1) {   // patch record
            V3 a(f[0], f[1], f[2]), b(f[3], f[4], f[5]), c(f[6], f[7], f[8]), d(f[9], f[10], f[11]);
            PatchHit h;
            if (!intersect_bilinear_patch(ro, rd, t_max, a, b, c, d, &h)) return false;
            *id = w[12]; *t = h.t; *u = h.u; *v = h.v; return true;
        } else {                  // sphere record: ofr rows 0..2, radius zmin zmax phimax(rad), id
            Sphere s;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) s.object_from_render.m[r][c] = f[r * 4 + c];
            s.object_from_render.m[3][0] = 0; s.object_from_render.m[3][1] = 0; s.object_from_render.m[3][2] = 0; s.object_from_render.m[3][3] = 1;
            s.radius = f[12]; s.z_min = f[13]; s.z_max = f[14]; s.phi_max = f[15];
            QuadricHit q;
            if (!sphere_basic_intersect(s, ro, rd, t_max, &q)) return false;
            float th0 = std::acos(clampf(s.z_min / s.radius, -1.0f, 1.0f)), th1 = std::acos(clampf(s.z_max / s.radius, -1.0f, 1.0f));
            *id = w[16]; *t = q.t_hit; *u = q.phi / s.phi_max; *v = (safe_acos(q.p_obj.z / s.radius) - th0) / (th1 - th0);
            return true;
        }
    }

    // ---- the kernel's float pre-filter for triangles, operation for operation (pm_device.hpp tri_prefilter)
    static V3 fcross(V3 a, V3 b) {
        return V3(std::fmaf(a.y, b.z, -(a.z * b.y)), std::fmaf(a.z, b.x, -(a.x * b.z)), std::fmaf(a.x, b.y, -(a.y * b.x)));
    }
    static float fdot(V3 a, V3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
    static float maxabs(V3 a) { float m = std::fabs(a.x); float y = std::fabs(a.y), z = std::fabs(a.z); if (!(y < m)) m = y; if (!(z < m)) m = z; return m; }
    static int tri_prefilter(V3 ro, V3 rd, float R, float D, V3 v0, V3 v1, V3 v2, float tcull, float* t_hi, float* t_lo) {
        V3 e1 = v1 - v0, e2 = v2 - v0, sv = ro - v0;
        V3 pv = fcross(rd, e2);
        float det = fdot(e1, pv), uu = fdot(sv, pv);
        V3 qv = fcross(sv, e1);
        float vv = fdot(rd, qv), tt = fdot(e2, qv);
        float L = rmax(maxabs(e1), maxabs(e2)), S = maxabs(sv);
        return tri_classify(det, uu, vv, tt, L, S, R, D, tcull, t_hi, t_lo);
    }
    // pm_device.hpp tri_prefilter_flat: the same triple products from the flat-table entry (e1, e2, n = e1 x e2, L, c1, kll),
    // no division: the culling distance is the fraction cn / cd, the candidate's lower bound the fraction ln / ld
    struct FlatCull { float cn, cd; };
    static int tri_prefilter_flat(V3 ro, V3 rd, float KD, float R2, V3 v0, V3 e1, V3 e2, V3 n, float L, float c1, float kll,
                                  float epsb_ray, FlatCull* cull, bool lower, float* ln, float* ld) {
        V3 sv = ro - v0;
        V3 m = fcross(sv, rd);
        float uu = fdot(e2, m), vv = -fdot(e1, m), det = -fdot(rd, n), tt = fdot(sv, n);
        if (det < 0.0f) { det = -det; uu = -uu; vv = -vv; tt = -tt; }
        const float Sx = R2 + c1;
        const float tb = (KD * L) * Sx;
        const float ttau = kll * Sx;
        const float dp = det + tb;
        const bool reject = (uu < -tb) | (vv < -tb) | (uu + vv > dp + tb) | (tt < -ttau) | ((tt - ttau) * cull->cd > cull->cn * dp);
        if (reject) return 0;
        *ln = 0.0f; *ld = 1.0f;
        if ((uu > tb) & (vv > tb) & (uu + vv < (det - tb) - tb) & (det > 2.0f * tb) & (tt > ttau)) {
            const float hn = (tt + ttau) * 1.00001f, hd = det - tb;
            const float lo = (tt - ttau) * 0.99999f;
            const float epsb = std::fmaf(3.2e-6f, c1, epsb_ray);
            if ((lo > epsb * dp) & (hn * cull->cd < cull->cn * hd)) {
                if (lower) { cull->cn = hn; cull->cd = hd; }
                *ln = lo; *ld = dp;
                return 2;
            }
        }
        return 1;
    }
    static int tri_classify(float det, float uu, float vv, float tt, float L, float S, float R, float D, float tcull, float* t_hi,
                            float* t_lo) {
        const float K = 7.62939453125e-06f;
        if (det < 0.0f) { det = -det; uu = -uu; vv = -vv; tt = -tt; }
        float Sx = S + R + L;
        float tb = K * D * L * Sx;
        float ttau = K * L * L * Sx;
        if (uu < -tb || vv < -tb || uu + vv > det + 2.0f * tb) return 0;
        if (tt < -ttau) return 0;
        if (tt > std::fmaf(tcull, det + tb, ttau)) return 0;
        *t_lo = 0.0f;
        if (uu > tb && vv > tb && uu + vv < det - 2.0f * tb && det > 2.0f * tb && tt > ttau) {
            float epsb = 8e-7f * (5.0f * R + D + 4.0f * (S + L));
            float th = ((tt + ttau) / (det - tb)) * 1.00001f;
            float tl = ((tt - ttau) / (det + tb)) * 0.99999f;
            if (tl > epsb && th < tcull) { *t_hi = th; *t_lo = tl; return 2; }
        }
        return 1;
    }
    // ---- the kernel's float pre-filter for spheres, operation for operation (pm_device.hpp sphere_prefilter)
    // pm_device.hpp patch_prefilter: the padded box of the four corners against the ray, slab distances with relative slack
    static bool patch_prefilter(V3 ro, V3 inv, float R, V3 p00, V3 p10, V3 p01, V3 p11, float tcull) {
        const V3 lo(rmin(rmin(p00.x, p10.x), rmin(p01.x, p11.x)), rmin(rmin(p00.y, p10.y), rmin(p01.y, p11.y)),
                    rmin(rmin(p00.z, p10.z), rmin(p01.z, p11.z)));
        const V3 hi(rmax(rmax(p00.x, p10.x), rmax(p01.x, p11.x)), rmax(rmax(p00.y, p10.y), rmax(p01.y, p11.y)),
                    rmax(rmax(p00.z, p10.z), rmax(p01.z, p11.z)));
        const float pad = 1e-4f * rmax(rmax(maxabs(lo), maxabs(hi)), R);
        const float ax = ((lo.x - pad) - ro.x) * inv.x, bx = ((hi.x + pad) - ro.x) * inv.x;
        const float ay = ((lo.y - pad) - ro.y) * inv.y, by = ((hi.y + pad) - ro.y) * inv.y;
        const float az = ((lo.z - pad) - ro.z) * inv.z, bz = ((hi.z + pad) - ro.z) * inv.z;
        const float tn = rmax(rmax(rmin(ax, bx), rmin(ay, by)), rmax(rmin(az, bz), 0.0f));
        const float tf = rmin(rmin(rmax(ax, bx), rmax(ay, by)), rmin(rmax(az, bz), tcull));
        return tn <= tf * 1.0001f;
    }
    static bool sphere_prefilter(V3 ro, V3 rd, float R, float D, const float* m, float radius, float m_norm, float t_norm, float tcull) {
        const float EPS = 4.76837158203125e-07f;
        V3 o(std::fmaf(m[0], ro.x, std::fmaf(m[1], ro.y, std::fmaf(m[2], ro.z, m[3]))),
             std::fmaf(m[4], ro.x, std::fmaf(m[5], ro.y, std::fmaf(m[6], ro.z, m[7]))),
             std::fmaf(m[8], ro.x, std::fmaf(m[9], ro.y, std::fmaf(m[10], ro.z, m[11]))));
        V3 d(std::fmaf(m[0], rd.x, std::fmaf(m[1], rd.y, m[2] * rd.z)),
             std::fmaf(m[4], rd.x, std::fmaf(m[5], rd.y, m[6] * rd.z)),
             std::fmaf(m[8], rd.x, std::fmaf(m[9], rd.y, m[10] * rd.z)));
        float a = fdot(d, d), b = fdot(o, d), oo = fdot(o, o), rr = radius * radius;
        float c = oo - rr;
        float Om = rmax(rmax(std::fabs(o.x), std::fabs(o.y)), std::fabs(o.z)), Dm = rmax(rmax(std::fabs(d.x), std::fabs(d.y)), std::fabs(d.z));
        float eo = EPS * std::fmaf(m_norm, R, t_norm);
        float ed = EPS * (m_norm * D);
        float db = 3.0f * std::fmaf(Om, ed, Dm * eo);
        float dc = 6.0f * (Om * eo), da = 6.0f * (Dm * ed);
        float base = EPS * std::fmaf(a, oo + rr, b * b);
        float disc = std::fmaf(b, b, -(a * c));
        float E = 4.0f * (std::fmaf(2.0f * std::fabs(b), db, std::fmaf(a, dc, std::fabs(c) * da)) + base);
        if (disc < -E) return false;
        float cl = c - 2.0f * std::fmaf(EPS, oo + rr, dc);
        if (cl > 0.0f) {
            if (b > db) return false;
            float den = (db - b) + std::sqrt(rmax(disc, 0.0f) + E) * 1.000001f;
            if (den > 0.0f && (cl / den) * 0.99999f > tcull) return false;
        }
        return true;
    }

    // ---- flat sweep (pm_device.hpp trace_flat) for scenes of at most PM_FLAT_MAX_PRIMS primitives.  The table is derived
    // from the exported buffers exactly as pm_build.cpp build_flat_table derives it: node index, slot, record order.
    struct FlatTri { V3 v0, e1, e2, n; float L, c1, kll; uint32_t unit; };
    struct FlatTable { std::vector<uint32_t> spheres, patches; std::vector<FlatTri> tris; };
    FlatTable flat_table() const {
        uint32_t n_nodes = 1;
        for (uint32_t i = 0; i < n_nodes; ++i) {
            const WNode& nd = nodes[i];
            if (nd.imask) n_nodes = std::max(n_nodes, nd.child_base + (uint32_t)__builtin_popcount(nd.imask));
        }
        FlatTable ft;
        for (uint32_t i = 0; i < n_nodes; ++i) {
            const WNode& nd = nodes[i];
            const uint32_t type = nd.prim_base >> 30, pbase = nd.prim_base & 0x3FFFFFFFu;
            const uint32_t units = type == 2 ? 3 : (type == 1 ? 4 : 5);
            for (int s = 0; s < 8; ++s) {
                if ((nd.imask >> s) & 1) continue;
                const uint32_t cnt = nd.meta[s] >> 5, off = nd.meta[s] & 31u;
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint32_t unit = pbase + (off + j) * units;
                    if (type == 0) ft.spheres.push_back(unit);
                    else if (type == 1) ft.patches.push_back(unit);
                    else {
                        const float* f = reinterpret_cast<const float*>(prims + (size_t)unit * 16);
                        FlatTri t;
                        t.v0 = V3(f[0], f[1], f[2]);
                        t.e1 = V3(f[3] - f[0], f[4] - f[1], f[5] - f[2]);
                        t.e2 = V3(f[6] - f[0], f[7] - f[1], f[8] - f[2]);
                        t.n = V3((float)((double)t.e1.y * (double)t.e2.z - (double)t.e1.z * (double)t.e2.y),
                                 (float)((double)t.e1.z * (double)t.e2.x - (double)t.e1.x * (double)t.e2.z),
                                 (float)((double)t.e1.x * (double)t.e2.y - (double)t.e1.y * (double)t.e2.x));
                        t.L = rmax(maxabs(t.e1), maxabs(t.e2));
                        t.c1 = maxabs(t.v0) + t.L;
                        t.kll = (7.62939453125e-06f * t.L) * t.L;
                        t.unit = unit;
                        ft.tris.push_back(t);
                    }
                }
            }
        }
        // table order (pm_layout.h, pm_build.cpp build_flat_table): groups of bit-equal v0 by first member; inside a group the
        // next entry is the first unplaced one whose e1 is the current entry's e2, else the first unplaced one.  (The kernel
        // reuses sv, sv x rd and the shared edge's product inside a group; recomputing them here gives the same bits.)
        {
            const std::vector<FlatTri> in = ft.tris;
            auto same = [](const V3& a, const V3& b) { return std::memcmp(&a, &b, sizeof(V3)) == 0; };
            std::vector<char> used(in.size(), 0);
            ft.tris.clear();
            for (size_t i = 0; i < in.size(); ++i) {
                if (used[i]) continue;
                std::vector<size_t> group;
                for (size_t j = i; j < in.size(); ++j) if (!used[j] && same(in[j].v0, in[i].v0)) group.push_back(j);
                long cur = -1;
                for (size_t left = group.size(); left; --left) {
                    long next = -1;
                    if (cur >= 0)
                        for (size_t j : group) if (!used[j] && same(in[j].e1, in[(size_t)cur].e2)) { next = (long)j; break; }
                    if (next < 0)
                        for (size_t j : group) if (!used[j]) { next = (long)j; break; }
                    ft.tris.push_back(in[(size_t)next]);
                    used[(size_t)next] = 1; cur = next;
                }
            }
        }
        return ft;
    }
    // pm_device.hpp trace_flat + deferred_fix: the lean sweep (pre-filters + exact triangle round), then the second pass
    // (deferred spheres / patches, or - for a slow ray - everything again through the exact tests)
    bool walk_flat(const FlatTable& ft, V3 ro, V3 rd, float ray_tmax, bool any_hit, uint32_t* hid, float* ht, float* hu, float* hv,
                   Counters* ctr) const {
        *hid = 0xFFFFFFFFu; *ht = ray_tmax; *hu = 0; *hv = 0;
        {
            const float dsum = std::fabs(rd.x) + std::fabs(rd.y) + std::fabs(rd.z), osum = std::fabs(ro.x) + std::fabs(ro.y) + std::fabs(ro.z);
            if (!(dsum > 0.0f && dsum < std::numeric_limits<float>::infinity() && osum < std::numeric_limits<float>::infinity())) return false;
        }
        const uint32_t CAND_MAX = 4, PATCH_TAG = 0x40000000u;       // PM_FLAT_CAND_MAX
        const float R = maxabs(ro), D = maxabs(rd);
        float tcull = ray_tmax;
        float t_max = ray_tmax;
        bool sure = false, found = false, slow = false;
        std::vector<uint32_t> deferred;
        auto defer_prim = [&](uint32_t tagged) { if (deferred.size() < 3) deferred.push_back(tagged); else slow = true; };
        auto test_now = [&](uint32_t type, uint32_t unit) {
            if (ctr) ctr->exact_tests++;
            uint32_t id; float t, u, v;
            if (test_prim(type, unit, ro, rd, t_max, &id, &t, &u, &v)) { t_max = t; found = true; *hid = id; *ht = t; *hu = u; *hv = v; return true; }
            return false;
        };
        if (any_hit)                                        // patches, any hit: reference arithmetic at once, before the triangles
            for (uint32_t unit : ft.patches) {
                if (slow || found) break;
                if (ctr) ctr->prims_tested++;
                if (test_now(1, unit)) tcull = rmin(tcull, *ht);
            }
        struct Cand { uint32_t unit; float ln, ld; };
        std::vector<Cand> cand;
        FlatCull cull{tcull, 1.0f};
        const float KD = 7.62939453125e-06f * D, R2 = 2.0f * R, epsb_ray = 8e-7f * std::fmaf(9.0f, R, D);
        for (const FlatTri& t : ft.tris) {
            if (slow || (any_hit && (found || sure))) break;
            if (ctr) ctr->prims_tested++;
            float ln, ld;
            int cls = tri_prefilter_flat(ro, rd, KD, R2, t.v0, t.e1, t.e2, t.n, t.L, t.c1, t.kll, epsb_ray, &cull, !any_hit, &ln, &ld);
            if (cls != 0) {
                if (cls == 2) sure = true;
                if (cand.size() < CAND_MAX) cand.push_back(Cand{t.unit, ln, ld});
                else slow = true;
            }
        }
        // spheres and (closest hit) patches AFTER the triangles: pre-filters against the distance a certain triangle hit leaves
        // (any hit: a decided ray skips them); what passes is deferred to the second pass
        float t_rej = std::numeric_limits<float>::infinity();
        if (!(any_hit && (found || sure))) {
            const float tcs = (cull.cn / cull.cd) * 1.00001f;
            for (uint32_t unit : ft.spheres) {
                if (slow) break;
                if (ctr) ctr->prims_tested++;
                const float* f = reinterpret_cast<const float*>(prims + (size_t)unit * 16);
                if (sphere_prefilter(ro, rd, R, D, f, f[12], f[18], f[19], tcs)) defer_prim(unit);
                else {      // the counting kernels' confirm path (pm_device.hpp trace_flat): the interval test all the same (not counted)
                    uint32_t id; float t, u, v;
                    if (test_prim(0, unit, ro, rd, ray_tmax, &id, &t, &u, &v)) t_rej = rmin(t_rej, t);
                }
            }
            if (!any_hit) {
                auto rcp = [D](float d) { return std::fabs(d) < 1e-18f * D ? std::copysign(1e18f / D, d) : 1.0f / d; };   // pm_device.hpp safe_rcp_dir
                const V3 inv(rcp(rd.x), rcp(rd.y), rcp(rd.z));
                for (uint32_t unit : ft.patches) {
                    if (slow) break;
                    if (ctr) ctr->prims_tested++;
                    const float* f = reinterpret_cast<const float*>(prims + (size_t)unit * 16);
                    if (patch_prefilter(ro, inv, R, V3(f[0], f[1], f[2]), V3(f[3], f[4], f[5]), V3(f[6], f[7], f[8]), V3(f[9], f[10], f[11]), tcs))
                        defer_prim(unit | PATCH_TAG);
                    else {      // the counting kernels' confirm path: the reference arithmetic all the same (pm_device.hpp trace_flat)
                        if (ctr) ctr->exact_tests++;
                        uint32_t id; float t, u, v;
                        if (test_prim(1, unit, ro, rd, ray_tmax, &id, &t, &u, &v)) t_rej = rmin(t_rej, t);
                    }
                }
            }
        }
        if (!slow) {
            const float tconf = next_float_down(cull.cn / cull.cd);
            // any hit: last candidate first (a certain hit ends the sweep, so it is the last one queued and the round is over
            // after one test; the candidates before it are mostly the triangle the shadow ray starts on)
            for (size_t ci = 0; ci < cand.size(); ++ci) {
                const Cand& cu = cand[any_hit ? cand.size() - 1 - ci : ci];
                if (cu.ln * cull.cd > cull.cn * cu.ld) continue;        // behind another certain hit (filtered before the round)
                if (any_hit && found) break;
                (void)test_now(2, cu.unit);
            }
            const bool false_reject = any_hit ? (!found && t_rej < std::numeric_limits<float>::infinity()) : t_rej < (found ? *ht : ray_tmax);
            if ((sure && !(found && (any_hit || *ht <= tconf))) || false_reject) { slow = true; if (ctr) ctr->fallback_walks++; }
        }
        if (slow) {             // second pass, slow entry: everything again, exactly
            t_max = ray_tmax; found = false;
            *hid = 0xFFFFFFFFu; *ht = ray_tmax; *hu = 0; *hv = 0;
            for (uint32_t unit : ft.spheres) { if (any_hit && found) break; if (ctr) ctr->prims_tested++; (void)test_now(0, unit); }
            for (uint32_t unit : ft.patches) { if (any_hit && found) break; if (ctr) ctr->prims_tested++; (void)test_now(1, unit); }
            for (const FlatTri& t : ft.tris) { if (any_hit && found) break; if (ctr) ctr->prims_tested++; (void)test_now(2, t.unit); }
            return found;
        }
        for (uint32_t tagged : deferred) {                          // second pass, ordinary entry
            if (any_hit && found) break;
            if (tagged & PATCH_TAG) (void)test_now(1, tagged & ~PATCH_TAG); else (void)test_now(0, tagged);
        }
        return found;
    }

    // returns true if something was hit; closest-hit unless any_hit.  Mirrors pm_device.hpp traverse<>: pre-scaled slab test
    // with folded margins, triangle pre-filter + queue of PM_CAND_MAX candidates tested exactly after the walk, culling by
    // tcull, fallback walk with immediate exact tests when the exact tests do not confirm the culling.
    bool walk(V3 ro, V3 rd, float ray_tmax, bool any_hit, uint32_t* hid, float* ht, float* hu, float* hv, Counters* ctr,
              std::vector<uint8_t>* itrace = nullptr) const {
        {   // degenerate rays hit nothing in the reference; the kernel answers them without walking (pm_device.hpp)
            const float dsum = std::fabs(rd.x) + std::fabs(rd.y) + std::fabs(rd.z), osum = std::fabs(ro.x) + std::fabs(ro.y) + std::fabs(ro.z);
            if (!(dsum > 0.0f && dsum < std::numeric_limits<float>::infinity() && osum < std::numeric_limits<float>::infinity())) {
                *hid = 0xFFFFFFFFu; *ht = ray_tmax; *hu = 0; *hv = 0;
                return false;
            }
        }
        const float widen = 1.0f + 2.0f * gamma_n(3);
        const float MARGIN = 2.384185791015625e-07f;
        const uint32_t CAND_MAX = 4;
        const float R = maxabs(ro), D = maxabs(rd);
        auto rcp = [D](float d) { return std::fabs(d) < 1e-18f * D ? std::copysign(1e18f / D, d) : 1.0f / d; };   // pm_device.hpp safe_rcp_dir
        const float inv[3] = {rcp(rd.x), rcp(rd.y), rcp(rd.z)};
        const bool ng[3] = {inv[0] < 0, inv[1] < 0, inv[2] < 0};
        const uint32_t oct = (ng[0] ? 1u : 0u) | (ng[1] ? 2u : 0u) | (ng[2] ? 4u : 0u);
        const float o3[3] = {ro.x, ro.y, ro.z};
        struct Group { uint32_t base, imask, pm; };
        bool found = false;
        bool exact_mode = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            float t_max = ray_tmax, tcull = ray_tmax;
            float t_rej = std::numeric_limits<float>::infinity();      // confirm path of the sphere pre-filter (counting kernels)
            bool sure = false;
            std::vector<std::pair<uint32_t, float>> cand;   // (record unit, lower bound of t: 0 unless a certain hit)
            std::vector<uint32_t> deferred;     // spheres that passed the pre-filter: exact tests after everything else
            std::vector<Group> stack;
            Group cur{0u, 1u, perm8(1u, oct)};
            found = false;
            *hid = 0xFFFFFFFFu; *ht = t_max; *hu = 0; *hv = 0;
            // the kernel's tie rule (pm_device.hpp tri_take): the exact triangle test is asked for t <= t_max, and an exact-t tie
            // with the hit held so far goes to the LOWER primitive id - independent of the order of the tests
            auto tri_take = [&](uint32_t unit, uint32_t ptype = 2) -> bool {       // (triangles and patches alike)
                uint32_t id; float t, u, v;
                if (!test_prim(ptype, unit, ro, rd, any_hit ? t_max : next_float_up(t_max), &id, &t, &u, &v)) return false;
                if (t < t_max || (!any_hit && found && t == t_max && id < *hid)) {
                    t_max = t; found = true; *hid = id; *ht = t; *hu = u; *hv = v;
                    return true;
                }
                return false;
            };
            for (;;) {
                if (any_hit && (found || sure)) break;
                if (cur.pm == 0) {
                    if (stack.empty()) break;
                    cur = stack.back(); stack.pop_back();
                    if (itrace) itrace->push_back(255);
                    continue;
                }
                if (itrace) itrace->push_back(0);
                uint32_t k = 0; while (!(cur.pm & (1u << k))) ++k;
                uint32_t slot = k ^ oct;
                uint32_t nidx = cur.base + (uint32_t)__builtin_popcount(cur.imask & ((1u << slot) - 1u));
                cur.pm &= ~(1u << k);
                if (cur.pm) stack.push_back(cur);
                const WNode& nd = nodes[nidx];
                if (ctr) ctr->nodes_visited++;
                float si[3], an[3], af[3], sf[3];
                for (int a = 0; a < 3; ++a) {
                    si[a] = u2f((uint32_t)nd.e[a] << 23) * inv[a];
                    float ad = (nd.o[a] - o3[a]) * inv[a];
                    float m = MARGIN * std::fmaf(256.0f, std::fabs(si[a]), std::fabs(ad));
                    an[a] = ad - m; af[a] = (ad + m) * widen; sf[a] = si[a] * widen;
                }
                uint32_t hits = 0;
                for (int s = 0; s < 8; ++s) {
                    float tn[3], tf[3];
                    for (int a = 0; a < 3; ++a) {
                        float qn = (float)(ng[a] ? nd.qhi[a][s] : nd.qlo[a][s]), qf = (float)(ng[a] ? nd.qlo[a][s] : nd.qhi[a][s]);
                        tn[a] = std::fmaf(qn, si[a], an[a]);
                        tf[a] = std::fmaf(qf, sf[a], af[a]);
                    }
                    float n = rmax(rmax(tn[0], tn[1]), rmax(tn[2], 0.0f));
                    float f = rmin(rmin(tf[0], tf[1]), rmin(tf[2], tcull));
                    if (n <= f) hits |= 1u << s;
                }
                // (unused child slots hold an inverted box; should one pass the slab test it is a leaf of zero primitives)
                uint32_t lm = perm8(hits & ~(uint32_t)nd.imask, oct);
                uint32_t type = nd.prim_base >> 30, pbase = nd.prim_base & 0x3FFFFFFFu;
                uint32_t units = type == 2 ? 3 : (type == 1 ? 4 : 5);
                for (uint32_t kl = 0; kl < 8 && !(any_hit && (found || sure)); ++kl) {
                    if (!(lm & (1u << kl))) continue;
                    uint32_t ls = kl ^ oct;
                    uint32_t cnt = nd.meta[ls] >> 5, off = nd.meta[ls] & 31u;
                    for (uint32_t j = 0; j < cnt && !(any_hit && (found || sure)); ++j) {
                        if (ctr) ctr->prims_tested++;
                        if (itrace && itrace->back() < 200) itrace->back()++;
                        uint32_t unit = pbase + (off + j) * units;
                        bool immediate = true;
                        if (type == 0 && !exact_mode && defer_spheres) {
                            const float* f = reinterpret_cast<const float*>(prims + (size_t)unit * 16);
                            if (!sphere_prefilter(ro, rd, R, D, f, f[12], f[18], f[19], tcull)) {
                                immediate = false;      // the counting kernels' confirm path: the interval test all the same (not counted)
                                uint32_t id; float t, u, v;
                                if (test_prim(0, unit, ro, rd, ray_tmax, &id, &t, &u, &v)) t_rej = rmin(t_rej, t);
                            }
                            else if (deferred.size() < 2) { deferred.push_back(unit); immediate = false; }
                        }
                        if (type == 2 && !exact_mode) {
                            const float* f = reinterpret_cast<const float*>(prims + (size_t)unit * 16);
                            float th, tl;
                            int cls = tri_prefilter(ro, rd, R, D, V3(f[0], f[1], f[2]), V3(f[3], f[4], f[5]), V3(f[6], f[7], f[8]), tcull, &th, &tl);
                            if (cls == 0) immediate = false;
                            else {
                                if (cls == 2) { sure = true; if (!any_hit) tcull = th; }
                                if (cand.size() < CAND_MAX) { cand.emplace_back(unit, tl); immediate = false; }
                            }
                        }
                        if (immediate) {
                            if (ctr) ctr->exact_tests++;
                            uint32_t id; float t, u, v;
                            if (type == 2) {
                                if (tri_take(unit)) { if (exact_mode) tcull = rmin(tcull, t_max); }
                            } else if (type == 1) {
                                if (tri_take(unit, 1)) tcull = rmin(tcull, t_max);
                            } else if (test_prim(type, unit, ro, rd, t_max, &id, &t, &u, &v)) {
                                t_max = t; found = true; *hid = id; *ht = t; *hu = u; *hv = v;
                                tcull = rmin(tcull, t);
                            }
                        }
                    }
                }
                cur = Group{nd.child_base, nd.imask, perm8(hits & nd.imask, oct)};
            }
            for (size_t ci = 0; ci < cand.size(); ++ci) {       // any hit: last candidate first (see walk_flat)
                const auto& cu = cand[any_hit ? cand.size() - 1 - ci : ci];
                if (any_hit && found) break;
                if (cu.second > tcull) continue;        // behind another certain hit
                if (ctr) ctr->exact_tests++;
                (void)tri_take(cu.first);
            }
            const bool false_reject = any_hit ? (!found && t_rej < std::numeric_limits<float>::infinity()) : t_rej < (found ? *ht : ray_tmax);
            bool confirmed = (!sure || (found && (any_hit || *ht <= tcull))) && !false_reject;
            if (confirmed || exact_mode) {
                for (uint32_t unit : deferred) {        // the kernel's second pass (sphere_fix)
                    if (any_hit && found) break;
                    if (ctr) ctr->exact_tests++;
                    uint32_t id; float t, u, v;
                    if (test_prim(0, unit, ro, rd, t_max, &id, &t, &u, &v)) { t_max = t; found = true; *hid = id; *ht = t; *hu = u; *hv = v; }
                }
                break;
            }
            exact_mode = true;
            if (ctr) ctr->fallback_walks++;
        }
        return found;
    }
};

}  // namespace ora
