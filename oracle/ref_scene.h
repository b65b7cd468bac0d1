// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_core.h header).
//
// CPU restatement of the reference's shapes and BVH aggregate:
//   src/shapes/sphere.rs, src/shapes/bilinear_patch.rs,
//   src/primitives/aggregates.rs, src/primitives/single.rs.
// Triangles do not exist in the reference; a triangle (v0,v1,v2)
// is the degenerate bilinear patch p00=v0, p10=v1, p01=v2, p11=v2, which the
// reference's patch code explicitly tolerates (bilinear_patch.rs:186-199).
#pragma once
#include <algorithm>
#include <vector>

#include "ref_core.h"

namespace ora {

enum PrimType : uint8_t { PRIM_SPHERE = 0, PRIM_PATCH = 1 };

struct Sphere {
    float radius, z_min, z_max, theta_z_min, theta_z_max, phi_max;
    M4 render_from_object, object_from_render;
    bool reverse_orientation;
};

// src/shapes/sphere.rs:34-68
static inline Sphere make_sphere(float radius, float z_min, float z_max, float phi_max_deg,
                                 const M4& rfo, const M4& ofr, bool reverse) {
    Sphere s;
    s.radius = radius;
    s.z_min = clampf(z_min, -radius, radius);
    s.z_max = clampf(z_max, -radius, radius);
    s.phi_max = clampf(phi_max_deg, 0.0f, 360.0f) * (PI_F / 180.0f);
    s.theta_z_min = std::acos(clampf(s.z_min / radius, -1.0f, 1.0f));
    s.theta_z_max = std::acos(clampf(s.z_max / radius, -1.0f, 1.0f));
    s.render_from_object = rfo;
    s.object_from_render = ofr;
    s.reverse_orientation = reverse;
    return s;
}

struct Patch {             // resolved corner positions + mesh flags
    V3 p00, p10, p01, p11;
    bool flip_normal;      // reverse_orientation ^ transform_swaps_handedness
    bool has_normals = false;
    V3 n00, n10, n01, n11; // vertex normals in render space (BilinearPatchMesh::vertex_normals)
    bool has_uv = false;
    float uv00[2], uv10[2], uv01[2], uv11[2];   // vertex texture coordinates (BilinearPatchMesh.uv)
};

struct Prim {
    PrimType type;
    uint32_t index;        // into spheres[] or patches[]
    uint32_t id;           // caller-visible primitive id (insertion order)
};

struct QuadricHit { float t_hit; V3 p_obj; float phi; };
struct PatchHit { float t, u, v; };

// src/shapes/sphere.rs:265-368
static inline bool sphere_basic_intersect(const Sphere& s, V3 ro, V3 rd, float t_max, QuadricHit* out) {
    V3fi o_obj = xf_point_fi(s.object_from_render, V3fi::exact(ro));
    V3fi d_obj = xf_vec_fi(s.object_from_render, V3fi::exact(rd));

    Interval a = d_obj.x.squared() + d_obj.y.squared() + d_obj.z.squared();
    Interval b = 2.0f * (d_obj.x * o_obj.x + d_obj.y * o_obj.y + d_obj.z * o_obj.z);
    Interval c = o_obj.x.squared() + o_obj.y.squared() + o_obj.z.squared() - Interval::exact(s.radius).squared();

    Interval f = b / (2.0f * a);
    V3fi v(o_obj.x - d_obj.x * f, o_obj.y - d_obj.y * f, o_obj.z - d_obj.z * f);
    Interval len = (v.x.squared() + v.y.squared() + v.z.squared()).sqrt();
    Interval discrim = 4.0f * a * (Interval::exact(s.radius) + len) * (Interval::exact(s.radius) - len);
    if (discrim.low < 0.0f) return false;

    Interval rd_ = discrim.sqrt();
    Interval q = (b.midpoint() < 0.0f) ? -0.5f * (b - rd_) : -0.5f * (b + rd_);
    Interval t0 = q / a;
    Interval t1 = c / q;
    if (t0.low > t1.low) std::swap(t0, t1);

    if (t0.high > t_max || t1.low <= 0.0f) return false;
    Interval t_shape;
    if (t0.low > 0.0f) t_shape = t0;
    else if (t1.high <= t_max) t_shape = t1;
    else return false;

    V3 om = o_obj.midpoints(), dm = d_obj.midpoints();
    auto point_and_phi = [&](Interval th, V3* p, float* phi) {
        V3 ph = om + th.midpoint() * dm;
        ph = ph * (s.radius / distance(ph, V3(0, 0, 0)));
        if (ph.x == 0.0f && ph.y == 0.0f) ph.x = 1e-5f * s.radius;
        float f = t_atan2(ph.y, ph.x);
        if (f <= 0.0f) f += 2.0f * PI_F;
        *p = ph; *phi = f;
    };
    V3 p_hit; float phi;
    point_and_phi(t_shape, &p_hit, &phi);
    auto clipped = [&]() {
        return (s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max) ||
               phi > s.phi_max;
    };
    if (clipped()) {
        if (t_shape == t1 || t1.high > t_max) return false;
        t_shape = t1;
        point_and_phi(t_shape, &p_hit, &phi);
        if (clipped()) return false;
    }
    out->t_hit = t_shape.midpoint();
    out->p_obj = p_hit;
    out->phi = phi;
    return true;
}

// src/shapes/sphere.rs:72-78
static inline B3 sphere_bounds(const Sphere& s) {
    B3 b = b3_from_points(V3(-s.radius, -s.radius, s.z_min), V3(s.radius, s.radius, s.z_max));
    return xf_bounds(s.render_from_object, b);
}

// src/core/square_matrix.rs:123-138
static inline float det3(const float m[3][3]) {
    float minor12 = difference_of_products(m[1][1], m[2][2], m[1][2], m[2][1]);
    float minor02 = difference_of_products(m[1][0], m[2][2], m[1][2], m[2][0]);
    float minor01 = difference_of_products(m[1][0], m[2][1], m[1][1], m[2][0]);
    return std::fmaf(m[0][2], minor01, difference_of_products(m[0][0], minor12, m[0][1], minor02));
}

// src/shapes/bilinear_patch.rs:506-612
static inline bool intersect_bilinear_patch(V3 ro, V3 rd, float t_max, V3 p00, V3 p10, V3 p01, V3 p11,
                                            PatchHit* out) {
    float a = dot(cross(p10 - p00, p01 - p11), rd);
    float c = dot(cross(p00 - ro, rd), p01 - p00);
    float b = dot(cross(p10 - ro, rd), p11 - p10) - (a + c);

    float u1, u2;
    if (a == 0.0f && b == 0.0f && c == 0.0f) return false;   // reference asserts (math.rs:44-47); treat as miss
    if (!solve_quadratic(a, b, c, &u1, &u2)) return false;

    float eps = gamma_n(10) * (max_component(vabs(ro)) + max_component(vabs(rd)) + max_component(vabs(p00)) +
                               max_component(vabs(p10)) + max_component(vabs(p01)) + max_component(vabs(p11)));

    float t = t_max, u = 0.0f, v = 0.0f;
    bool have = false;
    if (u1 >= 0.0f && u1 <= 1.0f) {
        V3 uo = vlerp(p00, p10, u1);
        V3 ud = vlerp(p01, p11, u1) - uo;
        V3 dlt = uo - ro;
        V3 perp = cross(rd, ud);
        float p2 = length_squared(perp);
        float mv[3][3] = {{dlt.x, rd.x, perp.x}, {dlt.y, rd.y, perp.y}, {dlt.z, rd.z, perp.z}};
        float mt[3][3] = {{dlt.x, ud.x, perp.x}, {dlt.y, ud.y, perp.y}, {dlt.z, ud.z, perp.z}};
        float v1 = det3(mv);
        float t1 = det3(mt);
        if (t1 > p2 * eps && (v1 >= 0.0f && v1 <= p2)) {
            u = u1; v = v1 / p2; t = t1 / p2; have = true;
        }
    }
    if (u2 >= 0.0f && u2 <= 1.0f && u2 != u1) {
        V3 uo = vlerp(p00, p10, u2);
        V3 ud = vlerp(p01, p11, u2) - uo;
        V3 dlt = uo - ro;
        V3 perp = cross(rd, ud);
        float p2 = length_squared(perp);
        float mv[3][3] = {{dlt.x, rd.x, perp.x}, {dlt.y, rd.y, perp.y}, {dlt.z, rd.z, perp.z}};
        float mt[3][3] = {{dlt.x, ud.x, perp.x}, {dlt.y, ud.y, perp.y}, {dlt.z, ud.z, perp.z}};
        float v2 = det3(mv);
        float t2 = det3(mt) / p2;                       // B10: divided before the epsilon compare
        if (t2 > p2 * eps && (v2 >= 0.0f && v2 <= p2) && t2 < t) {
            u = u2; v = v2 / p2; t = t2; have = true;
        }
    }
    if (t < t_max) {
        // the reference unwrap()s u, v here; `have` is always true when t < t_max
        (void)have;
        out->t = t; out->u = u; out->v = v;
        return true;
    }
    return false;
}

// src/shapes/bilinear_patch.rs:176-181
static inline B3 patch_bounds(const Patch& p) {
    return b3_union(b3_from_points(p.p00, p.p01), b3_from_points(p.p10, p.p11));
}

// ------------------------------------------------------------------ scene + hit record
struct Hit {
    uint32_t prim_id;   // 0xFFFFFFFF = miss
    uint32_t prim_slot; // index into Scene::prims (post-build order)
    float t, u, v;
    V3 p_obj; float phi;   // spheres only
};

struct LinearNode {      // aggregates.rs:719-732
    B3 bounds;
    bool leaf;
    uint32_t prims_offset; uint8_t num_prims;     // leaf
    uint32_t second_child; uint8_t axis;          // interior
};

struct Counters { uint64_t nodes_visited = 0, prims_tested = 0, exact_tests = 0, fallback_walks = 0; };

enum SplitMethod { SPLIT_SAH = 1, SPLIT_MIDDLE = 2, SPLIT_EQUAL_COUNTS = 3 };

struct Scene {
    std::vector<Sphere> spheres;
    std::vector<Patch> patches;
    std::vector<Prim> prims;          // reordered in place by the build, as in aggregates.rs:25-47
    std::vector<LinearNode> nodes;

    B3 prim_bounds(const Prim& p) const {
        return p.type == PRIM_SPHERE ? sphere_bounds(spheres[p.index]) : patch_bounds(patches[p.index]);
    }
    // Primitive::intersect for SimplePrimitive (single.rs:88-99) — returns the geometric hit only;
    // the SurfaceInteraction is rebuilt from it by ref_render (identical values, built once).
    bool prim_intersect(const Prim& p, V3 ro, V3 rd, float t_max, Hit* h) const {
        if (p.type == PRIM_SPHERE) {
            QuadricHit q;
            if (!sphere_basic_intersect(spheres[p.index], ro, rd, t_max, &q)) return false;
            h->t = q.t_hit; h->p_obj = q.p_obj; h->phi = q.phi; h->u = 0; h->v = 0;
        } else {
            const Patch& pt = patches[p.index];
            PatchHit ph;
            if (!intersect_bilinear_patch(ro, rd, t_max, pt.p00, pt.p10, pt.p01, pt.p11, &ph)) return false;
            h->t = ph.t; h->u = ph.u; h->v = ph.v; h->phi = 0; h->p_obj = V3();
        }
        h->prim_id = p.id;
        return true;
    }

    // ---------------------------------------------------------------- build (aggregates.rs:49-271)
    struct BuildNode {
        B3 bounds; bool leaf; uint32_t first; uint8_t n; uint8_t axis; int left, right;
    };
    std::vector<BuildNode> bn;

    int build_recursive(size_t lo, size_t hi, SplitMethod method, int max_prims, size_t first_off) {
        size_t n = hi - lo;
        B3 bounds = prim_bounds(prims[lo]);
        for (size_t i = lo + 1; i < hi; ++i) bounds = b3_union(bounds, prim_bounds(prims[i]));
        auto leaf = [&]() {
            // B6: n_primitives is u8 — `prims.len() as u8` truncates silently
            bn.push_back({bounds, true, (uint32_t)first_off, (uint8_t)n, 0, -1, -1});
            return (int)bn.size() - 1;
        };
        if (b3_surface_area(bounds) == 0.0f || n == 1) return leaf();
        B3 cb = B3::empty();
        for (size_t i = lo; i < hi; ++i) cb = b3_union_point(cb, b3_centroid(prim_bounds(prims[i])));
        int dim = b3_max_extent(cb);
        if (cb.hi[dim] == cb.lo[dim]) return leaf();

        auto cen = [&](const Prim& p) { return b3_centroid(prim_bounds(p))[dim]; };
        auto equal_counts = [&]() {    // aggregates.rs:178-186 (select_nth_unstable ~ nth_element;
            size_t mid = n / 2;        // arrangement inside each half is implementation-defined in both)
            std::nth_element(prims.begin() + lo, prims.begin() + lo + mid, prims.begin() + hi,
                             [&](const Prim& a, const Prim& b) { return cen(a) < cen(b); });
            return mid;
        };
        // itertools::partition (swap-from-both-ends, unstable)
        auto it_partition = [&](auto pred) {
            size_t split = 0, f = lo, bk = hi;
            while (f < bk) {
                if (!pred(prims[f])) {
                    bool found = false;
                    while (bk > f + 1) {
                        --bk;
                        if (pred(prims[bk])) { std::swap(prims[f], prims[bk]); found = true; break; }
                    }
                    if (!found) return split;
                }
                ++split; ++f;
            }
            return split;
        };
        size_t mid = 0;
        if (method == SPLIT_MIDDLE) {
            float dmid = (cb.lo[dim] + cb.hi[dim]) / 2.0f;
            mid = it_partition([&](const Prim& p) { return cen(p) < dmid; });
            if (!(mid >= 1 && mid < n)) mid = equal_counts();
        } else if (method == SPLIT_EQUAL_COUNTS) {
            mid = equal_counts();
        } else {   // SAH, aggregates.rs:189-271
            if (n <= 2) {
                mid = equal_counts();
            } else {
                const int NB = 12, NS = 11;
                struct Bucket { size_t count = 0; B3 b = B3::empty(); } buckets[NB];
                // bounds.rs:509-529: Bounds3f::offset computes the x offset for every component
                auto bucket_idx = [&](const Prim& p) {
                    V3 cp = b3_centroid(prim_bounds(p));
                    float ax = cp.x - cb.lo.x;
                    float off = ax != 0.0f ? ax / (cb.hi.x - cb.lo.x) : 0.0f;
                    float bo = (float)NB * off;
                    size_t bi = bo > 0.0f ? (size_t)bo : 0;   // `as usize` saturates
                    // B4: reference clamps with .min(12) and would panic on index 12; the oracle clamps to 11
                    return std::min(bi, (size_t)(NB - 1));
                };
                for (size_t i = lo; i < hi; ++i) {
                    size_t bi = bucket_idx(prims[i]);
                    buckets[bi].count++;
                    buckets[bi].b = b3_union(buckets[bi].b, prim_bounds(prims[i]));
                }
                float costs[NS] = {0};
                size_t cnt = 0; B3 bb = B3::empty();
                for (int i = 0; i < NS; ++i) {
                    bb = b3_union(bb, buckets[i].b); cnt += buckets[i].count;
                    costs[i] += (float)cnt * b3_surface_area(bb);
                }
                cnt = 0; bb = B3::empty();
                for (int i = NS; i >= 1; --i) {
                    bb = b3_union(bb, buckets[i].b); cnt += buckets[i].count;
                    costs[i - 1] += (float)cnt * b3_surface_area(bb);
                }
                int best = 0;
                for (int i = 1; i < NS; ++i) if (costs[i] < costs[best]) best = i;   // min_by keeps first min
                float leaf_cost = (float)n;
                float min_cost = 0.5f + costs[best] / b3_surface_area(bounds);
                if (min_cost < leaf_cost || n > (size_t)max_prims) {
                    mid = it_partition([&](const Prim& p) { return bucket_idx(p) <= (size_t)best; });
                } else {
                    return leaf();
                }
            }
        }
        int l = build_recursive(lo, lo + mid, method, max_prims, first_off);
        int r = build_recursive(lo + mid, hi, method, max_prims, first_off + mid);
        BuildNode in;
        in.bounds = b3_union(bn[l].bounds, bn[r].bounds);
        in.leaf = false; in.first = 0; in.n = 0; in.axis = (uint8_t)dim; in.left = l; in.right = r;
        bn.push_back(in);
        return (int)bn.size() - 1;
    }
    void flatten(int idx) {   // aggregates.rs:484-529
        const BuildNode b = bn[idx];
        LinearNode ln;
        ln.bounds = b.bounds; ln.leaf = b.leaf; ln.prims_offset = b.first; ln.num_prims = b.n;
        ln.second_child = 0; ln.axis = b.axis;
        nodes.push_back(ln);
        if (!b.leaf) {
            size_t me = nodes.size() - 1;
            flatten(b.left);
            nodes[me].second_child = (uint32_t)nodes.size();
            flatten(b.right);
        }
    }
    void build(SplitMethod method, int max_prims_in_node) {
        nodes.clear(); bn.clear();
        if (prims.empty()) return;
        // degenerate SAH recursion (mid == 0 or n) cannot terminate in the reference either;
        // it is not reachable for the scenes the tests use.
        int root = build_recursive(0, prims.size(), method, max_prims_in_node, 0);
        flatten(root);
        bn.clear(); bn.shrink_to_fit();
    }

    // debugging aid of the tests: when set (per thread), every ray traced is appended as 10 floats
    // {o xyz, d xyz, t_max, kind (0 closest / 1 any), bits of the hit id or of the occlusion flag, t}
    static std::vector<float>*& ray_log() { static thread_local std::vector<float>* log = nullptr; return log; }
    bool intersect(V3 ro, V3 rd, float t_max, Hit* best, Counters* ctr = nullptr) const {
        const bool any = intersect_impl(ro, rd, t_max, best, ctr);
        if (std::vector<float>* log = ray_log()) {
            float idf; std::memcpy(&idf, &best->prim_id, 4);
            const float rec[10] = {ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, t_max, 0.0f, idf, any ? best->t : t_max};
            log->insert(log->end(), rec, rec + 10);
        }
        return any;
    }
    bool intersect_p(V3 ro, V3 rd, float t_max, Counters* ctr = nullptr) const {
        const bool occ = intersect_p_impl(ro, rd, t_max, ctr);
        if (std::vector<float>* log = ray_log()) {
            const uint32_t o = occ ? 1u : 0u; float of; std::memcpy(&of, &o, 4);
            const float rec[10] = {ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, t_max, 1.0f, of, 0.0f};
            log->insert(log->end(), rec, rec + 10);
        }
        return occ;
    }
    // ---------------------------------------------------------------- traversal (aggregates.rs:537-655)
    bool intersect_impl(V3 ro, V3 rd, float t_max, Hit* best, Counters* ctr = nullptr) const {
        best->prim_id = 0xFFFFFFFFu;
        if (nodes.empty()) return false;
        float inv[3] = {1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
        bool neg[3] = {inv[0] < 0.0f, inv[1] < 0.0f, inv[2] < 0.0f};
        bool any = false;
        std::vector<uint32_t> stack; stack.reserve(64);     // B7: a heap allocation per ray, kept
        stack.push_back(0);
        while (!stack.empty()) {
            uint32_t idx = stack.back(); stack.pop_back();
            const LinearNode& nd = nodes[idx];
            if (ctr) ctr->nodes_visited++;
            if (!b3_intersect_p(nd.bounds, ro, rd, t_max)) continue;
            if (nd.leaf) {
                for (uint32_t i = nd.prims_offset; i < nd.prims_offset + nd.num_prims; ++i) {
                    Hit h;
                    if (ctr) ctr->prims_tested++;
                    if (prim_intersect(prims[i], ro, rd, t_max, &h)) {
                        t_max = h.t; h.prim_slot = i; *best = h; any = true;
                    }
                }
            } else if (neg[nd.axis]) {
                stack.push_back(idx + 1); stack.push_back(nd.second_child);
            } else {
                stack.push_back(nd.second_child); stack.push_back(idx + 1);
            }
        }
        return any;
    }
    bool intersect_p_impl(V3 ro, V3 rd, float t_max, Counters* ctr = nullptr) const {
        if (nodes.empty()) return false;
        float inv[3] = {1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
        bool neg[3] = {inv[0] < 0.0f, inv[1] < 0.0f, inv[2] < 0.0f};
        std::vector<uint32_t> stack; stack.reserve(64);
        stack.push_back(0);
        while (!stack.empty()) {
            uint32_t idx = stack.back(); stack.pop_back();
            const LinearNode& nd = nodes[idx];
            if (ctr) ctr->nodes_visited++;
            if (!b3_intersect_p(nd.bounds, ro, rd, t_max)) continue;
            if (nd.leaf) {
                for (uint32_t i = nd.prims_offset; i < nd.prims_offset + nd.num_prims; ++i) {
                    Hit h;
                    if (ctr) ctr->prims_tested++;
                    if (prim_intersect(prims[i], ro, rd, t_max, &h)) return true;
                }
            } else if (neg[nd.axis]) {
                stack.push_back(idx + 1); stack.push_back(nd.second_child);
            } else {
                stack.push_back(nd.second_child); stack.push_back(idx + 1);
            }
        }
        return false;
    }
    // All-primitives cross-check (no BVH): same acceptance rule, insertion order.
    bool brute_force(V3 ro, V3 rd, float t_max, Hit* best) const {
        best->prim_id = 0xFFFFFFFFu;
        bool any = false;
        for (size_t i = 0; i < prims.size(); ++i) {
            Hit h;
            if (prim_intersect(prims[i], ro, rd, t_max, &h)) { t_max = h.t; h.prim_slot = (uint32_t)i; *best = h; any = true; }
        }
        return any;
    }
    B3 bounds() const { return nodes.empty() ? B3::empty() : nodes[0].bounds; }
};

}  // namespace ora
