// Host-side construction of the wide (8-ary) quantised BVH.
//
// Reference counterpart: BVHAggregate::new / build_recursive / flatten_bvh
// (primitives/aggregates.rs:25-271,476-529).  The reference builds a binary tree with 64-byte
// nodes; closest-hit results do not depend on the tree shape (the leaf tests are the reference's
// arithmetic, the box tests are conservative), so the GPU tree is free to be wide and quantised.
// Pipeline: per-type binned-SAH binary build -> greedy collapse to 8 children (largest surface
// area first) -> octant slot assignment (so slot ^ ray_octant is a front-to-back order) ->
// outward-rounded 8-bit quantisation -> leaf records in traversal order.
#include "pm_build.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>

namespace pm {

static inline Box box_union(const Box& a, const Box& b) {
    Box r;
    for (int i = 0; i < 3; ++i) { r.lo[i] = std::fmin(a.lo[i], b.lo[i]); r.hi[i] = std::fmax(a.hi[i], b.hi[i]); }
    return r;
}
static inline Box box_empty() {
    Box b;
    for (int i = 0; i < 3; ++i) { b.lo[i] = std::numeric_limits<float>::max(); b.hi[i] = -std::numeric_limits<float>::max(); }
    return b;
}
static inline float box_area(const Box& b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}

struct BinNode { Box b; int left = -1, right = -1; uint32_t first = 0, count = 0; };

struct BinBuilder {
    const std::vector<Box>& pb;       // per-primitive boxes (indexed by local prim index)
    std::vector<float> cen;           // centroids, 3 per prim
    std::vector<uint32_t>& order;     // permutation being partitioned
    int max_leaf;
    // unified build over several primitive types (round 3): a leaf holds primitives of ONE type, at most max_leaf_of[type] of them
    const std::vector<uint8_t>* types = nullptr;
    int max_leaf_of[3] = {0, 0, 0};
    // may order[lo, hi) become a leaf?
    bool leaf_ok(uint32_t lo, uint32_t hi) const {
        const uint32_t n = hi - lo;
        if (!types) return n <= (uint32_t)max_leaf;
        const uint8_t t = (*types)[order[lo]];
        if (n > (uint32_t)max_leaf_of[t]) return false;
        for (uint32_t i = lo + 1; i < hi; ++i) if ((*types)[order[i]] != t) return false;
        return true;
    }
    BinBuilder(const std::vector<Box>& b, std::vector<uint32_t>& o, int ml) : pb(b), order(o), max_leaf(ml) {
        cen.resize(b.size() * 3);
        for (size_t i = 0; i < b.size(); ++i)
            for (int k = 0; k < 3; ++k) cen[3 * i + k] = 0.5f * b[i].lo[k] + 0.5f * b[i].hi[k];
    }
    // builds nodes for order[lo,hi) into `out` (root is out[base]); returns root index
    int build(std::vector<BinNode>& out, uint32_t lo, uint32_t hi, int depth) {
        BinNode nd;
        nd.b = box_empty();
        Box cb = box_empty();
        for (uint32_t i = lo; i < hi; ++i) {
            uint32_t p = order[i];
            nd.b = box_union(nd.b, pb[p]);
            for (int k = 0; k < 3; ++k) {
                cb.lo[k] = std::fmin(cb.lo[k], cen[3 * p + k]);
                cb.hi[k] = std::fmax(cb.hi[k], cen[3 * p + k]);
            }
        }
        uint32_t n = hi - lo;
        int me = (int)out.size();
        out.push_back(nd);
        auto make_leaf = [&]() { out[me].first = lo; out[me].count = n; return me; };
        if (n == 1) return make_leaf();
        int dim = 0;
        float ext[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
        if (ext[1] > ext[dim]) dim = 1;
        if (ext[2] > ext[dim]) dim = 2;
        uint32_t mid = lo;
        bool split_done = false;
        if (ext[dim] > 0.0f) {
            const int NB = 16;
            uint32_t cnt[NB] = {0};
            Box bb[NB];
            for (int i = 0; i < NB; ++i) bb[i] = box_empty();
            float k1 = (float)NB * (1.0f - 1e-6f) / ext[dim];
            auto bin_of = [&](uint32_t p) {
                int b = (int)((cen[3 * p + dim] - cb.lo[dim]) * k1);
                return b < 0 ? 0 : (b >= NB ? NB - 1 : b);
            };
            for (uint32_t i = lo; i < hi; ++i) { int b = bin_of(order[i]); cnt[b]++; bb[b] = box_union(bb[b], pb[order[i]]); }
            float right_area[NB]; uint32_t right_cnt[NB];
            Box acc = box_empty(); uint32_t c = 0;
            for (int i = NB - 1; i >= 1; --i) { acc = box_union(acc, bb[i]); c += cnt[i]; right_area[i] = box_area(acc); right_cnt[i] = c; }
            acc = box_empty(); c = 0;
            float best = std::numeric_limits<float>::infinity(); int best_i = -1;
            for (int i = 0; i < NB - 1; ++i) {
                acc = box_union(acc, bb[i]); c += cnt[i];
                if (c == 0 || right_cnt[i + 1] == 0) continue;
                float cost = (float)c * box_area(acc) + (float)right_cnt[i + 1] * right_area[i + 1];
                if (cost < best) { best = cost; best_i = i; }
            }
            if (best_i >= 0) {
                float leaf_cost = (float)n * box_area(nd.b);
                // traversal cost 1 node ~ 1 prim test; only keep a leaf when it is small enough and cheaper
                if (leaf_ok(lo, hi) && leaf_cost <= best + box_area(nd.b)) return make_leaf();
                auto it = std::partition(order.begin() + lo, order.begin() + hi, [&](uint32_t p) { return bin_of(p) <= best_i; });
                mid = (uint32_t)(it - order.begin());
                split_done = mid > lo && mid < hi;
            }
        }
        if (!split_done) {
            if (leaf_ok(lo, hi)) return make_leaf();
            mid = lo + n / 2;   // coincident centroids: split the index range
            std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                             [&](uint32_t a, uint32_t b) { return cen[3 * a + dim] < cen[3 * b + dim]; });
        }
        int l, r;
        if (n > 200000 && depth < 4) {
            // large ranges: build the two halves concurrently into private vectors, then splice
            std::vector<BinNode> lv, rv;
            auto fl = std::async(std::launch::async, [&]() { return build(lv, lo, mid, depth + 1); });
            int rr = build(rv, mid, hi, depth + 1);
            int lr = fl.get();
            int lbase = (int)out.size();
            for (auto& x : lv) { if (x.left >= 0) { x.left += lbase; x.right += lbase; } out.push_back(x); }
            int rbase = (int)out.size();
            for (auto& x : rv) { if (x.left >= 0) { x.left += rbase; x.right += rbase; } out.push_back(x); }
            l = lr + lbase; r = rr + rbase;
        } else {
            l = build(out, lo, mid, depth + 1);
            r = build(out, mid, hi, depth + 1);
        }
        out[me].left = l; out[me].right = r;
        return me;
    }
};

// --- quantisation ---------------------------------------------------------------------------
static inline float scale_of(uint8_t e) { uint32_t u = (uint32_t)e << 23; float f; std::memcpy(&f, &u, 4); return f; }

static void quantise_node(PmWideNode& wn, const Box& nb, const Box* cb, const bool* used) {
    for (int k = 0; k < 3; ++k) {
        wn.o[k] = nb.lo[k];
        float ext = nb.hi[k] - nb.lo[k];
        int e = 127 - 126;   // smallest normal scale
        if (ext > 0.0f) {
            int ex; std::frexp(ext / 255.0f, &ex);   // ext/255 = m * 2^ex, m in [0.5,1)  -> 2^ex >= ext/255
            e = ex + 127;
        }
        if (e < 1) e = 1;
        for (;; ++e) {   // verify every child fits in 0..255 with outward rounding, else coarsen
            float s = scale_of((uint8_t)e);
            bool ok = true;
            for (int c = 0; c < 8 && ok; ++c) {
                if (!used[c]) continue;
                float qh = std::ceil((cb[c].hi[k] - wn.o[k]) / s);
                if (!(qh <= 254.0f)) ok = false;   // leave one step of head-room for the fix-up loop
            }
            if (ok || e >= 254) break;
        }
        wn.e[k] = (uint8_t)e;
        float s = scale_of(wn.e[k]);
        for (int c = 0; c < 8; ++c) {
            if (!used[c]) { wn.qlo[k][c] = 255; wn.qhi[k][c] = 0; continue; }   // inverted box: never hit
            int ql = (int)std::floor((cb[c].lo[k] - wn.o[k]) / s);
            int qh = (int)std::ceil((cb[c].hi[k] - wn.o[k]) / s);
            ql = std::max(0, std::min(255, ql));
            qh = std::max(0, std::min(255, qh));
            // conservative against the EXACT plane o + q*s (evaluated in double: exact), which implies it for the
            // f32-rounded decode too; the kernel's pre-scaled slab test relies on the exact form
            while (ql > 0 && (double)ql * (double)s + (double)wn.o[k] > (double)cb[c].lo[k]) --ql;
            while (qh < 255 && (double)qh * (double)s + (double)wn.o[k] < (double)cb[c].hi[k]) ++qh;
            wn.qlo[k][c] = (uint8_t)ql;
            wn.qhi[k][c] = (uint8_t)qh;
        }
    }
}

struct Collapser {
    const std::vector<BinNode>& bn;
    const std::vector<uint32_t>& order;   // local prim order of this tree
    int type;                             // primitive type of the whole tree, or ...
    Accel& acc;
    const HostScene& sc;
    const std::vector<uint32_t>& local_to_global;   // local prim index -> index into the per-type array of its type
    const std::vector<uint8_t>* types = nullptr;    // ... (unified tree) the type of every local prim

    int type_of(uint32_t local) const { return types ? (int)(*types)[local] : type; }
    int leaf_type(int b) const { return type_of(order[bn[b].first]); }      // (leaves hold one type)

    void emit_record(uint32_t local) {
        uint32_t gi = local_to_global[local];
        const int ty = type_of(local);
        if (ty == PM_PRIM_TRI) {
            PmTriRecord r{};
            const HostTri& t = sc.tris[gi];
            std::memcpy(r.v0, t.v[0], 12); std::memcpy(r.v1, t.v[1], 12); std::memcpy(r.v2, t.v[2], 12);
            r.prim_id = t.prim_id;
            size_t off = acc.prims.size(); acc.prims.resize(off + sizeof(r)); std::memcpy(&acc.prims[off], &r, sizeof(r));
        } else if (ty == PM_PRIM_PATCH) {
            PmPatchRecord r{};
            const HostPatch& p = sc.patches[gi];
            std::memcpy(r.p00, p.p[0], 12); std::memcpy(r.p10, p.p[1], 12); std::memcpy(r.p01, p.p[2], 12); std::memcpy(r.p11, p.p[3], 12);
            r.prim_id = p.prim_id; r.flags = p.flip_normal ? 1u : 0u;
            size_t off = acc.prims.size(); acc.prims.resize(off + sizeof(r)); std::memcpy(&acc.prims[off], &r, sizeof(r));
        } else {
            PmSphereRecord r{};
            const HostSphere& s = sc.spheres[gi];
            std::memcpy(r.ofr, s.object_from_render, 48);
            r.radius = s.radius; r.z_min = s.z_min; r.z_max = s.z_max; r.phi_max = s.phi_max;
            r.prim_id = s.prim_id; r.flags = s.reverse_orientation ? 1u : 0u;
            r.m_norm = 0.0f; r.t_norm = 0.0f;
            for (int row = 0; row < 3; ++row) {
                const float* m = s.object_from_render + 4 * row;
                r.m_norm = std::fmax(r.m_norm, std::fabs(m[0]) + std::fabs(m[1]) + std::fabs(m[2]));
                r.t_norm = std::fmax(r.t_norm, std::fabs(m[3]));
            }
            size_t off = acc.prims.size(); acc.prims.resize(off + sizeof(r)); std::memcpy(&acc.prims[off], &r, sizeof(r));
        }
    }

    // One unit of work: fill wide node `wide` from the binary subtree rooted at `bin`, or - unified tree only - from a GROUP of
    // binary leaves of one type (`leaves`), see fill().
    struct Work { uint32_t wide; int bin; std::vector<int> leaves; };
    // a child of the wide node being filled: a binary node (interior or leaf), or a group of leaves that gets a node of its own
    struct Child { Box b; int bin; std::vector<int> group; bool interior() const { return !group.empty() || bin_interior; } bool bin_interior; };

    void fill(const Work& w, std::vector<Work>& queue) {
        int kids[8]; int nk = 0;
        Box nb;
        if (!w.leaves.empty()) {            // a group: its leaves are the children, nothing to expand
            nb = box_empty();
            for (int l : w.leaves) { kids[nk++] = l; nb = box_union(nb, bn[l].b); }
        } else {
            const int b = w.bin;
            nb = bn[b].b;
            if (bn[b].left < 0) { kids[nk++] = b; }
            else { kids[nk++] = bn[b].left; kids[nk++] = bn[b].right; }
            // greedy expansion: open the interior child with the largest surface area
            for (;;) {
                if (nk >= 8) break;
                int best = -1; float ba = -1.0f;
                for (int i = 0; i < nk; ++i)
                    if (bn[kids[i]].left >= 0) { float a = box_area(bn[kids[i]].b); if (a > ba) { ba = a; best = i; } }
                if (best < 0) break;
                int k = kids[best];
                kids[best] = bn[k].left; kids[nk++] = bn[k].right;
            }
        }
        // A wide node has ONE record array, i.e. its leaves share a primitive type.  In a unified tree over several types the
        // leaves of the type most of them have stay leaves of this node; the leaves of every other type move, together, into a
        // node of their own one level down (a Work item with `leaves`): one more node step for them, in exchange for ONE tree
        // that follows the geometry instead of one tree per type that every ray has to walk.
        int node_type = type;
        std::vector<Child> ch;
        if (types) {
            int n_of[3] = {0, 0, 0};
            for (int i = 0; i < nk; ++i) if (bn[kids[i]].left < 0) n_of[leaf_type(kids[i])]++;
            node_type = 0;
            for (int t = 1; t < 3; ++t) if (n_of[t] > n_of[node_type]) node_type = t;
            std::vector<int> minority[3];
            for (int i = 0; i < nk; ++i) {
                const int k = kids[i];
                if (bn[k].left < 0 && leaf_type(k) != node_type) minority[leaf_type(k)].push_back(k);
                else ch.push_back(Child{bn[k].b, k, {}, bn[k].left >= 0});
            }
            for (int t = 0; t < 3; ++t) {
                if (minority[t].empty()) continue;
                Box gb = box_empty();
                for (int l : minority[t]) gb = box_union(gb, bn[l].b);
                ch.push_back(Child{gb, -1, minority[t], false});
            }
        } else {
            for (int i = 0; i < nk; ++i) ch.push_back(Child{bn[kids[i]].b, kids[i], {}, bn[kids[i]].left >= 0});
        }
        const int nc = (int)ch.size();
        // octant slot assignment: maximise dot(child centroid - node centroid, slot direction), greedily
        float pc[3] = {0.5f * (nb.lo[0] + nb.hi[0]), 0.5f * (nb.lo[1] + nb.hi[1]), 0.5f * (nb.lo[2] + nb.hi[2])};
        float cost[8][8];
        for (int i = 0; i < nc; ++i) {
            const Box& cbx = ch[i].b;
            float d[3] = {0.5f * (cbx.lo[0] + cbx.hi[0]) - pc[0], 0.5f * (cbx.lo[1] + cbx.hi[1]) - pc[1], 0.5f * (cbx.lo[2] + cbx.hi[2]) - pc[2]};
            for (int s = 0; s < 8; ++s)
                cost[i][s] = ((s & 1) ? d[0] : -d[0]) + ((s & 2) ? d[1] : -d[1]) + ((s & 4) ? d[2] : -d[2]);
        }
        int slot_of[8]; bool child_done[8] = {false}, slot_used[8] = {false};
        for (int it = 0; it < nc; ++it) {
            int bi = -1, bs = -1; float bc = -std::numeric_limits<float>::infinity();
            for (int i = 0; i < nc; ++i) if (!child_done[i])
                for (int s = 0; s < 8; ++s) if (!slot_used[s] && cost[i][s] > bc) { bc = cost[i][s]; bi = i; bs = s; }
            slot_of[bi] = bs; child_done[bi] = true; slot_used[bs] = true;
        }
        int in_slot[8]; for (int s = 0; s < 8; ++s) in_slot[s] = -1;
        for (int i = 0; i < nc; ++i) in_slot[slot_of[i]] = i;

        PmWideNode wn{};
        Box cb[8]; bool used[8];
        uint8_t imask = 0; uint32_t n_int = 0;
        for (int s = 0; s < 8; ++s) {
            used[s] = in_slot[s] >= 0;
            if (used[s]) { cb[s] = ch[in_slot[s]].b; if (ch[in_slot[s]].interior()) { imask |= (uint8_t)(1u << s); ++n_int; } }
        }
        quantise_node(wn, nb, cb, used);
        wn.imask = imask;
        wn.child_base = (uint32_t)acc.nodes.size();
        uint32_t prim_base_units = (uint32_t)(acc.prims.size() / 16);
        wn.prim_base = prim_base_units | ((uint32_t)node_type << 30);
        uint32_t rec_off = 0;
        // reserve interior children contiguously, in slot order
        acc.nodes.resize(acc.nodes.size() + n_int);
        uint32_t rank = 0;
        for (int s = 0; s < 8; ++s) {
            if (!used[s]) { wn.meta[s] = 0; continue; }
            const Child& c = ch[in_slot[s]];
            if (c.interior()) {
                wn.meta[s] = 0;
                queue.push_back(Work{wn.child_base + rank, c.bin, c.group});
                ++rank;
            } else {
                const BinNode& l = bn[c.bin];
                wn.meta[s] = (uint8_t)((l.count << 5) | rec_off);
                for (uint32_t i = 0; i < l.count; ++i) emit_record(order[l.first + i]);
                rec_off += l.count;
            }
        }
        acc.nodes[w.wide] = wn;
    }

    void run(uint32_t root_wide, int root_bin) {
        std::vector<Work> queue;
        queue.push_back(Work{root_wide, root_bin, {}});
        size_t head = 0;
        while (head < queue.size()) {
            const Work w = queue[head++];
            fill(w, queue);
        }
    }
};

// PBRT_MI355_UNIFIED_TREE=0: round 2's tree per primitive type under a join node (A/B and tests)
static bool unified_tree() {
    const char* e = std::getenv("PBRT_MI355_UNIFIED_TREE");
    return !(e && std::atoi(e) == 0);
}

int build_accel(const HostScene& sc, int max_leaf, Accel& acc, std::string& err) {
    auto t0 = std::chrono::steady_clock::now();
    acc.nodes.clear(); acc.prims.clear();
    if (max_leaf < 1) max_leaf = 1;
    if (max_leaf > PM_MAX_LEAF) max_leaf = PM_MAX_LEAF;
    size_t n_total = sc.spheres.size() + sc.patches.size() + sc.tris.size();
    if (n_total == 0) { err = "no primitives"; return 1; }

    const int n_types = (sc.spheres.empty() ? 0 : 1) + (sc.patches.empty() ? 0 : 1) + (sc.tris.empty() ? 0 : 1);
    const bool unified = n_types > 1 && unified_tree();
    struct TypeTree { int type; std::vector<Box> boxes; std::vector<uint32_t> order, l2g; std::vector<BinNode> bn; int root; };
    std::vector<TypeTree> trees;
    auto add_type = [&](int type, size_t n, auto box_fn) {
        if (n == 0) return;
        trees.emplace_back();
        TypeTree& t = trees.back();
        t.type = type; t.boxes.resize(n); t.order.resize(n); t.l2g.resize(n);
        for (size_t i = 0; i < n; ++i) { t.boxes[i] = box_fn(i); t.order[i] = (uint32_t)i; t.l2g[i] = (uint32_t)i; }
        // Triangles always get single-primitive leaves: in a 64-lane wave every extra primitive of a leaf is an extra
        // pre-filter trip for all lanes, which costs more than the node it saves (measured, profiles/r01/leaf_sweep.log:
        // leaves <= 4 / <= 2 / 1 -> 2218 / 2233 / 2339 Mrays/s on C3, 214 / 221 / 252 Msamples/s on the C4-shaped scene).
        // max_prims_in_node stays the upper bound for spheres and patches, whose tests are not pre-filtered in the walk.
        if (unified) return;            // (one tree over all types below: only the boxes are needed)
        BinBuilder bb(t.boxes, t.order, type == PM_PRIM_TRI ? 1 : max_leaf);
        t.bn.reserve(2 * n);
        t.root = bb.build(t.bn, 0, (uint32_t)n, 0);
    };
    add_type(PM_PRIM_SPHERE, sc.spheres.size(), [&](size_t i) { return sc.spheres[i].bounds; });
    add_type(PM_PRIM_PATCH, sc.patches.size(), [&](size_t i) { return sc.patches[i].bounds; });
    add_type(PM_PRIM_TRI, sc.tris.size(), [&](size_t i) { return sc.tris[i].bounds; });

    Box total = box_empty();
    for (auto& t : trees) for (const Box& b : t.boxes) total = box_union(total, b);
    acc.bounds = total;

    if (trees.size() == 1) {
        acc.nodes.resize(1);
        Collapser c{trees[0].bn, trees[0].order, trees[0].type, acc, sc, trees[0].l2g};
        c.run(0, trees[0].root);
    } else if (unified) {
        // ONE tree over all primitives (round 3): binned SAH on the boxes alone, leaves of one type each (BinBuilder::leaf_ok),
        // wide nodes whose leaves share a type (Collapser::fill).  Round 2 built a tree per type under a join node; every ray
        // then walked every type's tree (a million triangles + 512 spheres: ~15 % more node steps; half triangles, half patches
        // in the same space: twice the steps).
        std::vector<Box> boxes; std::vector<uint32_t> order, l2g; std::vector<uint8_t> types;
        for (auto& t : trees) {
            for (size_t i = 0; i < t.boxes.size(); ++i) { boxes.push_back(t.boxes[i]); l2g.push_back((uint32_t)i); types.push_back((uint8_t)t.type); }
        }
        order.resize(boxes.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
        BinBuilder bb(boxes, order, max_leaf);
        bb.types = &types;
        bb.max_leaf_of[PM_PRIM_SPHERE] = max_leaf; bb.max_leaf_of[PM_PRIM_PATCH] = max_leaf; bb.max_leaf_of[PM_PRIM_TRI] = 1;
        std::vector<BinNode> bn; bn.reserve(2 * boxes.size());
        const int root = bb.build(bn, 0, (uint32_t)boxes.size(), 0);
        acc.nodes.resize(1);
        Collapser c{bn, order, 0, acc, sc, l2g, &types};
        c.run(0, root);
    } else {
        // join node: one interior child per primitive type
        acc.nodes.resize(1 + trees.size());
        PmWideNode wn{};
        Box cb[8]; bool used[8] = {false};
        for (size_t i = 0; i < trees.size(); ++i) { cb[i] = trees[i].bn[trees[i].root].b; used[i] = true; }
        quantise_node(wn, total, cb, used);
        wn.imask = (uint8_t)((1u << trees.size()) - 1);
        wn.child_base = 1; wn.prim_base = 0;
        acc.nodes[0] = wn;
        for (size_t i = 0; i < trees.size(); ++i) {
            Collapser c{trees[i].bn, trees[i].order, trees[i].type, acc, sc, trees[i].l2g};
            c.run(1 + (uint32_t)i, trees[i].root);
        }
    }
    acc.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

// Sweep order = node index, then slot, then record: the CPU checker (oracle/wide_walk.h) derives the same table from the
// exported buffers.
void build_flat_table(Accel& acc, uint32_t max_prims) {
    acc.flat_tris.clear(); acc.flat_units.clear();
    acc.flat_n[0] = acc.flat_n[1] = acc.flat_n[2] = 0;
    size_t total = 0;
    for (const PmWideNode& nd : acc.nodes)
        for (int s = 0; s < 8; ++s) if (!((nd.imask >> s) & 1)) total += nd.meta[s] >> 5;
    if (total == 0 || total > max_prims) return;
    std::vector<uint32_t> sph, pat;
    for (const PmWideNode& nd : acc.nodes) {
        const uint32_t type = nd.prim_base >> 30, pbase = nd.prim_base & 0x3FFFFFFFu;
        const uint32_t units = type == PM_PRIM_TRI ? PM_UNITS_TRI : (type == PM_PRIM_PATCH ? PM_UNITS_PATCH : PM_UNITS_SPHERE);
        for (int s = 0; s < 8; ++s) {
            if ((nd.imask >> s) & 1) continue;
            const uint32_t cnt = nd.meta[s] >> 5, off = nd.meta[s] & 31u;
            for (uint32_t j = 0; j < cnt; ++j) {
                const uint32_t unit = pbase + (off + j) * units;
                if (type == PM_PRIM_SPHERE) sph.push_back(unit);
                else if (type == PM_PRIM_PATCH) pat.push_back(unit);
                else {
                    const float* f = reinterpret_cast<const float*>(&acc.prims[(size_t)unit * 16]);
                    PmFlatTri t{};
                    float L = 0.0f;
                    for (int k = 0; k < 3; ++k) {
                        t.v0[k] = f[k];
                        t.e1[k] = f[3 + k] - f[k];
                        t.e2[k] = f[6 + k] - f[k];
                        L = std::fmax(L, std::fmax(std::fabs(t.e1[k]), std::fabs(t.e2[k])));
                    }
                    t.n[0] = (float)((double)t.e1[1] * (double)t.e2[2] - (double)t.e1[2] * (double)t.e2[1]);
                    t.n[1] = (float)((double)t.e1[2] * (double)t.e2[0] - (double)t.e1[0] * (double)t.e2[2]);
                    t.n[2] = (float)((double)t.e1[0] * (double)t.e2[1] - (double)t.e1[1] * (double)t.e2[0]);
                    t.L = L; t.unit = unit;
                    t.c1 = std::fmax(std::fmax(std::fabs(t.v0[0]), std::fabs(t.v0[1])), std::fabs(t.v0[2])) + L;
                    t.kll = (7.62939453125e-06f * L) * L;
                    acc.flat_tris.push_back(t);
                }
            }
        }
    }
    // ---- order (pm_layout.h): groups of bit-equal v0 in the order of their first member; inside a group, chains along shared
    // edges (the next entry is the first unplaced one whose e1 is the current e2; when there is none, the first unplaced one)
    {
        const std::vector<PmFlatTri> in = acc.flat_tris;
        const size_t n = in.size();
        auto same3 = [](const float* a, const float* b) { return std::memcmp(a, b, 3 * sizeof(float)) == 0; };
        std::vector<char> used(n, 0);
        std::vector<PmFlatTri> out;
        out.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            if (used[i]) continue;
            std::vector<size_t> group;
            for (size_t j = i; j < n; ++j) if (!used[j] && same3(in[j].v0, in[i].v0)) group.push_back(j);
            size_t left = group.size();
            long cur = -1;
            while (left) {
                long next = -1;
                bool edge = false;
                if (cur >= 0)
                    for (size_t j : group) if (!used[j] && same3(in[j].e1, in[(size_t)cur].e2)) { next = (long)j; edge = true; break; }
                if (next < 0)
                    for (size_t j : group) if (!used[j]) { next = (long)j; break; }
                PmFlatTri t = in[(size_t)next];
                if (cur >= 0)
                    t.unit |= PM_FLAT_SAME_V0 | (edge ? PM_FLAT_SAME_EDGE : 0u) | (same3(t.n, in[(size_t)cur].n) ? PM_FLAT_SAME_N : 0u);
                out.push_back(t);
                used[(size_t)next] = 1; cur = next; --left;
            }
        }
        acc.flat_tris = out;
    }
    acc.flat_n[0] = (uint32_t)sph.size(); acc.flat_n[1] = (uint32_t)pat.size(); acc.flat_n[2] = (uint32_t)acc.flat_tris.size();
    acc.flat_units = sph;
    acc.flat_units.insert(acc.flat_units.end(), pat.begin(), pat.end());
}

}  // namespace pm
