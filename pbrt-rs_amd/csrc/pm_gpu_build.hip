// GPU build of the wide quantised BVH for triangle scenes and - round 3 - for bilinear-patch scenes (PM_SPLIT_HLBVH): the
// reference's only mesh primitive is the BilinearPatchMesh (shapes/bilinear_patch.rs:615-707).
//
// The reference's BVHSplitMethod::HLBVH path (primitives/aggregates.rs:273-474: Morton codes -> radix sort -> LBVH
// treelets -> SAH upper tree) is broken upstream; closest-hit results do not depend on the tree, so what
// is built here is a correct linear BVH, entirely on the device:
//   1. morton_kernel     63-bit Morton code of every primitive centroid (util/routines.rs:34-50 encode_morton_3 widened)
//   2. hipcub radix sort of (code, primitive index)          - HBM-bound: 2 x 12 B per primitive per pass
//   3. radix_tree_kernel binary radix tree over the sorted codes (Karras 2012), one thread per internal node
//   4. leafbox / rmq_level / node_box kernels   boxes of the internal nodes as range queries over the sorted leaf boxes
//   5. collapse_kernel   level by level: one thread per wide node opens up to 8 binary descendants (largest area first),
//                        assigns them to octant slots, quantises the child boxes exactly like the host builder
//                        (pm_build.cpp) and gathers the leaf records
// The result has the same layout and the same invariants as the host-built tree (tests/test_gpu_build.py walks it on the
// CPU); it is coarser than the binned-SAH tree, so PM_SPLIT_SAH stays the default for static scenes.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdlib>

#include "pm_ctx.h"

namespace {

#define GB_BLOCK 256
#define GB_LEAF_FLAG 0x80000000u

struct GBox { float lo[3], hi[3]; };
static_assert(sizeof(pm::HostTri) == 64, "HostTri is uploaded as is");
// what the build needs of a bilinear patch (pm::HostPatch carries normals and uv as well): corners, id, flags, bounds
struct GBPatch { float p[4][3]; uint32_t prim_id; uint32_t flags; pm::Box bounds; };
static_assert(sizeof(GBPatch) == 80, "GBPatch");
// per primitive type: the leaf record and how it is written
template <class P> struct GBTraits;
template <> struct GBTraits<pm::HostTri> {
    using Rec = PmTriRecord;
    static constexpr uint32_t UNITS = PM_UNITS_TRI, TYPE = PM_PRIM_TRI;
    static __device__ __forceinline__ void write(Rec& r, const pm::HostTri& tr) {
        for (int k = 0; k < 3; ++k) { r.v0[k] = tr.v[0][k]; r.v1[k] = tr.v[1][k]; r.v2[k] = tr.v[2][k]; }
        r.prim_id = tr.prim_id; r.pad[0] = 0; r.pad[1] = 0;
    }
};
template <> struct GBTraits<GBPatch> {
    using Rec = PmPatchRecord;
    static constexpr uint32_t UNITS = PM_UNITS_PATCH, TYPE = PM_PRIM_PATCH;
    static __device__ __forceinline__ void write(Rec& r, const GBPatch& pt) {
        for (int k = 0; k < 3; ++k) { r.p00[k] = pt.p[0][k]; r.p10[k] = pt.p[1][k]; r.p01[k] = pt.p[2][k]; r.p11[k] = pt.p[3][k]; }
        r.prim_id = pt.prim_id; r.flags = pt.flags; r.pad[0] = 0; r.pad[1] = 0;
    }
};

__device__ __forceinline__ unsigned long long spread21(unsigned long long x) {   // 21 bits -> every third bit
    x &= 0x1FFFFFull;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

template <class P>
__global__ void morton_kernel(const P* __restrict__ tris, uint32_t n, GBox scene, unsigned long long* __restrict__ keys,
                              uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * GB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const pm::Box& b = tris[i].bounds;
    unsigned long long code = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float ext = scene.hi[k] - scene.lo[k];
        float t = ext > 0.0f ? ((0.5f * b.lo[k] + 0.5f * b.hi[k]) - scene.lo[k]) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const unsigned long long q = (unsigned long long)fminf(t * 2097152.0f, 2097151.0f);
        code |= spread21(q) << k;
    }
    keys[i] = code;
    vals[i] = i;
}

__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = keys[i] ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

// Karras 2012: internal node i of the binary radix tree over n sorted keys; children are internal indices or
// (leaf index | GB_LEAF_FLAG); range[i] = (first, last) sorted positions the node covers
__global__ void radix_tree_kernel(const unsigned long long* __restrict__ keys, int n, uint2* __restrict__ children,
                                  uint2* __restrict__ range, uint32_t* __restrict__ parent_of_internal,
                                  uint32_t* __restrict__ parent_of_leaf) {
    const int i = blockIdx.x * GB_BLOCK + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    const uint32_t left = first == gamma ? ((uint32_t)gamma | GB_LEAF_FLAG) : (uint32_t)gamma;
    const uint32_t right = last == gamma + 1 ? ((uint32_t)(gamma + 1) | GB_LEAF_FLAG) : (uint32_t)(gamma + 1);
    children[i] = make_uint2(left, right);
    range[i] = make_uint2((uint32_t)first, (uint32_t)last);
    if (left & GB_LEAF_FLAG) parent_of_leaf[gamma] = (uint32_t)i; else parent_of_internal[gamma] = (uint32_t)i;
    if (right & GB_LEAF_FLAG) parent_of_leaf[gamma + 1] = (uint32_t)i; else parent_of_internal[gamma + 1] = (uint32_t)i;
    if (i == 0) parent_of_internal[0] = 0xFFFFFFFFu;
}

template <class P>
__device__ __forceinline__ GBox box_of(uint32_t ref, const GBox* ibox, const P* __restrict__ tris,
                                       const uint32_t* __restrict__ sorted) {
    if (ref & GB_LEAF_FLAG) {
        const pm::Box& b = tris[sorted[ref & ~GB_LEAF_FLAG]].bounds;
        GBox r;
        for (int k = 0; k < 3; ++k) { r.lo[k] = b.lo[k]; r.hi[k] = b.hi[k]; }
        return r;
    }
    return ibox[ref];
}

// ---- boxes of the internal nodes WITHOUT a bottom-up walk.  Internal node i of the radix tree covers the contiguous run
// [first, last] of the sorted leaves, so its box is a range-min/max query over the leaf boxes: a sparse table
//   T[0][j] = box of sorted leaf j,   T[k][j] = T[k-1][j] U T[k-1][min(j + 2^(k-1), n-1)]
// (log2(n) streaming passes), then one thread per internal node takes the union of two overlapping power-of-two windows.
// min / max are exact and associative, so the boxes are bit-identical to a bottom-up refit.  Round 1's refit (one thread per
// leaf climbing the tree, an atomic and two agent-scope fences per level) took 134 ms of the 256 ms build of the 21 M-triangle
// scene; this takes a few ms and K * n * 24 bytes of scratch (12.6 GB for that scene: the device has 288 GB).
template <class P>
__global__ void leafbox_kernel(const P* __restrict__ tris, const uint32_t* __restrict__ sorted, int n, GBox* __restrict__ out) {
    const int i = blockIdx.x * GB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const pm::Box& b = tris[sorted[i]].bounds;
    GBox r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = b.lo[k]; r.hi[k] = b.hi[k]; }
    out[i] = r;
}
__device__ __forceinline__ GBox box_union(const GBox& a, const GBox& b) {
    GBox r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = fminf(a.lo[k], b.lo[k]); r.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
    return r;
}
__global__ void rmq_level_kernel(const GBox* __restrict__ prev, GBox* __restrict__ cur, int n, int half) {
    const int i = blockIdx.x * GB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int j = i + half < n ? i + half : n - 1;
    cur[i] = box_union(prev[i], prev[j]);
}
__global__ void node_box_kernel(const uint2* __restrict__ range, int n_internal, const GBox* __restrict__ table, size_t n,
                                GBox* __restrict__ ibox) {
    const int i = blockIdx.x * GB_BLOCK + threadIdx.x;
    if (i >= n_internal) return;
    const uint2 r = range[i];
    const uint32_t len = r.y - r.x + 1u;
    const int k = 31 - __clz(len);
    const GBox* lvl = table + (size_t)k * n;
    ibox[i] = box_union(lvl[r.x], lvl[r.y + 1u - (1u << k)]);
}

__device__ __forceinline__ float area_of(const GBox& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}
__device__ __forceinline__ float scale_of(int e) { return __uint_as_float((uint32_t)e << 23); }

// pm_build.cpp quantise_node, on the device (same arithmetic: outward rounding checked against the exact plane in double)
__device__ void quantise_node_dev(PmWideNode& wn, const GBox& nb, const GBox* cb, const bool* used) {
    for (int k = 0; k < 3; ++k) {
        wn.o[k] = nb.lo[k];
        const float ext = nb.hi[k] - nb.lo[k];
        int e = 1;
        if (ext > 0.0f) { int ex; (void)frexpf(ext / 255.0f, &ex); e = ex + 127; }
        if (e < 1) e = 1;
        for (;; ++e) {
            const float s = scale_of(e);
            bool ok = true;
            for (int c = 0; c < 8 && ok; ++c) {
                if (!used[c]) continue;
                const float qh = ceilf((cb[c].hi[k] - wn.o[k]) / s);
                if (!(qh <= 254.0f)) ok = false;
            }
            if (ok || e >= 254) break;
        }
        wn.e[k] = (uint8_t)e;
        const float s = scale_of(e);
        for (int c = 0; c < 8; ++c) {
            if (!used[c]) { wn.qlo[k][c] = 255; wn.qhi[k][c] = 0; continue; }
            int ql = (int)floorf((cb[c].lo[k] - wn.o[k]) / s);
            int qh = (int)ceilf((cb[c].hi[k] - wn.o[k]) / s);
            ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
            qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
            while (ql > 0 && (double)ql * (double)s + (double)wn.o[k] > (double)cb[c].lo[k]) --ql;
            while (qh < 255 && (double)qh * (double)s + (double)wn.o[k] < (double)cb[c].hi[k]) ++qh;
            wn.qlo[k][c] = (uint8_t)ql;
            wn.qhi[k][c] = (uint8_t)qh;
        }
    }
}

template <class P>
struct CollapseArgs {
    const P* tris; const uint32_t* sorted; const uint2* children; const uint2* range; const GBox* ibox;
    PmWideNode* nodes; uint32_t node_cap; typename GBTraits<P>::Rec* records;
    uint32_t* node_counter; uint32_t* rec_counter; uint32_t* overflow;
    int max_leaf;
};

// one thread per wide node of this level: in[] = (wide index, binary ref); out[] receives the interior children
template <class P>
__global__ void collapse_kernel(CollapseArgs<P> a, const uint2* __restrict__ in, uint32_t n_in, uint2* __restrict__ out,
                                uint32_t* __restrict__ out_count) {
    const uint32_t t = blockIdx.x * GB_BLOCK + threadIdx.x;
    if (t >= n_in) return;
    const uint32_t wi = in[t].x, root = in[t].y;
    auto leaf_like = [&](uint32_t ref) {
        if (ref & GB_LEAF_FLAG) return true;
        const uint2 r = a.range[ref];
        return (int)(r.y - r.x + 1u) <= a.max_leaf;
    };
    uint32_t kids[8]; int nk = 0;
    if (leaf_like(root)) kids[nk++] = root;
    else { const uint2 ch = a.children[root]; kids[nk++] = ch.x; kids[nk++] = ch.y; }
    for (;;) {      // greedy expansion: open the interior child with the largest surface area
        if (nk >= 8) break;
        int best = -1; float ba = -1.0f;
        for (int i = 0; i < nk; ++i)
            if (!leaf_like(kids[i])) { const float ar = area_of(a.ibox[kids[i]]); if (ar > ba) { ba = ar; best = i; } }
        if (best < 0) break;
        const uint2 ch = a.children[kids[best]];
        kids[best] = ch.x; kids[nk++] = ch.y;
    }
    const GBox nb = box_of(root, a.ibox, a.tris, a.sorted);
    GBox kb[8];
    for (int i = 0; i < nk; ++i) kb[i] = box_of(kids[i], a.ibox, a.tris, a.sorted);
    // octant slot assignment (pm_build.cpp Collapser::fill): maximise dot(child centre - node centre, slot direction), greedily
    const float pc[3] = {0.5f * (nb.lo[0] + nb.hi[0]), 0.5f * (nb.lo[1] + nb.hi[1]), 0.5f * (nb.lo[2] + nb.hi[2])};
    float cost[8][8];
    for (int i = 0; i < nk; ++i) {
        const float d[3] = {0.5f * (kb[i].lo[0] + kb[i].hi[0]) - pc[0], 0.5f * (kb[i].lo[1] + kb[i].hi[1]) - pc[1],
                            0.5f * (kb[i].lo[2] + kb[i].hi[2]) - pc[2]};
        for (int s = 0; s < 8; ++s) cost[i][s] = ((s & 1) ? d[0] : -d[0]) + ((s & 2) ? d[1] : -d[1]) + ((s & 4) ? d[2] : -d[2]);
    }
    int in_slot[8]; bool child_done[8], slot_used[8];
    for (int s = 0; s < 8; ++s) { in_slot[s] = -1; child_done[s] = false; slot_used[s] = false; }
    for (int it = 0; it < nk; ++it) {
        int bi = -1, bs = -1; float bc = -INFINITY;
        for (int i = 0; i < nk; ++i) if (!child_done[i])
            for (int s = 0; s < 8; ++s) if (!slot_used[s] && (bi < 0 || cost[i][s] > bc)) { bc = cost[i][s]; bi = i; bs = s; }
        in_slot[bs] = bi; child_done[bi] = true; slot_used[bs] = true;
    }
    PmWideNode wn;
    memset(&wn, 0, sizeof(wn));
    GBox cb[8]; bool used[8];
    uint32_t imask = 0, n_int = 0, n_rec = 0;
    for (int s = 0; s < 8; ++s) {
        used[s] = in_slot[s] >= 0;
        if (!used[s]) continue;
        cb[s] = kb[in_slot[s]];
        const uint32_t ref = kids[in_slot[s]];
        if (!leaf_like(ref)) { imask |= 1u << s; ++n_int; }
        else n_rec += (ref & GB_LEAF_FLAG) ? 1u : (a.range[ref].y - a.range[ref].x + 1u);
    }
    quantise_node_dev(wn, nb, cb, used);
    wn.imask = (uint8_t)imask;
    const uint32_t child_base = n_int ? atomicAdd(a.node_counter, n_int) : 0u;
    const uint32_t rec_base = n_rec ? atomicAdd(a.rec_counter, n_rec) : 0u;
    if (child_base + n_int > a.node_cap) { atomicExch(a.overflow, 1u); return; }
    wn.child_base = child_base;
    wn.prim_base = (rec_base * GBTraits<P>::UNITS) | (GBTraits<P>::TYPE << 30);
    const uint32_t out_base = n_int ? atomicAdd(out_count, n_int) : 0u;
    uint32_t rank = 0, rec_off = 0;
    for (int s = 0; s < 8; ++s) {
        if (!used[s]) continue;
        const uint32_t ref = kids[in_slot[s]];
        if (imask & (1u << s)) {
            out[out_base + rank] = make_uint2(child_base + rank, ref);
            ++rank;
        } else {
            uint32_t first, count;
            if (ref & GB_LEAF_FLAG) { first = ref & ~GB_LEAF_FLAG; count = 1; }
            else { first = a.range[ref].x; count = a.range[ref].y - first + 1u; }
            wn.meta[s] = (uint8_t)((count << 5) | rec_off);
            for (uint32_t i = 0; i < count; ++i) {
                typename GBTraits<P>::Rec r;
                GBTraits<P>::write(r, a.tris[a.sorted[first + i]]);
                a.records[rec_base + rec_off + i] = r;
            }
            rec_off += count;
        }
    }
    a.nodes[wi] = wn;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

}  // namespace

#define GBCHK(call)                                                                                              \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return PM_ERR_HIP; }    \
    } while (0)

// Builds the tree of a scene of ONE primitive type (host array `prims` of n triangles or patches) on the device.  On success
// c->d_nodes / c->d_prims hold it; host copies are fetched on demand (pm_get_node_buffer / pm_get_prim_buffer).
template <class P>
static int gpu_build(pm_ctx* c, const P* prims, size_t n, int max_leaf, std::string& err) {
    using Rec = typename GBTraits<P>::Rec;
    const auto t0 = std::chrono::steady_clock::now();
    // Leaves of a linear BVH are runs of the Morton order, not SAH clusters: their boxes are loose, and every extra
    // primitive in a leaf is an extra pre-filter test for each ray that touches the box.  Measured on the C3 geometry
    // (scripts/gpu_build_probe.py, profiles/r01/gpu_build_probe.log): leaves of <= 4 / 2 / 1 primitives give
    // 26.4 / 10.9 / 4.3 tests per ray and 954 / 1237 / 1461 Mrays/s, so the device build uses single-primitive leaves
    // whatever max_prims_in_node says (PBRT_MI355_LBVH_LEAF overrides, for experiments).
    (void)max_leaf;
    max_leaf = 1;
    if (const char* e = std::getenv("PBRT_MI355_LBVH_LEAF")) { const int v = std::atoi(e); if (v >= 1 && v <= PM_MAX_LEAF) max_leaf = v; }
    GBox scene;
    for (int k = 0; k < 3; ++k) { scene.lo[k] = INFINITY; scene.hi[k] = -INFINITY; }
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) { scene.lo[k] = std::fmin(scene.lo[k], prims[i].bounds.lo[k]); scene.hi[k] = std::fmax(scene.hi[k], prims[i].bounds.hi[k]); }

    hipStream_t st = c->stream;
    DevBuf tris, keys, keys2, vals, vals2, temp, children, range, par_i, par_l, ibox, q0, q1, counters, big_nodes;
    GBCHK(tris.alloc(n * sizeof(P)));
    GBCHK(hipMemcpyAsync(tris.p, prims, n * sizeof(P), hipMemcpyHostToDevice, st));
    GBCHK(keys.alloc(n * 8)); GBCHK(keys2.alloc(n * 8)); GBCHK(vals.alloc(n * 4)); GBCHK(vals2.alloc(n * 4));
    const uint32_t grid_n = (uint32_t)((n + GB_BLOCK - 1) / GB_BLOCK);
    hipLaunchKernelGGL(morton_kernel<P>, dim3(grid_n), dim3(GB_BLOCK), 0, st, tris.as<P>(), (uint32_t)n, scene,
                       keys.as<unsigned long long>(), vals.as<uint32_t>());
    size_t temp_bytes = 0;
    GBCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys.as<unsigned long long>(), keys2.as<unsigned long long>(),
                                             vals.as<uint32_t>(), vals2.as<uint32_t>(), (int)n, 0, 63, st));
    GBCHK(temp.alloc(temp_bytes));
    GBCHK(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys.as<unsigned long long>(), keys2.as<unsigned long long>(),
                                             vals.as<uint32_t>(), vals2.as<uint32_t>(), (int)n, 0, 63, st));
    GBCHK(children.alloc((n - 1) * 8)); GBCHK(range.alloc((n - 1) * 8)); GBCHK(par_i.alloc((n - 1) * 4)); GBCHK(par_l.alloc(n * 4));
    GBCHK(ibox.alloc((n - 1) * sizeof(GBox)));
    hipLaunchKernelGGL(radix_tree_kernel, dim3(grid_n), dim3(GB_BLOCK), 0, st, keys2.as<unsigned long long>(), (int)n,
                       children.as<uint2>(), range.as<uint2>(), par_i.as<uint32_t>(), par_l.as<uint32_t>());
    {   // internal-node boxes by range queries over the sorted leaf boxes (see leafbox_kernel)
        int levels = 1;
        while ((1ull << levels) <= n) ++levels;                 // levels 0 .. floor(log2(n))
        DevBuf table;
        // levels * n * 24 bytes (12.6 GB at 21 M triangles): when the device cannot spare it - ranks sharing a card, a nearly
        // full device - the scene goes to the host builder instead of failing (PM_ERR_UNSUPPORTED = "use the host build")
        if (table.alloc((size_t)levels * n * sizeof(GBox)) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(st);
            return PM_ERR_UNSUPPORTED;
        }
        GBox* T = table.as<GBox>();
        hipLaunchKernelGGL(leafbox_kernel<P>, dim3(grid_n), dim3(GB_BLOCK), 0, st, tris.as<P>(), vals2.as<uint32_t>(), (int)n, T);
        for (int k = 1; k < levels; ++k)
            hipLaunchKernelGGL(rmq_level_kernel, dim3(grid_n), dim3(GB_BLOCK), 0, st, T + (size_t)(k - 1) * n, T + (size_t)k * n, (int)n,
                               1 << (k - 1));
        hipLaunchKernelGGL(node_box_kernel, dim3(grid_n), dim3(GB_BLOCK), 0, st, range.as<uint2>(), (int)(n - 1), T, n, ibox.as<GBox>());
        GBCHK(hipStreamSynchronize(st));                        // the table is freed when this scope ends
    }
    GBCHK(hipGetLastError());

    // collapse, level by level.  A wide node always roots at a binary node with more than max_leaf primitives, so there
    // are fewer than n of them.
    const uint32_t node_cap = (uint32_t)n;
    GBCHK(big_nodes.alloc((size_t)node_cap * sizeof(PmWideNode)));
    void* d_records = nullptr;
    GBCHK(hipMalloc(&d_records, n * sizeof(Rec)));
    DevBuf rec_guard; rec_guard.p = d_records;
    GBCHK(q0.alloc((size_t)node_cap * 8)); GBCHK(q1.alloc((size_t)node_cap * 8)); GBCHK(counters.alloc(16));
    const uint32_t init[4] = {1u /* node counter: the root is node 0 */, 0u /* records */, 0u /* overflow */, 0u /* level out count */};
    GBCHK(hipMemcpyAsync(counters.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    const uint2 root_entry = make_uint2(0u, 0u);
    GBCHK(hipMemcpyAsync(q0.p, &root_entry, sizeof(root_entry), hipMemcpyHostToDevice, st));
    CollapseArgs<P> ca{tris.as<P>(), vals2.as<uint32_t>(), children.as<uint2>(), range.as<uint2>(), ibox.as<GBox>(),
                    big_nodes.as<PmWideNode>(), node_cap, reinterpret_cast<Rec*>(d_records),
                    counters.as<uint32_t>(), counters.as<uint32_t>() + 1, counters.as<uint32_t>() + 2, max_leaf};
    uint32_t n_in = 1;
    uint2 *qin = q0.as<uint2>(), *qout = q1.as<uint2>();
    for (int level = 0; n_in > 0 && level < 4096; ++level) {
        GBCHK(hipMemsetAsync(counters.as<uint32_t>() + 3, 0, 4, st));
        hipLaunchKernelGGL(collapse_kernel<P>, dim3((n_in + GB_BLOCK - 1) / GB_BLOCK), dim3(GB_BLOCK), 0, st, ca, qin, n_in, qout,
                           counters.as<uint32_t>() + 3);
        uint32_t h[4];
        GBCHK(hipMemcpyAsync(h, counters.p, sizeof(h), hipMemcpyDeviceToHost, st));
        GBCHK(hipStreamSynchronize(st));
        if (h[2]) { err = "gpu build: node buffer overflow"; return PM_ERR_HIP; }
        n_in = h[3];
        std::swap(qin, qout);
    }
    uint32_t h[4];
    GBCHK(hipMemcpy(h, counters.p, sizeof(h), hipMemcpyDeviceToHost));
    const uint32_t n_nodes = h[0];
    if (h[1] != (uint32_t)n) { err = "gpu build: record count mismatch"; return PM_ERR_HIP; }

    // right-sized node buffer; the records buffer already is
    PmWideNode* d_nodes = nullptr;
    GBCHK(hipMalloc((void**)&d_nodes, (size_t)n_nodes * sizeof(PmWideNode)));
    hipError_t e = hipMemcpy(d_nodes, big_nodes.p, (size_t)n_nodes * sizeof(PmWideNode), hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { (void)hipFree(d_nodes); err = hipGetErrorString(e); return PM_ERR_HIP; }
    if (c->d_nodes) (void)hipFree(c->d_nodes);
    if (c->d_prims) (void)hipFree(c->d_prims);
    c->d_nodes = d_nodes;
    c->d_prims = reinterpret_cast<uint4*>(d_records);
    rec_guard.p = nullptr;
    // no host copies here (1.3 GB over PCIe for the C4 scene): pm_get_node_buffer / pm_get_prim_buffer fetch them on demand
    c->accel.nodes.clear(); c->accel.nodes.shrink_to_fit();
    c->accel.prims.clear(); c->accel.prims.shrink_to_fit();
    c->accel.dev_nodes = n_nodes;
    c->accel.dev_prim_bytes = n * sizeof(Rec);
    for (int k = 0; k < 3; ++k) { c->accel.bounds.lo[k] = scene.lo[k]; c->accel.bounds.hi[k] = scene.hi[k]; }
    c->accel.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return PM_OK;
}

// PM_SPLIT_HLBVH entry point of pm_build_accel.  Returns PM_ERR_UNSUPPORTED when the scene does not qualify - spheres, a mix of
// primitive types, fewer than 4096 primitives - or when the device cannot spare the build's scratch (the caller then uses the
// host builder).
int pm_gpu_build_tris(pm_ctx* c, int max_leaf, std::string& err) {
    const pm::HostScene& sc = c->scene;
    if (!sc.spheres.empty() || (!sc.tris.empty() && !sc.patches.empty())) return PM_ERR_UNSUPPORTED;
    if (!sc.tris.empty()) {
        const size_t n = sc.tris.size();
        if (n < 4096 || n > 0x3FFFFFFFull / PM_UNITS_TRI) return PM_ERR_UNSUPPORTED;
        return gpu_build<pm::HostTri>(c, sc.tris.data(), n, max_leaf, err);
    }
    const size_t n = sc.patches.size();
    if (n < 4096 || n > 0x3FFFFFFFull / PM_UNITS_PATCH) return PM_ERR_UNSUPPORTED;
    std::vector<GBPatch> gp(n);
    for (size_t i = 0; i < n; ++i) {
        const pm::HostPatch& p = sc.patches[i];
        std::memcpy(gp[i].p, p.p, 48);
        gp[i].prim_id = p.prim_id; gp[i].flags = p.flip_normal ? 1u : 0u; gp[i].bounds = p.bounds;
    }
    return gpu_build<GBPatch>(c, gp.data(), n, max_leaf, err);
}

// Radix sort of (key, value) pairs on `st` (the ray-reordering experiment of pm_render.hip, PBRT_MI355_REORDER): kept here because
// this file already pays for the hipcub headers.  tmp == nullptr: size query.
int pm_sort_pairs_u32(hipStream_t st, void* tmp, size_t* tmp_bytes, const uint32_t* k_in, uint32_t* k_out, const uint32_t* v_in,
                      uint32_t* v_out, uint32_t n, int end_bit) {
    return (int)hipcub::DeviceRadixSort::SortPairs(tmp, *tmp_bytes, k_in, k_out, v_in, v_out, (int)n, 0, end_bit, st);
}
