// Wavefront path tracer: Integrate::render() of the reference (integrators/base.rs:82-219,
// integrators/simple_path.rs:105-269) as a pipeline of kernels over a pass of paths whose state lives in HBM:
//
//   raygen            one path per (pixel, sample index) of the pass: PCG32 stream, wavelengths, filter, camera ray
//   closest_hit       BVHAggregate::intersect for the active queue
//   shade             interaction + BSDF; miss -> infinite lights; NEE ray -> shadow queue; BSDF sample -> next queue
//                     (queue compaction: wave ballot + LDS prefix over the block's waves, one atomic per block)
//   shadow            BVHAggregate::intersect_p for the shadow queue; unoccluded -> L += contribution
//   accumulate        per pixel, in sample-index order: PixelSensor::to_sensor_rgb + f64 film add (film.rs:160-192)
//
// One film add per (pixel, sample) in sample-index order => the f64 sums are bit-reproducible and equal to
// the CPU's one-thread-per-pixel order, with no atomics.
#include "pm_ctx.h"
#include "pm_shade.hpp"

#ifndef RB
#define RB 256          // block size of the non-traversal kernels
#endif
// Streaming (non-temporal) accesses for shade_kernel's own traffic: the path state it reads is used once and what it writes is
// read by a later kernel after gigabytes of other traffic, so keeping it out of the caches' way leaves the L2 to the scene
// tables.  Measured (scripts/ab_c4s.sh): shade_kernel -4 %.  The same hint on the other kernels' traffic (raygen's stores,
// the traversal kernels' ray loads and hit stores, the RNG state) was measured and made shade_kernel 3 % SLOWER - part of
// what one kernel has just written is still in the Infinity Cache when the next one reads it.
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t nt_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store(float4* p, float4 v) { nt_f4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p)); }
__device__ __forceinline__ void nt_store(uint2* p, uint2 v) { nt_u2 t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<nt_u2*>(p)); }
__device__ __forceinline__ float4 nt_load(const float4* p) { const nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
#ifndef PM_SHADE_WAVES_PER_EU
#define PM_SHADE_WAVES_PER_EU 5     // 96 VGPRs; measured with the dense path-state layout (scripts/ab_c4s.sh): 5 beats the
                                    // compiler's own 4 by 5 %, 6 (80 VGPRs, spills) loses 11 %.  RB = 256: 128, 512, 1024 all lose
#endif
#if PM_SHADE_WAVES_PER_EU > 0
#define PM_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(PM_SHADE_WAVES_PER_EU, PM_SHADE_WAVES_PER_EU)))
#else
#define PM_SHADE_ATTR
#endif
#define TB TRACE_BLOCK  // block size of the traversal kernels

using namespace pmd;

// Path state.  The state a bounce needs FOLLOWS THE QUEUE: entry i of the current queue lives at index i of generation g of
// the arrays below, and the shade pass writes the survivors - compacted - into generation g ^ 1 at their index in the next
// queue.  Every kernel therefore reads and writes dense, coalesced ranges whatever fraction of the paths is still alive
// (round 1 addressed twelve arrays by the path's original slot through an index queue: after compaction the gathers were
// sparse, and the same kernel moved 2.5 TB/s with every fourth slot missing against 5.3 TB/s dense, profiles/r01/bw_probe.log).
// Only what is touched once or rarely stays addressed by slot (pixel-in-8x8-tile order x sample): the radiance sum L
// (added to by unoccluded shadow rays and escaped rays), and what accumulate_kernel needs at the end.
#define PM_DEAD 0xFFFFFFFFu
#define PM_CNT_STRIDE 64
#define PM_PATH_BYTES (2 * (16 + 16 + 16 + 8) + 16 + 48 + 16 + 16 + 16 + 4)       /* device bytes per path of a pass */
struct PathArrays {
    float4* ro[2];      // xyz ray origin, w = slot (uint bits)
    float4* rd[2];      // xyz ray direction, w = depth | specular_bounce << 8 (uint bits); PM_DEAD = padded slot of a ragged window
    float4* beta[2];
    uint2* rng[2];      // PCG32 state; the increment follows from the slot's sample index (rng_inc_of)
    float4* hit;        // per entry of the current queue: closest-hit writes, shade reads
    float4 *sh_o, *sh_d, *sh_L;     // per entry of the SHADOW queue: origin + t_max, direction + slot, contribution if unoccluded
    float4 *L, *lambda_s, *pdf;     // per slot
    float* fw;                      // per slot: filter weight
    uint32_t* counts;     // 2 * (max_depth + 2) counters, PM_CNT_STRIDE words apart: active count per bounce, then shadow count
                          // per bounce.  Every block of a shade pass adds to two of them with one atomic each: on separate
                          // 256-byte lines the two streams of same-address atomics do not queue behind each other
    float4* rw_f;         // RandomWalk only: f * |cos| of bounce d of path `slot` at rw_f[d * cap + slot]
    uint32_t cap;         // entries per array
};
__device__ __forceinline__ float4 with_w(f3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }
struct SceneArrays {
    const PmWideNode* nodes; const uint4* prims;
    const uint32_t* prim_ref;        // prim id -> type << 30 | index into shade arrays; NULL in a scene without spheres: then
                                     // patches[] is indexed by the primitive id itself (shade_ref)
    const DShadePatch* patches; const DShadeSphere* spheres;     // each record names its material (.mat) and its area light (.flags >> 8)
    const DMaterial* materials; const DLight* lights; const int* infinite;
    const float* pool;
    int area_lights;                 // the scene has DiffuseAreaLights
};
__device__ __forceinline__ uint32_t shade_ref(const SceneArrays& sc, uint32_t prim_id) {
    return sc.prim_ref ? sc.prim_ref[prim_id] : (((uint32_t)PM_PRIM_PATCH << 30) | prim_id);
}

__device__ __forceinline__ unsigned long long rng_inc_of(const RenderConst& rc, uint32_t slot) {     // rng_start, dimension 0
    const unsigned long long s = (unsigned long long)(rc.s_base + (int)(slot / (uint32_t)rc.npix_pad));
    return ((s * 65536ULL) << 1) | 1ULL;
}
__device__ __forceinline__ bool slot_pixel(const RenderConst& rc, uint32_t slot, int* px, int* py, int* s) {
    const uint32_t p = slot % (uint32_t)rc.npix_pad;
    *s = rc.s_base + (int)(slot / (uint32_t)rc.npix_pad);
    const uint32_t tile = p >> 6, it = p & 63u;
    const int x = (int)(tile % (uint32_t)rc.tiles_x) * 8 + (int)(it & 7u);
    const int y = (int)(tile / (uint32_t)rc.tiles_x) * 8 + (int)(it >> 3);
    *px = rc.win_x0 + x; *py = rc.win_y0 + y;
    return x < rc.win_w && y < rc.win_h;
}

// ------------------------------------------------------------------------------------------ raygen
__global__ __launch_bounds__(RB) void raygen_kernel(RenderConst rc, SceneArrays sc, PathArrays pa, uint32_t npaths) {
    const uint32_t slot = blockIdx.x * RB + threadIdx.x;
    if (slot >= npaths) return;
    int px, py, s;
    const bool inside = slot_pixel(rc, slot, &px, &py, &s);
    if (slot == 0) pa.counts[0] = npaths;     // queue 0 is the identity: entry i holds slot i
    if (!inside) {      // padded slot of a ragged window: a dead entry, retired by the first shade pass
        pa.rd[0][slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(PM_DEAD));
        pa.L[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    Rng rng = rng_start(px, py, rc.seed, (unsigned long long)s, 0ull);
    const float lu = rng_uniform(&rng);
    float lam[4], pdf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float up = lu + (float)i / 4.0f;
        if (up > 1.0f) up -= 1.0f;
        lam[i] = sample_visible_wavelength(up);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) pdf[i] = visible_pdf(lam[i]);
    f2 uf; uf.x = rng_uniform(&rng); uf.y = rng_uniform(&rng);
    f2 fp; float fw;
    filter_sample(sc.pool, rc.filter, uf, &fp, &fw);
    f2 pf; pf.x = (float)px + fp.x + 0.5f; pf.y = (float)py + fp.y + 0.5f;
    (void)rng_uniform(&rng);    // time
    f2 pl; pl.x = rng_uniform(&rng); pl.y = rng_uniform(&rng);
    f3 o, d;
    generate_camera_ray(rc, pf, pl, &o, &d);
    pa.ro[0][slot] = with_w(o, slot);
    pa.rd[0][slot] = with_w(d, 0u | (1u << 8));                    // depth 0, specular_bounce = true
    // (beta of a camera ray is all ones: not written - shade_kernel knows the first bounce from bit 1 of its `g` argument)
    pa.rng[0][slot] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
    pa.L[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    pa.lambda_s[slot] = make_float4(lam[0], lam[1], lam[2], lam[3]);
    pa.pdf[slot] = make_float4(pdf[0], pdf[1], pdf[2], pdf[3]);
    pa.fw[slot] = fw;
}

// ------------------------------------------------------------------------------------------ traversal
template <bool COUNT, int KIND>
__global__ __launch_bounds__(TB) PM_TRAV_ATTR(false, KIND) void closest_kernel(SceneArrays sc, PathArrays pa, uint32_t g,
                                                     const uint32_t* __restrict__ count, uint32_t* __restrict__ work, uint32_t refill,
                                                     uint4* __restrict__ defer_list, uint32_t defer_cap, unsigned long long* __restrict__ counters,
                                                     const uint32_t* __restrict__ perm) {
    __shared__ uint2 stack[PM_LDS_STACK * TB];
    __shared__ uint32_t cand[PM_CAND_WORDS(false, COUNT, KIND) * TB];     // units, then t lower bounds, then the ring of parked leaves
    TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    // perm != NULL (PBRT_MI355_REORDER): the work counter hands out positions in a sorted order of the queue, perm[position] is
    // the queue entry; the hit goes back to the entry, so nothing downstream sees the order
    auto fetch = [&](uint32_t i, Ray* ray) -> bool {
        if (perm) i = perm[i];
        const float4 d = pa.rd[g][i];
        if (__float_as_uint(d.w) == PM_DEAD) return false;          // dead entry (raygen)
        const float4 o = pa.ro[g][i];
        ray->o = mk3(o.x, o.y, o.z); ray->d = mk3(d.x, d.y, d.z); ray->tmax = INFINITY;
        return true;
    };
    auto commit = [&](uint32_t i, bool found, const HitRec& h) {
        if (perm) i = perm[i];
        pa.hit[i] = make_float4(__uint_as_float(h.prim_id), h.t, h.u, h.v);
    };
    trace_persistent<false, COUNT, KIND>(sc.nodes, sc.prims, *count, work, refill, stack + threadIdx.x, cand + threadIdx.x, TB,
                                   defer_list, work + PM_WORK_SLOTS, defer_cap, &tc, fetch, commit);
    if (COUNT) {
        atomicAdd(&counters[1], (unsigned long long)tc.nodes); atomicAdd(&counters[2], (unsigned long long)tc.prims);
        atomicAdd(&counters[6], (unsigned long long)tc.exact); if (tc.fallback) atomicAdd(&counters[7], (unsigned long long)tc.fallback);
    }
    if (tc.overflow) atomicAdd(&counters[3], 1ull);
}
// PBRT_MI355_REORDER experiment: sort key of queue entry i = Morton code of the ray origin (bits per axis)
// and the direction's octant.  Entries past the queue's end and dead entries sort last.
__device__ __forceinline__ uint32_t spread3(uint32_t x) {       // 10 bits -> every third bit
    x &= 0x3FFu; x = (x | (x << 16)) & 0x030000FFu; x = (x | (x << 8)) & 0x0300F00Fu; x = (x | (x << 4)) & 0x030C30C3u; x = (x | (x << 2)) & 0x09249249u;
    return x;
}
__global__ __launch_bounds__(RB) void reorder_key_kernel(PathArrays pa, uint32_t g, uint32_t n, f3 lo, f3 scale, int bits, int oct_high,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    const float4 d = pa.rd[g][i], o = pa.ro[g][i];
    uint32_t key = 0xFFFFFFFFu >> (32 - (3 * bits + 3));
    if (__float_as_uint(d.w) != PM_DEAD) {
        const float q = (float)((1u << bits) - 1u);
        const uint32_t x = (uint32_t)fminf(fmaxf((o.x - lo.x) * scale.x, 0.0f), q), y = (uint32_t)fminf(fmaxf((o.y - lo.y) * scale.y, 0.0f), q),
                       z = (uint32_t)fminf(fmaxf((o.z - lo.z) * scale.z, 0.0f), q);
        const uint32_t m = spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
        const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
        key = oct_high ? ((oct << (3 * bits)) | m) : ((m << 3) | oct);
    }
    keys[i] = key; vals[i] = i;
}
int pm_sort_pairs_u32(hipStream_t st, void* tmp, size_t* tmp_bytes, const uint32_t* k_in, uint32_t* k_out, const uint32_t* v_in,
                      uint32_t* v_out, uint32_t n, int end_bit);      // pm_gpu_build.hip

template <bool COUNT, int KIND>
__global__ __launch_bounds__(TB) PM_TRAV_ATTR(true, KIND) void shadow_kernel(SceneArrays sc, PathArrays pa, const uint32_t* __restrict__ count,
                                                    uint32_t* __restrict__ work, uint32_t refill, uint4* __restrict__ defer_list, uint32_t defer_cap,
                                                    unsigned long long* __restrict__ counters) {
    __shared__ uint2 stack[PM_LDS_STACK * TB];
    __shared__ uint32_t cand[PM_CAND_WORDS(true, COUNT, KIND) * TB];     // units, then t lower bounds, then the ring of parked leaves
    TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    auto fetch = [&](uint32_t i, Ray* ray) -> bool {
        const float4 o = pa.sh_o[i], d = pa.sh_d[i];
        ray->o = mk3(o.x, o.y, o.z); ray->d = mk3(d.x, d.y, d.z); ray->tmax = o.w;
        return true;
    };
    auto commit = [&](uint32_t i, bool occluded, const HitRec&) {
        if (!occluded) {
            const uint32_t slot = __float_as_uint(pa.sh_d[i].w);
            float4 L = pa.L[slot];
            const float4 c = pa.sh_L[i];
            L.x += c.x; L.y += c.y; L.z += c.z; L.w += c.w;
            pa.L[slot] = L;
        }
    };
    trace_persistent<true, COUNT, KIND>(sc.nodes, sc.prims, *count, work, refill, stack + threadIdx.x, cand + threadIdx.x, TB,
                                  defer_list, work + PM_WORK_SLOTS, defer_cap, &tc, fetch, commit);
    if (COUNT) {
        atomicAdd(&counters[4], (unsigned long long)tc.nodes); atomicAdd(&counters[5], (unsigned long long)tc.prims);
        atomicAdd(&counters[6], (unsigned long long)tc.exact); if (tc.fallback) atomicAdd(&counters[7], (unsigned long long)tc.fallback);
    }
    if (tc.overflow) atomicAdd(&counters[3], 1ull);
}
// flat-sweep variants for scenes of at most PM_FLAT_MAX_PRIMS primitives (pm_device.hpp trace_flat)
template <bool COUNT, bool PATCH, bool COH = false>
__global__ __launch_bounds__(TB) PM_FLAT_ATTR void closest_flat_kernel(PmFlatScene fs, SceneArrays sc, PathArrays pa, uint32_t g,
                                                          const uint32_t* __restrict__ count, uint4* __restrict__ defer_list, uint32_t defer_cap,
                                                          uint32_t* __restrict__ defer_count, uint32_t* __restrict__ slow_list,
                                                          unsigned long long* __restrict__ counters) {
    __shared__ uint32_t cand[3 * PM_FLAT_CAND_MAX * TB];     // units, then the lower bounds of t as fractions (numerators, denominators)
    TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    auto fetch = [&](uint32_t i, Ray* ray) -> bool {
        const float4 d = pa.rd[g][i];
        if (__float_as_uint(d.w) == PM_DEAD) return false;          // dead entry (raygen)
        const float4 o = pa.ro[g][i];
        ray->o = mk3(o.x, o.y, o.z); ray->d = mk3(d.x, d.y, d.z); ray->tmax = INFINITY;
        return true;
    };
    auto commit = [&](uint32_t i, bool found, const HitRec& h) {
        pa.hit[i] = make_float4(__uint_as_float(h.prim_id), h.t, h.u, h.v);
    };
    trace_flat<false, COUNT, PATCH, COH>(fs, sc.prims, *count, cand + threadIdx.x, TB, defer_list, defer_count, slow_list,
                             defer_count + PM_WORK_SLOTS, defer_cap, &tc, fetch, commit);
    if (COUNT) {
        atomicAdd(&counters[2], (unsigned long long)tc.prims); atomicAdd(&counters[6], (unsigned long long)tc.exact);
        if (tc.fallback) atomicAdd(&counters[7], (unsigned long long)tc.fallback);
    }
    if (tc.overflow) atomicAdd(&counters[3], 1ull);
}
template <bool COUNT, bool PATCH>
__global__ __launch_bounds__(TB) PM_FLAT_ATTR void shadow_flat_kernel(PmFlatScene fs, SceneArrays sc, PathArrays pa, const uint32_t* __restrict__ count,
                                                         uint4* __restrict__ defer_list, uint32_t defer_cap, uint32_t* __restrict__ defer_count,
                                                         uint32_t* __restrict__ slow_list, unsigned long long* __restrict__ counters) {
    __shared__ uint32_t cand[3 * PM_FLAT_CAND_MAX * TB];     // units, then the lower bounds of t as fractions (numerators, denominators)
    TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    auto fetch = [&](uint32_t i, Ray* ray) -> bool {
        const float4 o = pa.sh_o[i], d = pa.sh_d[i];
        ray->o = mk3(o.x, o.y, o.z); ray->d = mk3(d.x, d.y, d.z); ray->tmax = o.w;
        return true;
    };
    auto commit = [&](uint32_t i, bool occluded, const HitRec&) {
        if (!occluded) {
            const uint32_t slot = __float_as_uint(pa.sh_d[i].w);
            float4 L = pa.L[slot];
            const float4 c = pa.sh_L[i];
            L.x += c.x; L.y += c.y; L.z += c.z; L.w += c.w;
            pa.L[slot] = L;
        }
    };
    trace_flat<true, COUNT, PATCH>(fs, sc.prims, *count, cand + threadIdx.x, TB, defer_list, defer_count, slow_list,
                            defer_count + PM_WORK_SLOTS, defer_cap, &tc, fetch, commit);
    if (COUNT) {
        atomicAdd(&counters[5], (unsigned long long)tc.prims); atomicAdd(&counters[6], (unsigned long long)tc.exact);
        if (tc.fallback) atomicAdd(&counters[7], (unsigned long long)tc.fallback);
    }
    if (tc.overflow) atomicAdd(&counters[3], 1ull);
}
// second pass (pm_device.hpp deferred_fix): exact tests of what the kernels above deferred, with the same fetch / commit
template <bool COUNT>
__global__ __launch_bounds__(TB) void closest_fix_kernel(PmFlatScene fs, SceneArrays sc, PathArrays pa, uint32_t g,
                                                         const uint4* __restrict__ defer_list, uint32_t defer_cap, const uint32_t* __restrict__ defer_count,
                                                         const uint32_t* __restrict__ slow_list, unsigned long long* __restrict__ counters) {
    TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    auto fetch = [&](uint32_t i, Ray* ray) -> bool {
        const float4 o = pa.ro[g][i], d = pa.rd[g][i];
        ray->o = mk3(o.x, o.y, o.z); ray->d = mk3(d.x, d.y, d.z); ray->tmax = INFINITY;
        return true;
    };
    auto load_t = [&](uint32_t i) -> float { return pa.hit[i].y; };
    auto commit = [&](uint32_t i, bool found, const HitRec& h) {
        pa.hit[i] = make_float4(__uint_as_float(h.prim_id), h.t, h.u, h.v);
    };
    __shared__ uint4 gather[128 * (TB / 64)];
    deferred_fix<false, COUNT>(fs, sc.prims, defer_list, defer_count, slow_list, defer_count + PM_WORK_SLOTS, defer_cap,
                               gather + 128 * (threadIdx.x >> 6), &tc, fetch, load_t, commit);
    if (COUNT && tc.exact) atomicAdd(&counters[6], (unsigned long long)tc.exact);
    if (COUNT && tc.prims) atomicAdd(&counters[2], (unsigned long long)tc.prims);
}
template <bool COUNT>
__global__ __launch_bounds__(TB) void shadow_fix_kernel(PmFlatScene fs, SceneArrays sc, PathArrays pa, const uint4* __restrict__ defer_list, uint32_t defer_cap,
                                                        const uint32_t* __restrict__ defer_count, const uint32_t* __restrict__ slow_list,
                                                        unsigned long long* __restrict__ counters) {
    TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    auto fetch = [&](uint32_t i, Ray* ray) -> bool {
        const float4 o = pa.sh_o[i], d = pa.sh_d[i];
        ray->o = mk3(o.x, o.y, o.z); ray->d = mk3(d.x, d.y, d.z); ray->tmax = o.w;
        return true;
    };
    auto load_t = [&](uint32_t) -> float { return 0.0f; };
    auto commit = [&](uint32_t i, bool occluded, const HitRec&) {
        if (!occluded) {
            const uint32_t slot = __float_as_uint(pa.sh_d[i].w);
            float4 L = pa.L[slot];
            const float4 c = pa.sh_L[i];
            L.x += c.x; L.y += c.y; L.z += c.z; L.w += c.w;
            pa.L[slot] = L;
        }
    };
    __shared__ uint4 gather[128 * (TB / 64)];
    deferred_fix<true, COUNT>(fs, sc.prims, defer_list, defer_count, slow_list, defer_count + PM_WORK_SLOTS, defer_cap,
                              gather + 128 * (threadIdx.x >> 6), &tc, fetch, load_t, commit);
    if (COUNT && tc.exact) atomicAdd(&counters[6], (unsigned long long)tc.exact);
    if (COUNT && tc.prims) atomicAdd(&counters[5], (unsigned long long)tc.prims);
}

// ------------------------------------------------------------------------------------------ shade
__device__ __forceinline__ Spec4 light_L(const SceneArrays& sc, const DLight& l, const float* lam) {
    Spec4 r;
    float t[4];
    dense_at4(sc.pool, l.table, lam, t);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = t[i] * l.scale;
    return r;
}

// two block-wide appends behind one pair of barriers: this lane's indices in queue a / queue b (or ~0u when not wanted)
__device__ __forceinline__ void block_append2(bool want_a, uint32_t* __restrict__ counter_a, bool want_b,
                                              uint32_t* __restrict__ counter_b, uint32_t* lds /* [2 * (RB/64 + 1)] */,
                                              uint32_t* idx_a, uint32_t* idx_b) {
    const unsigned long long ma = __ballot(want_a), mb = __ballot(want_b);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t* la = lds;
    uint32_t* lb = lds + RB / 64 + 1;
    if (lane == 0) { la[wave] = (uint32_t)__popcll(ma); lb[wave] = (uint32_t)__popcll(mb); }
    __syncthreads();
    if (threadIdx.x < 2) {      // lane 0 scans queue a, lane 1 queue b
        uint32_t* l = threadIdx.x ? lb : la;
        uint32_t tot = 0;
        for (uint32_t w = 0; w < RB / 64; ++w) { const uint32_t c = l[w]; l[w] = tot; tot += c; }
        l[RB / 64] = tot ? atomicAdd(threadIdx.x ? counter_b : counter_a, tot) : 0u;
    }
    __syncthreads();
    *idx_a = want_a ? la[RB / 64] + la[wave] + (uint32_t)__popcll(ma & lt) : 0xFFFFFFFFu;
    *idx_b = want_b ? lb[RB / 64] + lb[wave] + (uint32_t)__popcll(mb & lt) : 0xFFFFFFFFu;
}
// block-wide append: returns this lane's index in the output queue (or ~0u when !want)
__device__ __forceinline__ uint32_t block_append(bool want, uint32_t* __restrict__ counter, uint32_t* lds /* [RB/64 + 1] */) {
    const unsigned long long mask = __ballot(want);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) lds[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < RB / 64; ++w) { const uint32_t c = lds[w]; lds[w] = tot; tot += c; }
        lds[RB / 64] = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    const uint32_t idx = lds[RB / 64] + lds[wave] + rank;
    __syncthreads();
    return want ? idx : 0xFFFFFFFFu;
}

// SurfaceInteraction::get_bsdf (core/interaction.rs:186-215, materials/material.rs:122-195) at the sampled wavelengths;
// may terminate the secondary wavelengths (pdfw is updated in registers and in the path state)
// TEX: the scene has textured materials (compiled into the EXT shade kernel only); (u, v) = the hit's parametric coordinates
template <bool TEX>
__device__ __forceinline__ Bsdf make_bsdf(const SceneArrays& sc, const PathArrays& pa, const SurfInt& si, const DMaterial& mat,
                                          uint32_t ref, const float* lam, uint32_t slot, float u = 0.0f, float v = 0.0f) {
    Bsdf bsdf;
    bsdf.type = mat.type;
    if (mat.type == PM_MAT_DIFFUSE) {
        bool second = false;
        if (TEX && mat.texture == PM_TEX_CHECKERBOARD) {
            // st of the hit: lerped vertex uv of a patch that carries them, else the shape's own (u, v); then UvMapping::map
            f2 st; st.x = u; st.y = v;
            if ((ref >> 30) != PM_PRIM_SPHERE) {
                const DShadePatch& sp = sc.patches[ref & 0x3FFFFFFFu];
                if (sp.flags & 4u) {
                    const float* q = sc.pool + sp.uv_off;
                    st.x = lerpf(lerpf(q[0], q[4], v), lerpf(q[2], q[6], v), u);
                    st.y = lerpf(lerpf(q[1], q[5], v), lerpf(q[3], q[7], v), u);
                }
            }
            const float s = mat.su * st.x + mat.du, t = mat.sv * st.y + mat.dv;
            second = (((int)floorf(s) + (int)floorf(t)) & 1) != 0;
        }
        float refl[4];
        spectrum_at4(sc.pool, second ? mat.spec2 : mat.spec, lam, refl);
#pragma unroll
        for (int c = 0; c < 4; ++c) bsdf.refl.v[c] = clampf(refl[c], 0.0f, 1.0f);
        bsdf.eta = 1.0f; bsdf.tr.ax = 0.0f; bsdf.tr.ay = 0.0f;
    } else {
        bsdf.refl = spec_c(0.0f);
        bsdf.eta = spectrum_at(sc.pool, mat.spec, lam[0]);
        if (mat.spec.kind != PM_SPEC_CONSTANT) {   // SampledWavelengths::terminate_secondary (the only reader of pdf here)
            const float4 p4 = pa.pdf[slot];
            float pdfw[4] = {p4.x, p4.y, p4.z, p4.w};
            if (!(pdfw[1] == 0.0f && pdfw[2] == 0.0f && pdfw[3] == 0.0f)) { pdfw[0] /= 4.0f; pdfw[1] = 0.0f; pdfw[2] = 0.0f; pdfw[3] = 0.0f; }
            pa.pdf[slot] = make_float4(pdfw[0], pdfw[1], pdfw[2], pdfw[3]);
        }
        float ur = mat.ur, vr = mat.vr;
        if (mat.remap) { ur = sqrtf(ur); vr = sqrtf(vr); }
        bsdf.tr.ax = ur; bsdf.tr.ay = vr;
    }
    bsdf.fx = norm3(si.dpdus); bsdf.fz = si.ns; bsdf.fy = cross3(bsdf.fz, bsdf.fx);
    return bsdf;
}

// Light::sample_li of every light type for the receiving point pm (lights/{point,directional,infinite}.rs; DiffuseAreaLight
// over BilinearPatch::sample_with_context is an extension, EXT kernels only).  false: the sample carries nothing.
// SPH (round 3): the scene has area lights on SPHERES (Sphere::sample_with_context); ctx is then the receiving surface point
// (null for a vertex in the medium).
template <bool EXT, bool SPH = false>
__device__ __forceinline__ bool sample_light(const RenderConst& rc, const SceneArrays& sc, const DLight& lt, f3 pm, f2 u2,
                                             const float* lam, Spec4* Ls, f3* wi, float* lpdf, f3* p_light,
                                             const SurfInt* ctx = nullptr) {
    *Ls = light_L(sc, lt, lam);
    if (lt.type == PM_LIGHT_UNIFORM_INFINITE) {
        *wi = sample_uniform_sphere(u2); *lpdf = PM_INV_4PI;
        *p_light = pm + *wi * (2.0f * rc.scene_radius);
    } else if (lt.type == PM_LIGHT_DIRECTIONAL) {
        *wi = norm3(mk3(0.0f * lt.m[0] + 0.0f * lt.m[1] + 1.0f * lt.m[2], 0.0f * lt.m[4] + 0.0f * lt.m[5] + 1.0f * lt.m[6],
                        0.0f * lt.m[8] + 0.0f * lt.m[9] + 1.0f * lt.m[10]));
        *lpdf = 1.0f;
        *p_light = pm + *wi * (2.0f * rc.scene_radius);
    } else if (!EXT || lt.type == PM_LIGHT_POINT) {
        const f3 p = mk3(0.0f * lt.m[0] + 0.0f * lt.m[1] + 0.0f * lt.m[2] + lt.m[3], 0.0f * lt.m[4] + 0.0f * lt.m[5] + 0.0f * lt.m[6] + lt.m[7],
                         0.0f * lt.m[8] + 0.0f * lt.m[9] + 0.0f * lt.m[10] + lt.m[11]);
        const f3 dv = p - pm;
        *wi = norm3(dv); *lpdf = 1.0f;
        const float d2 = dot3(dv, dv);
#pragma unroll
        for (int c = 0; c < 4; ++c) Ls->v[c] /= d2;
        *p_light = p;
    } else {
        ShapeSample ss;
        const uint32_t ref = shade_ref(sc, lt.prim_id);
        if (SPH && (ref >> 30) == PM_PRIM_SPHERE) {
            const DShadeSphere& sph = sc.spheres[ref & 0x3FFFFFFFu];
            const f3 po = ctx ? offset_ray_origin(*ctx, sphere_center(sph) - pm) : pm;
            if (!sphere_sample_ctx(sph, pm, po, u2, &ss)) return false;
        } else if (!patch_sample_ctx(sc.patches[ref & 0x3FFFFFFFu], sc.pool, lt.m, lt.m[4] != 0.0f, pm, u2, &ss)) return false;
        const f3 dv = ss.p - pm;
        if (ss.pdf == 0.0f || dot3(dv, dv) == 0.0f) return false;
        *wi = norm3(dv);
        if (!lt.two_sided && dot3(ss.n, -*wi) < 0.0f) *Ls = spec_c(0.0f);
        *lpdf = ss.pdf; *p_light = ss.p;
    }
    return true;
}

// A continuation ray towards the LAST vertex of a path (depth == max_depth once it is traced) can only pick up emission there:
// the environment's or an area light's radiance, and only if (specular_bounce || !sample_lights) (simple_path.rs:124-132,
// 156-158; the loop breaks before anything else happens at that vertex).  When that is impossible the reference still traces
// the ray and discards the result; here it is not queued at all - same film, one closest-hit launch less per frame for a
// scene without such lights (C2), and only the specular paths in the last launch otherwise.
__device__ __forceinline__ bool dead_end(const RenderConst& rc, int depth_after, bool specular) {
    return depth_after == rc.max_depth && !(rc.terminal_emission && (specular || !rc.sample_lights));
}

// Regroups the vertices of a block by kind (0, 1, then 2 = none): thread t afterwards holds the words of the vertex with rank t
// in that order, and its kind.  Returns false - and changes nothing - when the block holds only one of the kinds 0 / 1.
#define PM_XCH_WORDS 27
__device__ __forceinline__ bool block_regroup(int* kind, uint32_t (&w)[PM_XCH_WORDS], uint32_t* xch /* [PM_XCH_WORDS * RB] */,
                                              uint32_t* cnt /* [2 * RB / 64] */) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long b0 = __ballot(*kind == 0), b1 = __ballot(*kind == 1);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (lane == 0) { cnt[wave] = (uint32_t)__popcll(b0); cnt[RB / 64 + wave] = (uint32_t)__popcll(b1); }
    __syncthreads();
    uint32_t n0 = 0, n1 = 0, o0 = 0, o1 = 0;
#pragma unroll
    for (uint32_t k = 0; k < RB / 64; ++k) {
        const uint32_t c0 = cnt[k], c1 = cnt[RB / 64 + k];
        if (k < wave) { o0 += c0; o1 += c1; }
        n0 += c0; n1 += c1;
    }
    if (n0 == 0u || n1 == 0u) return false;            // (block-uniform)
    if (*kind != 2) {
        const uint32_t dest = *kind == 0 ? o0 + (uint32_t)__popcll(b0 & lt) : n0 + o1 + (uint32_t)__popcll(b1 & lt);
#pragma unroll
        for (int k = 0; k < PM_XCH_WORDS; ++k) xch[k * RB + dest] = w[k];
    }
    __syncthreads();
    *kind = threadIdx.x < n0 ? 0 : (threadIdx.x < n0 + n1 ? 1 : 2);
    if (*kind != 2) {
#pragma unroll
        for (int k = 0; k < PM_XCH_WORDS; ++k) w[k] = xch[k * RB + threadIdx.x];
    }
    return true;
}

// EXT: the integrator is PM_INTEGRATOR_VOL_PATH with a medium set, or the scene holds textured materials.  That code (the medium
// event as a third outcome beside "surface" and "escaped", the regrouping of a block's vertices, texture lookups) is compiled only
// into the EXT variant: it costs registers, and the plain variant is the hot kernel of the C2 / C4 frames.
// AREA with EXT (round 3): the scene has area lights on SPHERES; only that instantiation carries Sphere::sample_with_context.
// AREA (plain variant; round 3): the scene holds diffuse area lights but neither a medium nor textures - SURVEY 8(d)'s C2, a
// Cornell box under a ceiling quad emitter.  The plain straight-line body plus emission on a hit and the shape-sampling branch of
// sample_light; round 2 sent such scenes through the EXT kernel.
template <bool EXT, bool AREA = false>
__global__ __launch_bounds__(RB) PM_SHADE_ATTR void shade_kernel(RenderConst rc, SceneArrays sc, PathArrays pa, uint32_t g_first,
                                                   const uint32_t* __restrict__ count_in,
                                                   uint32_t* __restrict__ count_out, uint32_t* __restrict__ shadow_count) {
    // g_first = generation of the path-state arrays | 2 for the first bounce of a pass (camera rays: beta = 1, never stored)
    const uint32_t g = g_first & 1u;
    const bool first = (g_first & 2u) != 0u;
    __shared__ uint32_t lds_ab[2 * (RB / 64 + 1)];
    const uint32_t i = blockIdx.x * RB + threadIdx.x;
    const uint32_t n_in = *count_in;
    if (blockIdx.x * RB >= n_in) return;             // whole block idle (uniform exit, before any barrier)
    bool cont = false, want_shadow = false;
    // what this lane hands on, written after the block-wide appends at its index in the next / the shadow queue
    uint32_t slot = 0, c_misc = 0;
    f3 c_o = mk3(0.f, 0.f, 0.f), c_d = mk3(0.f, 0.f, 0.f), s_d = mk3(0.f, 0.f, 0.f);
    float4 s_o = make_float4(0.f, 0.f, 0.f, 0.f), s_L = make_float4(0.f, 0.f, 0.f, 0.f);
    float beta[4] = {0.f, 0.f, 0.f, 0.f};
    float4 l4 = make_float4(0.f, 0.f, 0.f, 0.f);
    Rng rng; rng.state = 0ull; rng.inc = 0ull;
    // (EXT) what phase 2 of a vertex needs beyond the queue words; kind: 0 medium collision, 1 surface hit, 2 no vertex
    int kind = 2, x_depth = 0;
    uint32_t x_prim = 0xFFFFFFFFu;
    bool x_spec = false;
    float x_tmed = 0.f, x_lam[4] = {0.f, 0.f, 0.f, 0.f}, x_sig[4] = {0.f, 0.f, 0.f, 0.f};
    __shared__ uint32_t xch[EXT ? PM_XCH_WORDS * RB : 1];
    __shared__ uint32_t xch_cnt[2 * (RB / 64)];
    const bool live = i < n_in;
    // everything this vertex has at its queue index is read in ONE group (a padded entry of a ragged window reads words that
    // mean nothing and are not used): read after the PM_DEAD test, these loads were a second round trip behind the first
    float4 rd4 = make_float4(0.f, 0.f, 0.f, __uint_as_float(PM_DEAD)), ro4 = rd4, hv = rd4, b4 = rd4;
    uint2 r2 = make_uint2(0u, 0u);
    if (live) {
        rd4 = nt_load(pa.rd[g] + i); ro4 = nt_load(pa.ro[g] + i); hv = nt_load(pa.hit + i);
        b4 = first ? make_float4(1.f, 1.f, 1.f, 1.f) : nt_load(pa.beta[g] + i);
        r2 = pa.rng[g][i];
    }
    const uint32_t mw = __float_as_uint(rd4.w);
    if (live && mw != PM_DEAD) {
        slot = __float_as_uint(ro4.w);
        const uint32_t prim_id = __float_as_uint(hv.x);
        l4 = pa.lambda_s[slot];         // by slot: the wavelengths never change along a path, so they are not carried in the queue
        beta[0] = b4.x; beta[1] = b4.y; beta[2] = b4.z; beta[3] = b4.w;
        const float lam[4] = {l4.x, l4.y, l4.z, l4.w};
        int depth = (int)(mw & 0xFFu);
        bool specular = ((mw >> 8) & 1u) != 0u;
        const f3 rd = mk3(rd4.x, rd4.y, rd4.z);
        rng.state = ((unsigned long long)r2.y << 32) | r2.x; rng.inc = rng_inc_of(rc, slot);
        if constexpr (!EXT) {
            // The plain kernel (hot in C2 / C4): surface vertices only.  Kept as its own straight-line version: the unified form
            // below compiles to a slower kernel here (measured 14.2 -> 15.2 ms per C2 frame).
            constexpr bool use_medium = false;
            const float sig_t[4] = {0.f, 0.f, 0.f, 0.f};
            if (prim_id == 0xFFFFFFFFu) {
                if (!rc.sample_lights || specular) {
                    const float4 L4 = pa.L[slot];           // only escaped rays touch L here (hits: the shadow kernel adds NEE)
                    float L[4] = {L4.x, L4.y, L4.z, L4.w};
                    for (int k = 0; k < rc.n_infinite; ++k) {
                        const Spec4 le = light_L(sc, sc.lights[sc.infinite[k]], lam);
#pragma unroll
                        for (int c = 0; c < 4; ++c) L[c] += beta[c] * le.v[c];
                    }
                    pa.L[slot] = make_float4(L[0], L[1], L[2], L[3]);
                }
            } else if (AREA || depth != rc.max_depth) {
                // ---- interaction
                // (the material record is fetched beside the shading record, not behind the interaction's arithmetic)
                const uint32_t ref = shade_ref(sc, prim_id);
                SurfInt si;
                uint32_t mat_id, rec_flags = 0u;
                if ((ref >> 30) == PM_PRIM_SPHERE) {
                    const DShadeSphere& ss = sc.spheres[ref & 0x3FFFFFFFu];
                    mat_id = ss.mat;
                    si = sphere_interaction(ss, mk3(ro4.x, ro4.y, ro4.z), rd, hv.y);
                } else {
                    const DShadePatch& sp = sc.patches[ref & 0x3FFFFFFFu];
                    mat_id = sp.mat; rec_flags = sp.flags;
                    si = patch_interaction(sp, hv.z, hv.w, sc.pool);
                }
                const DMaterial mat = sc.materials[mat_id];
                // ---- emitted radiance of an area light on this primitive (simple_path.rs:156-158; AREA instantiation only)
                if (AREA && (!rc.sample_lights || specular)) {
                    const int li = (int)(rec_flags >> 8) - 1;
                    if (li >= 0) {
                        const DLight lt = sc.lights[li];
                        if (lt.two_sided || !(dot3(si.n, -rd) < 0.0f)) {
                            const Spec4 le = light_L(sc, lt, lam);
                            const float4 L4 = pa.L[slot];
                            pa.L[slot] = make_float4(L4.x + beta[0] * le.v[0], L4.y + beta[1] * le.v[1], L4.z + beta[2] * le.v[2], L4.w + beta[3] * le.v[3]);
                        }
                    }
                }
                if (depth != rc.max_depth) {
                depth++;
                Bsdf bsdf = make_bsdf<false>(sc, pa, si, mat, ref, lam, slot, hv.z, hv.w);
                const f3 wo = -rd;
                // ---- next-event estimation (simple_path.rs:182-213)
                if (rc.sample_lights) {
                    const float ul = rng_uniform(&rng);
                    if (rc.n_lights > 0) {
                        const float fi = ul * (float)rc.n_lights;
                        int idx = fi > 0.0f ? (int)fminf(fi, 2147483520.0f) : 0;
                        if (idx > rc.n_lights - 1) idx = rc.n_lights - 1;
                        const float prob = 1.0f / (float)rc.n_lights;
                        f2 u2; u2.x = rng_uniform(&rng); u2.y = rng_uniform(&rng);
                        Spec4 Ls; f3 wi, p_light; float lpdf;
                        const f3 pm = si_p(si);
                        if (sample_light<AREA>(rc, sc, sc.lights[idx], pm, u2, lam, &Ls, &wi, &lpdf, &p_light) && !spec_zero(Ls) && lpdf > 0.0f) {
                            Spec4 f = bsdf_eval(bsdf, wo, wi);
                            const float ad = fabsf(dot3(wi, si.ns));
#pragma unroll
                            for (int c = 0; c < 4; ++c) f.v[c] *= ad;
                            if (!spec_zero(f)) {
                                const f3 dir = p_light - pm;
                                const f3 so = offset_ray_origin(si, dir);
                                const float den = prob * lpdf;
                                float tr[4] = {1.f, 1.f, 1.f, 1.f};
                                float s0, s1;
                                if (EXT && use_medium && medium_segment(rc.med_bounds, so, dir, 1.0f - 0.0001f, &s0, &s1)) {
                                    const float sl = (s1 - s0) * len3(dir);
                                    cr_exp_neg4(sig_t, sl, tr);
                                }
                                s_o = make_float4(so.x, so.y, so.z, 1.0f - 0.0001f); s_d = dir;
                                if (EXT && use_medium)
                                    s_L = make_float4(beta[0] * f.v[0] * Ls.v[0] * tr[0] / den, beta[1] * f.v[1] * Ls.v[1] * tr[1] / den,
                                                      beta[2] * f.v[2] * Ls.v[2] * tr[2] / den, beta[3] * f.v[3] * Ls.v[3] * tr[3] / den);
                                else
                                    s_L = make_float4(beta[0] * f.v[0] * Ls.v[0] / den, beta[1] * f.v[1] * Ls.v[1] / den,
                                                      beta[2] * f.v[2] * Ls.v[2] / den, beta[3] * f.v[3] * Ls.v[3] / den);
                                want_shadow = true;
                            }
                        }
                    }
                }
                // ---- continue the path (simple_path.rs:215-265)
                f3 wi_next = mk3(0.f, 0.f, 0.f);
                bool ok = false;
                if (rc.sample_bsdf) {
                    const float uc = rng_uniform(&rng);
                    f2 u2; u2.x = rng_uniform(&rng); u2.y = rng_uniform(&rng);
                    BsdfSample bs;
                    if (bsdf_sample(bsdf, wo, uc, u2, &bs)) {
                        const float ad = fabsf(dot3(bs.wi, si.ns));
#pragma unroll
                        for (int c = 0; c < 4; ++c) beta[c] *= bs.value.v[c] * ad / bs.pdf;
                        specular = (bs.flags & PM_BXDF_SPECULAR) != 0;
                        wi_next = bs.wi; ok = true;
                    }
                } else {
                    const int fl = bsdf_flags(bsdf);
                    f3 wi; float spdf;
                    f2 u2; u2.x = rng_uniform(&rng); u2.y = rng_uniform(&rng);
                    if ((fl & PM_BXDF_REFLECTION) && (fl & PM_BXDF_TRANSMISSION)) { wi = sample_uniform_sphere(u2); spdf = PM_INV_4PI; }
                    else {
                        wi = sample_uniform_hemisphere(u2); spdf = PM_INV_2PI;
                        const float pr = dot3(wo, si.n) * dot3(wi, si.n);
                        if (((fl & PM_BXDF_REFLECTION) && pr < 0.0f) || ((fl & PM_BXDF_TRANSMISSION) && pr > 0.0f)) wi = -wi;
                    }
                    const Spec4 f = bsdf_eval(bsdf, wo, wi);
                    const float ad = fabsf(dot3(wi, si.ns));
#pragma unroll
                    for (int c = 0; c < 4; ++c) beta[c] *= f.v[c] * ad / spdf;
                    specular = false;
                    wi_next = wi; ok = true;
                }
                if (ok && !(beta[0] == 0.0f && beta[1] == 0.0f && beta[2] == 0.0f && beta[3] == 0.0f) && !dead_end(rc, depth, specular)) {
                    c_o = offset_ray_origin(si, wi_next); c_d = wi_next;
                    c_misc = (uint32_t)depth | (specular ? 1u << 8 : 0u);
                    cont = true;
                }
                }
            }
        } else {
            // ---- What kind of path vertex is this?  A real collision inside the homogeneous medium, a surface hit, or none
            // (escaped / absorbed / last vertex).  Phase 1, here: the medium event and the radiance of an escaped ray.  Phase 2,
            // below and outside this branch: the vertex itself.
            x_prim = prim_id; x_depth = depth; x_spec = specular;
#pragma unroll
            for (int c = 0; c < 4; ++c) x_lam[c] = lam[c];
            bool surface = true;                // false: the path ended or scattered in the medium
            if (rc.medium_on) {
                const f3 ro = mk3(ro4.x, ro4.y, ro4.z);
                float sig_a[4], sig_s[4];
                dense_at4(sc.pool, rc.med_sigma_a, lam, sig_a);
                dense_at4(sc.pool, rc.med_sigma_s, lam, sig_s);
#pragma unroll
                for (int c = 0; c < 4; ++c) x_sig[c] = sig_a[c] + sig_s[c];
                float t0, t1;
                if (medium_segment(rc.med_bounds, ro, rd, prim_id == 0xFFFFFFFFu ? INFINITY : hv.y, &t0, &t1)) {
                    const float dlen = len3(rd), seg = (t1 - t0) * dlen;
                    const float u = rng_uniform(&rng), um = rng_uniform(&rng);
                    const float s = x_sig[0] > 0.0f ? -cr_log(1.0f - u) / x_sig[0] : INFINITY;
                    if (s < seg) {
                        surface = false;
                        if (!(um < sig_a[0] / x_sig[0])) {          // scattered (absorbed otherwise: the path ends here)
                            float T[4];
                            cr_exp_neg4(x_sig, s, T);
#pragma unroll
                            for (int c = 0; c < 4; ++c) beta[c] *= T[c] * sig_s[c] / (T[0] * sig_s[0]);
                            if (depth != rc.max_depth) {
                                x_depth = depth + 1;
                                kind = 0;
                                x_tmed = t0 + s / dlen;
                            }
                        }
                    } else {
                        float T[4];                                      // passed through: T_maj over the pdf of doing so
                        cr_exp_neg4(x_sig, seg, T);
#pragma unroll
                        for (int c = 0; c < 4; ++c) beta[c] *= T[c] / T[0];
                    }
                }
            }
            if (!surface) {
            } else if (prim_id == 0xFFFFFFFFu) {
                if (!rc.sample_lights || specular) {
                    const float4 L4 = pa.L[slot];           // only escaped rays touch L here (hits: the shadow kernel adds NEE)
                    float L[4] = {L4.x, L4.y, L4.z, L4.w};
                    for (int k = 0; k < rc.n_infinite; ++k) {
                        const Spec4 le = light_L(sc, sc.lights[sc.infinite[k]], lam);
#pragma unroll
                        for (int c = 0; c < 4; ++c) L[c] += beta[c] * le.v[c];
                    }
                    pa.L[slot] = make_float4(L[0], L[1], L[2], L[3]);
                }
            } else kind = 1;
        }
    }
    if constexpr (EXT) {
        // ---- With a medium, a block's vertices are a mix of medium collisions and surface hits (C5: about 3 : 2), and their
        // code differs (phase function against interaction + BSDF): a wave holding both runs both.  The block therefore REGROUPS
        // its vertices by kind through LDS - collisions first, surface hits next, ended paths last - so that at most one of its
        // four waves is mixed.  Which thread finishes a vertex changes nothing in the result: all state travels with it, and the
        // film is accumulated by slot.
        if (rc.medium_on) {
            uint32_t w[PM_XCH_WORDS];
            w[0] = slot; w[1] = x_prim; w[2] = __float_as_uint(hv.y); w[3] = __float_as_uint(hv.z); w[4] = __float_as_uint(hv.w);
            w[5] = __float_as_uint(ro4.x); w[6] = __float_as_uint(ro4.y); w[7] = __float_as_uint(ro4.z);
            w[8] = __float_as_uint(rd4.x); w[9] = __float_as_uint(rd4.y); w[10] = __float_as_uint(rd4.z);
            w[11] = __float_as_uint(x_tmed);
            w[12] = (uint32_t)x_depth | (x_spec ? 1u << 8 : 0u);
            w[13] = (uint32_t)rng.state; w[14] = (uint32_t)(rng.state >> 32);
#pragma unroll
            for (int c = 0; c < 4; ++c) { w[15 + c] = __float_as_uint(beta[c]); w[19 + c] = __float_as_uint(x_lam[c]); w[23 + c] = __float_as_uint(x_sig[c]); }
            if (block_regroup(&kind, w, xch, xch_cnt)) {
                slot = w[0]; x_prim = w[1]; hv.y = __uint_as_float(w[2]); hv.z = __uint_as_float(w[3]); hv.w = __uint_as_float(w[4]);
                ro4.x = __uint_as_float(w[5]); ro4.y = __uint_as_float(w[6]); ro4.z = __uint_as_float(w[7]);
                rd4.x = __uint_as_float(w[8]); rd4.y = __uint_as_float(w[9]); rd4.z = __uint_as_float(w[10]);
                x_tmed = __uint_as_float(w[11]);
                x_depth = (int)(w[12] & 0xFFu); x_spec = ((w[12] >> 8) & 1u) != 0u;
                rng.state = ((unsigned long long)w[14] << 32) | w[13]; rng.inc = rng_inc_of(rc, slot);
#pragma unroll
                for (int c = 0; c < 4; ++c) { beta[c] = __uint_as_float(w[15 + c]); x_lam[c] = __uint_as_float(w[19 + c]); x_sig[c] = __uint_as_float(w[23 + c]); }
            }
        }
        if (kind != 2) {
            const bool use_medium = rc.medium_on;
            const bool vtx_medium = kind == 0;
            const uint32_t prim_id = x_prim;
            const f3 rd = mk3(rd4.x, rd4.y, rd4.z);
            const float lam[4] = {x_lam[0], x_lam[1], x_lam[2], x_lam[3]};
            const float sig_t[4] = {x_sig[0], x_sig[1], x_sig[2], x_sig[3]};
            int depth = x_depth;
            bool specular = x_spec;
            bool vtx = vtx_medium;
            f3 vp = mk3(0.f, 0.f, 0.f), wo = mk3(0.f, 0.f, 0.f);
            SurfInt si;
            Bsdf bsdf;
            if (vtx_medium) {
                vp = mk3(ro4.x, ro4.y, ro4.z) + rd * x_tmed;
                wo = (-rd) / len3(rd);
            } else {
                // ---- interaction (fetching the material record here, beside the shading record, costs this kernel registers
                // it does not have: 44.4 -> 48.5 ms per C5 frame)
                const uint32_t ref = shade_ref(sc, prim_id);
                uint32_t mat_id, rec_flags = 0u;
                if ((ref >> 30) == PM_PRIM_SPHERE) {
                    const DShadeSphere& ss = sc.spheres[ref & 0x3FFFFFFFu];
                    mat_id = ss.mat; rec_flags = ss.flags;
                    si = sphere_interaction(ss, mk3(ro4.x, ro4.y, ro4.z), rd, hv.y);
                } else {
                    const DShadePatch& sp = sc.patches[ref & 0x3FFFFFFFu];
                    mat_id = sp.mat; rec_flags = sp.flags;
                    si = patch_interaction(sp, hv.z, hv.w, sc.pool);
                }
                // ---- emitted radiance of an area light on this primitive (simple_path.rs:156-158)
                if (sc.area_lights && (!rc.sample_lights || specular)) {
                    const int li = (int)(rec_flags >> 8) - 1;
                    if (li >= 0) {
                        const DLight lt = sc.lights[li];
                        if (lt.two_sided || !(dot3(si.n, -rd) < 0.0f)) {
                            const Spec4 le = light_L(sc, lt, lam);
                            const float4 L4 = pa.L[slot];
                            pa.L[slot] = make_float4(L4.x + beta[0] * le.v[0], L4.y + beta[1] * le.v[1], L4.z + beta[2] * le.v[2], L4.w + beta[3] * le.v[3]);
                        }
                    }
                }
                if (depth != rc.max_depth) {
                    depth++;
                    bsdf = make_bsdf<EXT>(sc, pa, si, sc.materials[mat_id], ref, lam, slot, hv.z, hv.w);
                    vtx = true;
                    vp = si_p(si);
                    wo = -rd;
                }
            }
            if (vtx) {
                // ---- next-event estimation (simple_path.rs:182-213; the medium vertex evaluates the phase function instead)
                if (rc.sample_lights) {
                    const float ul = rng_uniform(&rng);
                    if (rc.n_lights > 0) {
                        const float fi = ul * (float)rc.n_lights;
                        int idx = fi > 0.0f ? (int)fminf(fi, 2147483520.0f) : 0;
                        if (idx > rc.n_lights - 1) idx = rc.n_lights - 1;
                        const float prob = 1.0f / (float)rc.n_lights;
                        f2 u2; u2.x = rng_uniform(&rng); u2.y = rng_uniform(&rng);
                        Spec4 Ls; f3 wi, p_light; float lpdf;
                        if (sample_light<EXT, AREA>(rc, sc, sc.lights[idx], vp, u2, lam, &Ls, &wi, &lpdf, &p_light, vtx_medium ? nullptr : &si) && !spec_zero(Ls) && lpdf > 0.0f) {
                            Spec4 f;
                            if (EXT && vtx_medium) f = spec_c(henyey_greenstein(dot3(wo, wi), rc.med_g));
                            else {
                                f = bsdf_eval(bsdf, wo, wi);
                                const float ad = fabsf(dot3(wi, si.ns));
#pragma unroll
                                for (int c = 0; c < 4; ++c) f.v[c] *= ad;
                            }
                            if (!spec_zero(f)) {
                                const f3 dir = p_light - vp;
                                const f3 so = (EXT && vtx_medium) ? vp : offset_ray_origin(si, dir);
                                const float den = prob * lpdf;
                                float tr[4] = {1.f, 1.f, 1.f, 1.f};
                                float s0, s1;
                                if (EXT && use_medium && medium_segment(rc.med_bounds, so, dir, 1.0f - 0.0001f, &s0, &s1)) {
                                    const float sl = (s1 - s0) * len3(dir);
                                    cr_exp_neg4(sig_t, sl, tr);
                                }
                                s_o = make_float4(so.x, so.y, so.z, 1.0f - 0.0001f); s_d = dir;
                                if (EXT && use_medium)
                                    s_L = make_float4(beta[0] * f.v[0] * Ls.v[0] * tr[0] / den, beta[1] * f.v[1] * Ls.v[1] * tr[1] / den,
                                                      beta[2] * f.v[2] * Ls.v[2] * tr[2] / den, beta[3] * f.v[3] * Ls.v[3] * tr[3] / den);
                                else
                                    s_L = make_float4(beta[0] * f.v[0] * Ls.v[0] / den, beta[1] * f.v[1] * Ls.v[1] / den,
                                                      beta[2] * f.v[2] * Ls.v[2] / den, beta[3] * f.v[3] * Ls.v[3] / den);
                                want_shadow = true;
                            }
                        }
                    }
                }
                // ---- continue the path (simple_path.rs:215-265; the medium vertex samples the phase function)
                f3 wi_next = mk3(0.f, 0.f, 0.f);
                bool ok = false;
                if (EXT && vtx_medium) {
                    f2 u2; u2.x = rng_uniform(&rng); u2.y = rng_uniform(&rng);
                    float ppdf;
                    wi_next = sample_henyey_greenstein(wo, rc.med_g, u2, &ppdf);
                    ok = ppdf != 0.0f;
                    specular = false;
                } else if (rc.sample_bsdf) {
                    const float uc = rng_uniform(&rng);
                    f2 u2; u2.x = rng_uniform(&rng); u2.y = rng_uniform(&rng);
                    BsdfSample bs;
                    if (bsdf_sample(bsdf, wo, uc, u2, &bs)) {
                        const float ad = fabsf(dot3(bs.wi, si.ns));
#pragma unroll
                        for (int c = 0; c < 4; ++c) beta[c] *= bs.value.v[c] * ad / bs.pdf;
                        specular = (bs.flags & PM_BXDF_SPECULAR) != 0;
                        wi_next = bs.wi; ok = true;
                    }
                } else {
                    const int fl = bsdf_flags(bsdf);
                    f3 wi; float spdf;
                    f2 u2; u2.x = rng_uniform(&rng); u2.y = rng_uniform(&rng);
                    if ((fl & PM_BXDF_REFLECTION) && (fl & PM_BXDF_TRANSMISSION)) { wi = sample_uniform_sphere(u2); spdf = PM_INV_4PI; }
                    else {
                        wi = sample_uniform_hemisphere(u2); spdf = PM_INV_2PI;
                        const float pr = dot3(wo, si.n) * dot3(wi, si.n);
                        if (((fl & PM_BXDF_REFLECTION) && pr < 0.0f) || ((fl & PM_BXDF_TRANSMISSION) && pr > 0.0f)) wi = -wi;
                    }
                    const Spec4 f = bsdf_eval(bsdf, wo, wi);
                    const float ad = fabsf(dot3(wi, si.ns));
#pragma unroll
                    for (int c = 0; c < 4; ++c) beta[c] *= f.v[c] * ad / spdf;
                    specular = false;
                    wi_next = wi; ok = true;
                }
                if (ok && !(beta[0] == 0.0f && beta[1] == 0.0f && beta[2] == 0.0f && beta[3] == 0.0f) && !dead_end(rc, depth, specular)) {
                    c_o = (EXT && vtx_medium) ? vp : offset_ray_origin(si, wi_next); c_d = wi_next;
                    c_misc = (uint32_t)depth | (specular ? 1u << 8 : 0u);
                    cont = true;
                }
            }
        }
    }
    uint32_t oi, si_;
    block_append2(cont, count_out, want_shadow, shadow_count, lds_ab, &oi, &si_);
    if (cont) {         // the survivor's state, compacted: entry oi of the next queue, generation g ^ 1
        const uint32_t h = g ^ 1u;
        nt_store(pa.ro[h] + oi, with_w(c_o, slot));
        nt_store(pa.rd[h] + oi, with_w(c_d, c_misc));
        nt_store(pa.beta[h] + oi, make_float4(beta[0], beta[1], beta[2], beta[3]));
        nt_store(pa.rng[h] + oi, make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32)));
    }
    if (want_shadow) {
        nt_store(pa.sh_o + si_, s_o);
        nt_store(pa.sh_d + si_, with_w(s_d, slot));
        nt_store(pa.sh_L + si_, s_L);
    }
}

// ------------------------------------------------------------------------------------------ shade (RandomWalk)
// RandomWalkIntegrator::incident_radiance_random_walk (integrators/random_walk.rs:106-173), recursion unrolled: bounce d
// stores f * |cos| in rw_f[d]; when the walk ends with radiance Le, the result is folded from the deepest bounce outwards
// exactly as the recursion returns it:  L = 0 + f_d * L / (1 / (4 pi)).
__global__ __launch_bounds__(RB) void shade_rw_kernel(RenderConst rc, SceneArrays sc, PathArrays pa, uint32_t g,
                                                      const uint32_t* __restrict__ count_in, uint32_t* __restrict__ count_out) {
    __shared__ uint32_t lds_a[RB / 64 + 1];
    const uint32_t i = blockIdx.x * RB + threadIdx.x;
    const uint32_t n_in = *count_in;
    if (blockIdx.x * RB >= n_in) return;             // whole block idle (uniform exit, before any barrier)
    bool cont = false;
    uint32_t slot = 0, c_misc = 0;
    f3 c_o = mk3(0.f, 0.f, 0.f), c_d = mk3(0.f, 0.f, 0.f);
    float4 l4 = make_float4(0.f, 0.f, 0.f, 0.f);
    Rng rng; rng.state = 0ull; rng.inc = 0ull;
    const bool live = i < n_in;
    const float4 rd4 = live ? pa.rd[g][i] : make_float4(0.f, 0.f, 0.f, __uint_as_float(PM_DEAD));
    const uint32_t mw = __float_as_uint(rd4.w);
    if (live && mw != PM_DEAD) {
        const float4 ro4 = pa.ro[g][i];
        slot = __float_as_uint(ro4.w);
        const float4 hv = pa.hit[i];
        const uint32_t prim_id = __float_as_uint(hv.x);
        l4 = pa.lambda_s[slot];
        const float lam[4] = {l4.x, l4.y, l4.z, l4.w};
        const int depth = (int)(mw & 0xFFu);
        float Lr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        bool done = true;
        if (prim_id == 0xFFFFFFFFu) {
            for (int k = 0; k < rc.n_infinite; ++k) {       // Sum::sum from the zero spectrum
                const Spec4 le = light_L(sc, sc.lights[sc.infinite[k]], lam);
#pragma unroll
                for (int c = 0; c < 4; ++c) Lr[c] += le.v[c];
            }
        } else if (depth != rc.max_depth) {
            const f3 ro = mk3(ro4.x, ro4.y, ro4.z), rd = mk3(rd4.x, rd4.y, rd4.z);
            const uint32_t ref = shade_ref(sc, prim_id);
            SurfInt si;
            uint32_t mat_id;
            if ((ref >> 30) == PM_PRIM_SPHERE) { mat_id = sc.spheres[ref & 0x3FFFFFFFu].mat; si = sphere_interaction(sc.spheres[ref & 0x3FFFFFFFu], ro, rd, hv.y); }
            else { mat_id = sc.patches[ref & 0x3FFFFFFFu].mat; si = patch_interaction(sc.patches[ref & 0x3FFFFFFFu], hv.z, hv.w, sc.pool); }
            const Bsdf bsdf = make_bsdf<false>(sc, pa, si, sc.materials[mat_id], ref, lam, slot);
            const f3 wo = -rd;
            const uint2 r2 = pa.rng[g][i];
            rng.state = ((unsigned long long)r2.y << 32) | r2.x; rng.inc = rng_inc_of(rc, slot);
            f2 u2; u2.x = rng_uniform(&rng); u2.y = rng_uniform(&rng);
            const f3 wp = sample_uniform_sphere(u2);
            Spec4 f = bsdf_eval(bsdf, wo, wp);
            if (!spec_zero(f)) {
                const float ad = fabsf(dot3(wp, si.ns));
#pragma unroll
                for (int c = 0; c < 4; ++c) f.v[c] *= ad;
                pa.rw_f[(size_t)depth * pa.cap + slot] = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
                c_o = offset_ray_origin(si, wp); c_d = wp; c_misc = (uint32_t)(depth + 1);
                cont = true; done = false;
            }
        }
        if (done) {
            const float pdf = 1.0f / (4.0f * 3.14159265358979323846f);
            for (int k = depth - 1; k >= 0; --k) {
                const float4 f = pa.rw_f[(size_t)k * pa.cap + slot];
                Lr[0] = 0.0f + f.x * Lr[0] / pdf; Lr[1] = 0.0f + f.y * Lr[1] / pdf;
                Lr[2] = 0.0f + f.z * Lr[2] / pdf; Lr[3] = 0.0f + f.w * Lr[3] / pdf;
            }
            pa.L[slot] = make_float4(Lr[0], Lr[1], Lr[2], Lr[3]);
        }
    }
    const uint32_t oi = block_append(cont, count_out, lds_a);
    if (cont) {
        const uint32_t h = g ^ 1u;
        pa.ro[h][oi] = with_w(c_o, slot);
        pa.rd[h][oi] = with_w(c_d, c_misc);
        pa.rng[h][oi] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
    }
}

// ------------------------------------------------------------------------------------------ accumulate
__global__ __launch_bounds__(RB) void accumulate_kernel(RenderConst rc, SceneArrays sc, PathArrays pa, double* __restrict__ film) {
    const uint32_t p = blockIdx.x * RB + threadIdx.x;
    if (p >= (uint32_t)rc.npix_pad) return;
    int px, py, s0;
    if (!slot_pixel(rc, p, &px, &py, &s0)) return;
    double* fp = film + ((size_t)(py - rc.film_y0) * rc.film_w + (px - rc.film_x0)) * 4;
    double acc[4] = {fp[0], fp[1], fp[2], fp[3]};
    for (int s = 0; s < rc.s_count; ++s) {
        const uint32_t slot = (uint32_t)s * (uint32_t)rc.npix_pad + p;
        const float4 L4 = pa.L[slot], l4 = pa.lambda_s[slot], p4 = pa.pdf[slot];
        const float L[4] = {L4.x, L4.y, L4.z, L4.w}, lam[4] = {l4.x, l4.y, l4.z, l4.w}, pd[4] = {p4.x, p4.y, p4.z, p4.w};
        const float w = pa.fw[slot];
        float Lp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) Lp[i] = pd[i] != 0.0f ? (L[i] * 1.0f) / pd[i] : 0.0f;
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sum = 0.0f, sv[4];
            dense_at4(sc.pool, rc.sensor_table + 471 * c, lam, sv);
#pragma unroll
            for (int i = 0; i < 4; ++i) sum += sv[i] * Lp[i];
            rgb[c] = sum / 4.0f * rc.imaging_ratio;
        }
        float m = rgb[0]; if (!(rgb[1] < m)) m = rgb[1]; if (!(rgb[2] < m)) m = rgb[2];
        if (m > rc.max_component_value) { const float sc_ = rc.max_component_value / m; rgb[0] *= sc_; rgb[1] *= sc_; rgb[2] *= sc_; }
        const double wd = (double)w;
        acc[0] += wd * (double)rgb[0]; acc[1] += wd * (double)rgb[1]; acc[2] += wd * (double)rgb[2]; acc[3] += wd;
    }
    fp[0] = acc[0]; fp[1] = acc[1]; fp[2] = acc[2]; fp[3] = acc[3];
}

// PBRT_MI355_VERIFY=1 (debugging aid): after every timed closest-hit launch the COUNTING kernel - whose walk does not depend on
// the wave and equals the CPU mirror's - traces the same queue again, and this kernel compares the two hit arrays
struct VerifyRec { float o[4], d[4]; float4 timed, counting; uint32_t index, bounce, pad[2]; };
__global__ void verify_hits_kernel(PathArrays pa, uint32_t g, const uint32_t* __restrict__ count, const float4* __restrict__ timed,
                                   uint32_t bounce, uint32_t* __restrict__ n_bad, VerifyRec* __restrict__ recs, uint32_t max_recs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *count) return;
    const float4 d = pa.rd[g][i];
    if (__float_as_uint(d.w) == PM_DEAD) return;
    const float4 a = timed[i], b = pa.hit[i];
    if (__float_as_uint(a.x) != __float_as_uint(b.x) || __float_as_uint(a.y) != __float_as_uint(b.y) ||
        __float_as_uint(a.z) != __float_as_uint(b.z) || __float_as_uint(a.w) != __float_as_uint(b.w)) {
        const uint32_t k = atomicAdd(n_bad, 1u);
        if (k < max_recs) {
            const float4 o = pa.ro[g][i];
            VerifyRec r;
            r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z; r.o[3] = o.w; r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z; r.d[3] = d.w;
            r.timed = a; r.counting = b; r.index = i; r.bounce = bounce; r.pad[0] = 0; r.pad[1] = 0;
            recs[k] = r;
        }
    }
}

__global__ void film_add_kernel(double* __restrict__ film, const double* __restrict__ other, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) film[i] += other[i];
}
__global__ void film_image_kernel(const double* __restrict__ film, float* __restrict__ rgb, size_t npix, const float* __restrict__ m9) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const double* p = film + i * 4;
    float c[3] = {(float)p[0], (float)p[1], (float)p[2]};
    if (p[3] != 0.0) { const float ws = (float)p[3]; c[0] /= ws; c[1] /= ws; c[2] /= ws; }
    for (int r = 0; r < 3; ++r) {
        float acc = 0.0f;
        for (int k = 0; k < 3; ++k) acc += m9[r * 3 + k] * c[k];
        rgb[i * 3 + r] = acc;
    }
}

// ========================================================================================== host side
struct RenderState {
    // host copies of what the caller set
    std::vector<PmMaterial> mats; std::vector<std::vector<float>> mat_tables, mat_tables2; bool have_textures = false;
    std::vector<uint32_t> prim_mat;
    std::vector<PmLight> lights; std::vector<std::vector<float>> light_tables;
    bool have_medium = false; PmMedium medium{}; std::vector<float> med_sigma_a, med_sigma_s;
    int med_off_a = 0, med_off_s = 0; bool have_area_lights = false; bool have_sphere_lights = false; bool direct_records = false;
    PmCamera cam{}; bool have_cam = false;
    PmFilm film{}; bool have_film = false;
    std::vector<float> sensor, filter_vals;
    // device
    bool scene_dirty = true; uint64_t geom_gen = ~0ull;     // pm_ctx::geom_gen the tables were uploaded for
    float* d_pool = nullptr; DMaterial* d_mats = nullptr; DLight* d_lights = nullptr; int* d_inf = nullptr;
    uint32_t* d_prim_ref = nullptr; DShadePatch* d_patches = nullptr; DShadeSphere* d_spheres = nullptr;
    float* d_m9 = nullptr;
    int n_inf = 0;
    DFilter dfilter{}; int sensor_off = 0;
    double* d_film = nullptr; size_t film_pixels = 0;
    // path arrays: one set per LANE.  A frame of several passes is rendered on two lanes - two streams, each with its own path
    // state, work counters and deferred lists - whose passes alternate (render_impl): the sweep / walk kernels of one lane are
    // bound by instruction issue, the shade kernel of the other waits on memory, and together they fill the chip better than
    // either alone.  Lane 0 uses the context's stream and lists, lane 1 what is kept here.
    struct PathBuf { PathArrays pa{}; size_t cap = 0; int depth = 0; std::vector<void*> allocs; };
    PathBuf pb[2];
    hipStream_t stream1 = nullptr;
    uint32_t* d_work1 = nullptr; uint4* d_defer1 = nullptr; uint32_t* d_slow1 = nullptr; size_t defer_cap1 = 0;
    hipEvent_t ev_go = nullptr, ev_acc[2] = {nullptr, nullptr}, ev_join = nullptr;
    uint32_t* h_counts[2] = {nullptr, nullptr}; size_t h_counts_words = 0;     // pinned: queue lengths of a lane's pass
    PmRenderStats stats{};
    std::vector<uint64_t> q_closest, q_shadow;      // queue entries per bounce of the last render (all passes)
    std::vector<hipEvent_t> ev_pool[2];   // per lane, per-launch timing: one event before every kernel of a pass + one after
    // PBRT_MI355_REORDER: keys / values in and out of the radix sort, its scratch
    uint32_t* d_sort[4] = {nullptr, nullptr, nullptr, nullptr}; void* d_sort_tmp = nullptr; size_t sort_cap = 0, sort_tmp_bytes = 0;
};

void pm_render_state_destroy(pm_ctx* c) {
    RenderState* r = c->render;
    if (!r) return;
    for (auto& b : r->pb) for (void* p : b.allocs) (void)hipFree(p);
    for (auto& pool : r->ev_pool) for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : {r->ev_go, r->ev_acc[0], r->ev_acc[1], r->ev_join}) if (e) (void)hipEventDestroy(e);
    for (uint32_t* h : r->h_counts) if (h) (void)hipHostFree(h);
    if (r->d_work1) (void)hipFree(r->d_work1);
    if (r->d_defer1) (void)hipFree(r->d_defer1);
    if (r->d_slow1) (void)hipFree(r->d_slow1);
    if (r->stream1) (void)hipStreamDestroy(r->stream1);
    void* ptrs[] = {r->d_pool, r->d_mats, r->d_lights, r->d_inf, r->d_prim_ref, r->d_patches, r->d_spheres, r->d_m9, r->d_film};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (uint32_t* p : r->d_sort) if (p) (void)hipFree(p);
    if (r->d_sort_tmp) (void)hipFree(r->d_sort_tmp);
    delete r;
    c->render = nullptr;
}
static RenderState* rs(pm_ctx* c) { if (!c->render) c->render = new RenderState(); return c->render; }

static bool spec_needs_table(int kind) { return kind == PM_SPEC_DENSE || kind == PM_SPEC_RGB_ILLUMINANT; }

extern "C" int pm_set_materials(pm_ctx* c, const PmMaterial* m, uint32_t n, uint32_t struct_bytes) {
    if (!c || (!m && n)) return fail(c, PM_ERR_INVALID, "pm_set_materials: null argument");
    PM_ABI_CHECK(c, PmMaterial, struct_bytes, "pm_set_materials");
    RenderState* r = rs(c);
    for (uint32_t i = 0; i < n; ++i) {      // validate everything first: a rejected call leaves the previous materials in place
        if (m[i].type != PM_MAT_DIFFUSE && m[i].type != PM_MAT_DIELECTRIC) return fail(c, PM_ERR_INVALID, "pm_set_materials: unknown material type");
        if (m[i].spectrum.kind < PM_SPEC_CONSTANT || m[i].spectrum.kind > PM_SPEC_BLACKBODY) return fail(c, PM_ERR_INVALID, "pm_set_materials: unknown spectrum kind");
        if (spec_needs_table(m[i].spectrum.kind) && !m[i].spectrum.table) return fail(c, PM_ERR_INVALID, "pm_set_materials: spectrum needs a 471-entry table");
        if (m[i].texture != PM_TEX_NONE && m[i].texture != PM_TEX_CHECKERBOARD) return fail(c, PM_ERR_INVALID, "pm_set_materials: unknown texture kind");
        if (m[i].texture != PM_TEX_NONE) {
            if (m[i].type != PM_MAT_DIFFUSE) return fail(c, PM_ERR_UNSUPPORTED, "pm_set_materials: textures are supported on diffuse materials only");
            if (m[i].spectrum2.kind < PM_SPEC_CONSTANT || m[i].spectrum2.kind > PM_SPEC_BLACKBODY) return fail(c, PM_ERR_INVALID, "pm_set_materials: unknown spectrum kind (spectrum2)");
            if (spec_needs_table(m[i].spectrum2.kind) && !m[i].spectrum2.table) return fail(c, PM_ERR_INVALID, "pm_set_materials: spectrum2 needs a 471-entry table");
        }
    }
    r->mats.assign(m, m + n); r->mat_tables.assign(n, {}); r->mat_tables2.assign(n, {});
    for (uint32_t i = 0; i < n; ++i) {
        if (spec_needs_table(m[i].spectrum.kind)) r->mat_tables[i].assign(m[i].spectrum.table, m[i].spectrum.table + 471);
        r->mats[i].spectrum.table = nullptr;
        if (m[i].texture != PM_TEX_NONE && spec_needs_table(m[i].spectrum2.kind))
            r->mat_tables2[i].assign(m[i].spectrum2.table, m[i].spectrum2.table + 471);
        r->mats[i].spectrum2.table = nullptr;
    }
    r->scene_dirty = true;
    return PM_OK;
}
extern "C" int pm_set_prim_materials(pm_ctx* c, const uint32_t* ids, uint32_t n) {
    if (!c || (!ids && n)) return fail(c, PM_ERR_INVALID, "pm_set_prim_materials: null argument");
    RenderState* r = rs(c);
    r->prim_mat.assign(ids, ids + n);
    r->scene_dirty = true;
    return PM_OK;
}
extern "C" int pm_set_lights(pm_ctx* c, const PmLight* l, uint32_t n, uint32_t struct_bytes) {
    if (!c || (!l && n)) return fail(c, PM_ERR_INVALID, "pm_set_lights: null argument");
    PM_ABI_CHECK(c, PmLight, struct_bytes, "pm_set_lights");
    RenderState* r = rs(c);
    for (uint32_t i = 0; i < n; ++i) {
        if (l[i].type < PM_LIGHT_POINT || l[i].type > PM_LIGHT_DIFFUSE_AREA) return fail(c, PM_ERR_INVALID, "pm_set_lights: unknown light type");
        if (!l[i].spectrum) return fail(c, PM_ERR_INVALID, "pm_set_lights: spectrum table missing");
    }
    r->lights.assign(l, l + n); r->light_tables.assign(n, {});
    for (uint32_t i = 0; i < n; ++i) { r->light_tables[i].assign(l[i].spectrum, l[i].spectrum + 471); r->lights[i].spectrum = nullptr; }
    r->scene_dirty = true;
    return PM_OK;
}
extern "C" int pm_set_medium(pm_ctx* c, const PmMedium* m, uint32_t struct_bytes) {
    if (!c) return PM_ERR_INVALID;
    RenderState* r = rs(c);
    if (!m) { r->have_medium = false; r->scene_dirty = true; return PM_OK; }
    PM_ABI_CHECK(c, PmMedium, struct_bytes, "pm_set_medium");
    // everything is validated BEFORE the state is touched: a rejected medium leaves the previous one (or none) in place
    if (!m->sigma_a || !m->sigma_s) return fail(c, PM_ERR_INVALID, "pm_set_medium: sigma_a and sigma_s tables (471 floats each) are required");
    for (int k = 0; k < 3; ++k) if (!(m->bounds[k] < m->bounds[3 + k])) return fail(c, PM_ERR_INVALID, "pm_set_medium: empty bounds");
    if (!(std::fabs(m->g) < 1.0f)) return fail(c, PM_ERR_INVALID, "pm_set_medium: |g| must be < 1");
    for (int k = 0; k < 471; ++k)
        if (!(m->sigma_a[k] >= 0.0f) || !(m->sigma_s[k] >= 0.0f) || std::isinf(m->sigma_a[k]) || std::isinf(m->sigma_s[k]))
            return fail(c, PM_ERR_INVALID, "pm_set_medium: negative, infinite or NaN coefficient");
    r->medium = *m; r->have_medium = true;
    r->med_sigma_a.assign(m->sigma_a, m->sigma_a + 471); r->med_sigma_s.assign(m->sigma_s, m->sigma_s + 471);
    r->medium.sigma_a = nullptr; r->medium.sigma_s = nullptr;
    r->scene_dirty = true;
    return PM_OK;
}
extern "C" int pm_debug_medium_state(pm_ctx* c, float* out4) {
    if (!c || !out4) return fail(c, PM_ERR_INVALID, "pm_debug_medium_state: null argument");
    RenderState* r = rs(c);
    out4[0] = r->have_medium ? 1.0f : 0.0f; out4[1] = r->have_medium ? r->medium.g : 0.0f;
    out4[2] = r->have_medium ? r->med_sigma_a[0] : 0.0f; out4[3] = r->have_medium ? r->med_sigma_s[0] : 0.0f;
    return PM_OK;
}
extern "C" int pm_set_camera(pm_ctx* c, const PmCamera* cam, uint32_t struct_bytes) {
    if (!c || !cam) return fail(c, PM_ERR_INVALID, "pm_set_camera: null argument");
    PM_ABI_CHECK(c, PmCamera, struct_bytes, "pm_set_camera");
    const float* m = cam->render_from_camera;
    if (!(m[12] == 0.f && m[13] == 0.f && m[14] == 0.f && m[15] == 1.f)) return fail(c, PM_ERR_UNSUPPORTED, "pm_set_camera: render_from_camera must be affine");
    RenderState* r = rs(c);
    r->cam = *cam; r->have_cam = true;
    return PM_OK;
}

// PiecewiseConstant1D::new (sampling/routines.rs:351-387)
static float pc1d_build(const float* f, int n, float mn, float mx, std::vector<float>& func, std::vector<float>& cdf) {
    func.resize(n); cdf.resize(n + 1);
    for (int i = 0; i < n; ++i) func[i] = std::fabs(f[i]);
    cdf[0] = 0.0f;
    for (int i = 1; i <= n; ++i) cdf[i] = cdf[i - 1] + func[i - 1] * (mx - mn) / (float)n;
    const float integral = cdf[n];
    if (integral == 0.0f) { for (int i = 1; i <= n; ++i) cdf[i] = (float)i / (float)n; }
    else { for (int i = 0; i <= n; ++i) cdf[i] /= integral; }
    return integral;
}

extern "C" int pm_set_film(pm_ctx* c, const PmFilm* f, uint32_t struct_bytes) {
    if (!c || !f) return fail(c, PM_ERR_INVALID, "pm_set_film: null argument");
    PM_ABI_CHECK(c, PmFilm, struct_bytes, "pm_set_film");
    if (c->device < 0) return fail(c, PM_ERR_INVALID, "pm_set_film: host-only context");
    const int w = f->pixel_bounds[2] - f->pixel_bounds[0], h = f->pixel_bounds[3] - f->pixel_bounds[1];
    if (w <= 0 || h <= 0) return fail(c, PM_ERR_INVALID, "pm_set_film: empty pixel_bounds");
    if (!f->sensor_rgb_bar) return fail(c, PM_ERR_INVALID, "pm_set_film: sensor tables missing");
    if (f->filter_kind < PM_FILTER_BOX || f->filter_kind > PM_FILTER_GAUSSIAN) return fail(c, PM_ERR_INVALID, "pm_set_film: unknown filter");
    if (f->filter_kind == PM_FILTER_GAUSSIAN && (!f->filter_vals || f->filter_nx < 1 || f->filter_ny < 1))
        return fail(c, PM_ERR_INVALID, "pm_set_film: Gaussian filter needs its FilterSampler table");
    HIPCHK(c, hipSetDevice(c->device));
    RenderState* r = rs(c);
    r->film = *f; r->have_film = true;
    r->sensor.assign(f->sensor_rgb_bar, f->sensor_rgb_bar + 3 * 471);
    r->filter_vals.clear();
    if (f->filter_kind == PM_FILTER_GAUSSIAN) r->filter_vals.assign(f->filter_vals, f->filter_vals + (size_t)f->filter_nx * f->filter_ny);
    r->film.sensor_rgb_bar = nullptr; r->film.filter_vals = nullptr;
    if (r->d_film) { HIPCHK(c, hipFree(r->d_film)); r->d_film = nullptr; }
    r->film_pixels = (size_t)w * h;
    HIPCHK(c, hipMalloc((void**)&r->d_film, r->film_pixels * 4 * sizeof(double)));
    HIPCHK(c, pm_memset_sync(c, r->d_film, r->film_pixels * 4 * sizeof(double)));
    if (!r->d_m9) HIPCHK(c, hipMalloc((void**)&r->d_m9, 9 * sizeof(float)));
    HIPCHK(c, hipMemcpy(r->d_m9, f->output_rgb_from_sensor_rgb, 9 * sizeof(float), hipMemcpyHostToDevice));
    r->scene_dirty = true;
    return PM_OK;
}
extern "C" int pm_film_clear(pm_ctx* c) {
    if (!c || !c->render || !c->render->d_film) return fail(c, PM_ERR_INVALID, "pm_film_clear: no film");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, pm_memset_sync(c, c->render->d_film, c->render->film_pixels * 4 * sizeof(double)));
    return PM_OK;
}
extern "C" int pm_film_read(pm_ctx* c, double* out) {
    if (!c || !out || !c->render || !c->render->d_film) return fail(c, PM_ERR_INVALID, "pm_film_read: no film");
    HIPCHK(c, hipSetDevice(c->device));
    // on the context's stream and waited for: a device-to-device hipMemcpy / hipMemset on the null stream may return before it
    // has run, and the null stream does not order against this (non-blocking) stream
    HIPCHK(c, hipMemcpyAsync(out, c->render->d_film, c->render->film_pixels * 4 * sizeof(double),
                             is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PM_OK;
}
extern "C" int pm_film_add(pm_ctx* c, const double* in) {
    if (!c || !in || !c->render || !c->render->d_film) return fail(c, PM_ERR_INVALID, "pm_film_add: no film");
    HIPCHK(c, hipSetDevice(c->device));
    RenderState* r = c->render;
    const size_t n = r->film_pixels * 4;
    const double* d_in = in; double* tmp = nullptr;
    if (!is_device_ptr(in)) {
        HIPCHK(c, hipMalloc((void**)&tmp, n * sizeof(double)));
        HIPCHK(c, hipMemcpy(tmp, in, n * sizeof(double), hipMemcpyHostToDevice));
        d_in = tmp;
    }
    hipLaunchKernelGGL(film_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, r->d_film, d_in, n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (tmp) HIPCHK(c, hipFree(tmp));
    return PM_OK;
}
extern "C" int pm_film_image(pm_ctx* c, float* rgb) {
    if (!c || !rgb || !c->render || !c->render->d_film) return fail(c, PM_ERR_INVALID, "pm_film_image: no film");
    HIPCHK(c, hipSetDevice(c->device));
    RenderState* r = c->render;
    const size_t n = r->film_pixels;
    float* d_rgb = rgb;
    const bool dev = is_device_ptr(rgb);
    if (!dev) HIPCHK(c, hipMalloc((void**)&d_rgb, n * 3 * sizeof(float)));
    hipLaunchKernelGGL(film_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, r->d_film, d_rgb, n, r->d_m9);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!dev) { HIPCHK(c, hipMemcpy(rgb, d_rgb, n * 3 * sizeof(float), hipMemcpyDeviceToHost)); HIPCHK(c, hipFree(d_rgb)); }
    return PM_OK;
}

template <class T>
static int upload(pm_ctx* c, T** dst, const std::vector<T>& v) {
    if (*dst) { HIPCHK(c, hipFree(*dst)); *dst = nullptr; }
    HIPCHK(c, hipMalloc((void**)dst, std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) HIPCHK(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return PM_OK;
}

static int upload_scene(pm_ctx* c, RenderState* r) {
    const uint32_t n_prims = c->scene.next_prim_id;
    if (r->prim_mat.size() != n_prims) return fail(c, PM_ERR_INVALID, "pm_render: pm_set_prim_materials must give one material id per primitive");
    for (uint32_t id : r->prim_mat) if (id >= r->mats.size()) return fail(c, PM_ERR_INVALID, "pm_render: material id out of range");
    std::vector<float> pool;
    auto add_table = [&](const std::vector<float>& t) { int off = (int)pool.size(); pool.insert(pool.end(), t.begin(), t.end()); return off; };
    std::vector<DMaterial> dm(r->mats.size());
    r->have_textures = false;
    for (size_t i = 0; i < dm.size(); ++i) {
        const PmMaterial& m = r->mats[i];
        dm[i].type = m.type; dm[i].ur = m.u_roughness; dm[i].vr = m.v_roughness; dm[i].remap = m.remap_roughness;
        dm[i].spec.kind = m.spectrum.kind; dm[i].spec.a = m.spectrum.a; dm[i].spec.b = m.spectrum.b; dm[i].spec.c = m.spectrum.c; dm[i].spec.d = m.spectrum.d;
        dm[i].spec.table = r->mat_tables[i].empty() ? -1 : add_table(r->mat_tables[i]);
        dm[i].texture = m.texture; dm[i].su = m.su; dm[i].sv = m.sv; dm[i].du = m.du; dm[i].dv = m.dv;
        dm[i].spec2.kind = m.spectrum2.kind; dm[i].spec2.a = m.spectrum2.a; dm[i].spec2.b = m.spectrum2.b; dm[i].spec2.c = m.spectrum2.c; dm[i].spec2.d = m.spectrum2.d;
        dm[i].spec2.table = r->mat_tables2[i].empty() ? -1 : add_table(r->mat_tables2[i]);
        if (m.texture != PM_TEX_NONE) r->have_textures = true;
    }
    std::vector<DLight> dl(r->lights.size());
    std::vector<int> inf;
    for (size_t i = 0; i < dl.size(); ++i) {
        dl[i].type = r->lights[i].type; dl[i].scale = r->lights[i].scale;
        std::memcpy(dl[i].m, r->lights[i].render_from_light, 48);
        dl[i].table = add_table(r->light_tables[i]);
        dl[i].prim_id = r->lights[i].prim_id; dl[i].two_sided = r->lights[i].two_sided;
        if (dl[i].type == PM_LIGHT_UNIFORM_INFINITE) inf.push_back((int)i);   // integrators/base.rs:57-63
    }
    // GeometricPrimitive.area_light (primitives/single.rs:13-19): one DiffuseAreaLight per emitting patch / triangle
    std::vector<int> prim_light;
    r->have_area_lights = false;
    for (size_t i = 0; i < dl.size(); ++i) {
        if (dl[i].type != PM_LIGHT_DIFFUSE_AREA) continue;
        if (dl[i].prim_id >= n_prims) return fail(c, PM_ERR_INVALID, "pm_render: area light on a primitive id that does not exist");
        if (prim_light.empty()) prim_light.assign(n_prims, -1);
        prim_light[dl[i].prim_id] = (int)i;
        r->have_area_lights = true;
    }
    r->have_sphere_lights = false;
    for (const auto& sph : c->scene.spheres)
        if (!prim_light.empty() && prim_light[sph.prim_id] >= 0) r->have_sphere_lights = true;
    if (r->have_medium) { r->med_off_a = add_table(r->med_sigma_a); r->med_off_s = add_table(r->med_sigma_s); }
    r->n_inf = (int)inf.size();
    r->sensor_off = add_table(r->sensor);
    // FilterSampler distribution (imaging/filter.rs:187-216, sampling/routines.rs:437-462)
    DFilter df{};
    df.kind = r->film.filter_kind; df.rx = r->film.filter_radius[0]; df.ry = r->film.filter_radius[1];
    if (df.kind == PM_FILTER_GAUSSIAN) {
        df.nx = r->film.filter_nx; df.ny = r->film.filter_ny;
        std::vector<float> vals_abs, cond_cdf, cond_int(df.ny), func, cdf;
        for (int v = 0; v < df.ny; ++v) {
            cond_int[v] = pc1d_build(&r->filter_vals[(size_t)v * df.nx], df.nx, -df.rx, df.rx, func, cdf);
            cond_cdf.insert(cond_cdf.end(), cdf.begin(), cdf.end());
        }
        std::vector<float> mfunc, mcdf;
        df.marg_int = pc1d_build(cond_int.data(), df.ny, -df.ry, df.ry, mfunc, mcdf);
        for (float x : r->filter_vals) if (x < 0.0f) return fail(c, PM_ERR_UNSUPPORTED, "pm_set_film: negative filter table entries");
        df.vals = add_table(r->filter_vals); df.cond_cdf = add_table(cond_cdf); df.cond_int = add_table(cond_int);
        df.marg_func = add_table(mfunc); df.marg_cdf = add_table(mcdf);
    }
    r->dfilter = df;
    // shading records.  Every record names its material and its area light.  In a scene WITHOUT spheres record i belongs to
    // primitive id i (direct: no prim_ref table on the device); with spheres the two record arrays are reached through prim_ref
    const bool direct = c->scene.spheres.empty();
    r->direct_records = direct;
    std::vector<uint32_t> ref(direct ? 0 : n_prims);
    std::vector<DShadePatch> sp; std::vector<DShadeSphere> ss;
    const size_t n_flat = c->scene.patches.size() + c->scene.tris.size();
    if (direct) sp.resize(n_flat); else sp.reserve(n_flat);
    auto light_bits = [&](uint32_t prim_id) -> uint32_t {
        const int li = prim_light.empty() ? -1 : prim_light[prim_id];
        return li >= 0 ? ((uint32_t)(li + 1) << 8) : 0u;
    };
    if (dl.size() >= (1u << 24)) return fail(c, PM_ERR_UNSUPPORTED, "pm_render: more than 2^24 - 1 lights");
    auto place = [&](uint32_t prim_id, const DShadePatch& d) {
        if (direct) sp[prim_id] = d;
        else { ref[prim_id] = ((uint32_t)PM_PRIM_PATCH << 30) | (uint32_t)sp.size(); sp.push_back(d); }
    };
    for (const auto& p : c->scene.patches) {
        DShadePatch d{}; std::memcpy(d.p, p.p, 48); d.flags = (p.flip_normal ? 1u : 0u) | light_bits(p.prim_id);
        d.mat = r->prim_mat[p.prim_id];
        if (p.has_normals) {
            d.flags |= 2u; d.nrm_off = (uint32_t)pool.size();
            pool.insert(pool.end(), &p.n[0][0], &p.n[0][0] + 12);
        }
        if (p.has_uv) {
            d.flags |= 4u; d.uv_off = (uint32_t)pool.size();
            pool.insert(pool.end(), &p.uv[0][0], &p.uv[0][0] + 8);
        }
        place(p.prim_id, d);
    }
    for (const auto& t : c->scene.tris) {
        DShadePatch d{}; std::memcpy(d.p, t.v, 36); std::memcpy(d.p + 9, t.v[2], 12); d.flags = light_bits(t.prim_id);
        d.mat = r->prim_mat[t.prim_id];
        place(t.prim_id, d);
    }
    for (const auto& s : c->scene.spheres) {
        DShadeSphere d{};
        std::memcpy(d.rfo, s.render_from_object, 48); std::memcpy(d.ofr, s.object_from_render, 48);
        d.radius = s.radius; d.z_min = s.z_min; d.z_max = s.z_max; d.phi_max = s.phi_max; d.th_min = s.theta_z_min; d.th_max = s.theta_z_max;
        d.flags = (s.reverse_orientation ? 1u : 0u) | light_bits(s.prim_id);
        d.mat = r->prim_mat[s.prim_id];
        ref[s.prim_id] = ((uint32_t)PM_PRIM_SPHERE << 30) | (uint32_t)ss.size(); ss.push_back(d);
    }
    // BilinearPatch::sample's per-patch constants of every area light (pm_shade.hpp patch_area_constants), in DLight.m
    for (size_t i = 0; i < dl.size(); ++i) {
        if (dl[i].type != PM_LIGHT_DIFFUSE_AREA) continue;
        std::memset(dl[i].m, 0, sizeof(dl[i].m));
        if (!direct && (ref[dl[i].prim_id] >> 30) == PM_PRIM_SPHERE) continue;        // a sphere emitter: nothing to precompute
        // (the emitter's record: a triangle's holds v0 v1 v2 v2, the degenerate patch it is)
        const float* q = direct ? sp[dl[i].prim_id].p : sp[ref[dl[i].prim_id] & 0x3FFFFFFFu].p;
        float w4[4]; bool rect = false;
        patch_area_constants(mk3(q[0], q[1], q[2]), mk3(q[3], q[4], q[5]), mk3(q[6], q[7], q[8]), mk3(q[9], q[10], q[11]), w4, &rect);
        for (int k = 0; k < 4; ++k) dl[i].m[k] = w4[k];
        dl[i].m[4] = rect ? 1.0f : 0.0f;
    }
    int rc;
    if ((rc = upload(c, &r->d_pool, pool))) return rc;
    if ((rc = upload(c, &r->d_mats, dm))) return rc;
    if ((rc = upload(c, &r->d_lights, dl))) return rc;
    if ((rc = upload(c, &r->d_inf, inf))) return rc;
    if ((rc = upload(c, &r->d_prim_ref, ref))) return rc;
    if ((rc = upload(c, &r->d_patches, sp))) return rc;
    if ((rc = upload(c, &r->d_spheres, ss))) return rc;
    r->scene_dirty = false;
    r->geom_gen = c->geom_gen;
    return PM_OK;
}

static int ensure_paths(pm_ctx* c, RenderState::PathBuf* b, size_t npaths, int max_depth, bool random_walk) {
    if (b->cap >= npaths && b->depth >= max_depth && (!random_walk || b->pa.rw_f)) return PM_OK;
    for (void* p : b->allocs) HIPCHK(c, hipFree(p));
    b->allocs.clear(); b->cap = 0;
    auto al = [&](void** p, size_t bytes) -> int { HIPCHK(c, hipMalloc(p, bytes)); b->allocs.push_back(*p); return PM_OK; };
    PathArrays& pa = b->pa;
    float4** f4[] = {&pa.ro[0], &pa.ro[1], &pa.rd[0], &pa.rd[1], &pa.beta[0], &pa.beta[1],
                     &pa.hit, &pa.sh_o, &pa.sh_d, &pa.sh_L, &pa.L, &pa.lambda_s, &pa.pdf};
    int rc;
    for (auto p : f4) if ((rc = al((void**)p, npaths * sizeof(float4)))) return rc;
    if ((rc = al((void**)&pa.rng[0], npaths * sizeof(uint2)))) return rc;
    if ((rc = al((void**)&pa.rng[1], npaths * sizeof(uint2)))) return rc;
    if ((rc = al((void**)&pa.fw, npaths * sizeof(float)))) return rc;
    if ((rc = al((void**)&pa.counts, 2 * (size_t)(max_depth + 2) * 4 * PM_CNT_STRIDE))) return rc;
    pa.rw_f = nullptr; pa.cap = (uint32_t)npaths;
    if (random_walk && (rc = al((void**)&pa.rw_f, (size_t)std::max(1, max_depth) * npaths * sizeof(float4)))) return rc;
    b->cap = npaths; b->depth = max_depth;
    return PM_OK;
}

static int render_impl(pm_ctx* c, const PmRenderParams* p, bool debug_one, int dpx, int dpy, int ds, float* out13) {
    if (!c || !p) return fail(c, PM_ERR_INVALID, "pm_render: null argument");
    if (c->device < 0) return fail(c, PM_ERR_INVALID, "pm_render: host-only context");
    if (!c->accel_valid) return fail(c, PM_ERR_NO_ACCEL, "pm_render: pm_build_accel has not been called");
    RenderState* r = rs(c);
    if (!r->have_cam || !r->have_film) return fail(c, PM_ERR_INVALID, "pm_render: camera and film must be set");
    if (p->integrator != PM_INTEGRATOR_SIMPLE_PATH && p->integrator != PM_INTEGRATOR_RANDOM_WALK && p->integrator != PM_INTEGRATOR_VOL_PATH)
        return fail(c, PM_ERR_INVALID, "pm_render: unknown integrator");
    if (p->max_depth < 0 || p->max_depth > 254) return fail(c, PM_ERR_INVALID, "pm_render: max_depth out of range");
    const int* pb = r->film.pixel_bounds;
    int x0 = p->window[0], y0 = p->window[1], x1 = p->window[2], y1 = p->window[3], s0 = p->s0, s1 = p->s1;
    if (debug_one) { x0 = dpx; y0 = dpy; x1 = dpx + 1; y1 = dpy + 1; s0 = ds; s1 = ds + 1; }
    if (x0 < pb[0] || y0 < pb[1] || x1 > pb[2] || y1 > pb[3] || x0 > x1 || y0 > y1) return fail(c, PM_ERR_INVALID, "pm_render: window outside pixel_bounds");
    // IndependentSampler::start_pixel_sample errors for sample_index >= samples_per_pixel (sampler.rs:110-122)
    if (s0 < 0 || s1 > p->spp || s0 > s1) return fail(c, PM_ERR_INVALID, "pm_render: sample window outside [0, spp)");
    HIPCHK(c, hipSetDevice(c->device));
    int rcode;
    if (r->scene_dirty || r->geom_gen != c->geom_gen) if ((rcode = upload_scene(c, r))) return rcode;
    r->stats = PmRenderStats{};
    r->q_closest.assign((size_t)p->max_depth + 1, 0); r->q_shadow.assign((size_t)p->max_depth + 1, 0);
    r->stats.state_bytes_per_path = PM_PATH_BYTES;
    const int W = x1 - x0, H = y1 - y0;
    if (W == 0 || H == 0 || s0 == s1) return PM_OK;
    const int tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
    const size_t npix_pad = (size_t)tiles_x * tiles_y * 64;
    // paths per pass: 2^26 (13 GB of path state + 1 GB of deferred list - sized for a 288 GB device).  Measured
    // (scripts/pass_sweep.sh): every doubling from 2^20 helped the persistent traversal kernels (fewer, longer launches: less
    // tail, fewer small deep-bounce queues) up to 2^25..2^26: C2 +11 %, C4-shaped path +10 % over 2^23.
    // PBRT_MI355_PASS_PATHS overrides (tuning aid; the tests use it to force several passes).
    size_t target = (size_t)1 << 26;
    if (const char* e = std::getenv("PBRT_MI355_PASS_PATHS")) { long long v = std::atoll(e); if (v >= 64 && v <= (1ll << 30)) target = (size_t)v; }
    {   // ... but never more than half of the memory that is free on the device now (another process may share it)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t held = (r->pb[0].cap + r->pb[1].cap) * (size_t)PM_PATH_BYTES;         // what ensure_paths would free first
            const size_t per_path = PM_PATH_BYTES + 16 + 4 + (p->integrator == PM_INTEGRATOR_RANDOM_WALK ? (size_t)std::max(1, p->max_depth) * 16 : 0);
            target = std::min(target, std::max<size_t>((size_t)1 << 16, (free_b + held) / 2 / per_path));
        }
    }
    // Lanes (round 3).  PBRT_MI355_LANES: 1 = one stream as before; 2 = two lanes whenever the call has at least two sample
    // indices; unset = two lanes for calls of at least 2^22 paths (below that the launches are too short to gain from it).  The
    // path-state memory is the same either way: with two lanes a pass is half as large.  scripts/overlap_probe.py was the
    // feasibility test (two contexts on two host threads: +11 % on C2); profiles/r03/ab_lanes.log is the measurement.
    int lanes = 0;
    if (const char* e = std::getenv("PBRT_MI355_LANES")) lanes = std::atoi(e);
    if (lanes != 1 && lanes != 2) lanes = npix_pad * (size_t)(s1 - s0) >= ((size_t)1 << 22) ? 2 : 1;
    if (debug_one || (s1 - s0) < 2 || std::getenv("PBRT_MI355_VERIFY") || std::getenv("PBRT_MI355_REORDER")) lanes = 1;
    int s_per_pass = (int)std::max<size_t>(1, std::min<size_t>((size_t)(s1 - s0), target / npix_pad));
    if (lanes == 2) s_per_pass = (int)std::max<size_t>(1, std::min<size_t>((size_t)(s1 - s0 + 1) / 2, (target / 2) / npix_pad));
    const size_t max_paths = npix_pad * (size_t)s_per_pass;
    if (max_paths > 0x7FFFFFFFull) return fail(c, PM_ERR_INVALID, "pm_render: window too large for one pass");
    const bool random_walk = p->integrator == PM_INTEGRATOR_RANDOM_WALK;
    if (random_walk && (r->have_textures || r->have_area_lights))
        return fail(c, PM_ERR_UNSUPPORTED, "pm_render: the RandomWalk kernel supports neither textures nor area lights (extensions of SimplePath / VolPath)");
    for (int l = 0; l < lanes; ++l) if ((rcode = ensure_paths(c, &r->pb[l], max_paths, p->max_depth, random_walk))) return rcode;
    if ((rcode = ensure_defer(c, max_paths))) return rcode;
    if (lanes == 2) {
        if (!r->stream1) HIPCHK(c, hipStreamCreateWithFlags(&r->stream1, hipStreamNonBlocking));
        if (!r->d_work1) HIPCHK(c, hipMalloc((void**)&r->d_work1, 3 * PM_WORK_SLOTS * sizeof(uint32_t)));
        if ((rcode = ensure_defer_buf(c, max_paths, &r->d_defer1, &r->d_slow1, &r->defer_cap1))) return rcode;
    }
    for (hipEvent_t* e : {&r->ev_go, &r->ev_acc[0], &r->ev_acc[1], &r->ev_join}) if (!*e) HIPCHK(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    const bool has_spheres = !c->scene.spheres.empty();
    const int kind = pm_walk_kind(c);      // tree-walk kernel variant
    // PBRT_MI355_REORDER=<bits per axis, 1..9> (experiment, off by default - measured in profiles/r03/ab_reorder.log): before the
    // closest-hit launch of every bounce after the first, sort the queue by (Morton code of the ray origin, direction octant) and
    // walk it in that order.  PBRT_MI355_REORDER_OCT_HIGH=1 puts the octant above the Morton code.  Costs one host read of the
    // queue length per bounce.  The film does not change (hits do not depend on the order rays are walked in).
    int reorder_bits = 0, reorder_oct_high = 0;
    if (const char* e = std::getenv("PBRT_MI355_REORDER")) { int v = std::atoi(e); if (v >= 1 && v <= 9) reorder_bits = v; }
    if (const char* e = std::getenv("PBRT_MI355_REORDER_OCT_HIGH")) reorder_oct_high = std::atoi(e) == 1;
    if (c->flat.tris != nullptr || has_spheres) reorder_bits = 0;
    if (reorder_bits && r->sort_cap < max_paths) {
        for (uint32_t*& q : r->d_sort) { if (q) (void)hipFree(q); q = nullptr; HIPCHK(c, hipMalloc((void**)&q, max_paths * sizeof(uint32_t))); }
        if (r->d_sort_tmp) (void)hipFree(r->d_sort_tmp);
        r->d_sort_tmp = nullptr; r->sort_tmp_bytes = 0;
        if (pm_sort_pairs_u32(c->stream, nullptr, &r->sort_tmp_bytes, r->d_sort[0], r->d_sort[1], r->d_sort[2], r->d_sort[3], (uint32_t)max_paths, 30))
            return fail(c, PM_ERR_HIP, "pm_render: radix sort size query failed");
        HIPCHK(c, hipMalloc(&r->d_sort_tmp, std::max<size_t>(r->sort_tmp_bytes, 16)));
        r->sort_cap = max_paths;
    }

    RenderConst rc{};
    std::memcpy(rc.cfr, r->cam.camera_from_raster, 64); std::memcpy(rc.rfc, r->cam.render_from_camera, 48);
    rc.lens_radius = r->cam.lens_radius; rc.focal_distance = r->cam.focal_distance;
    rc.filter = r->dfilter; rc.sensor_table = r->sensor_off;
    rc.imaging_ratio = r->film.imaging_ratio; rc.max_component_value = r->film.max_component_value;
    rc.integrator = p->integrator; rc.max_depth = p->max_depth; rc.sample_lights = p->sample_lights; rc.sample_bsdf = p->sample_bsdf;
    rc.seed = p->seed; rc.n_lights = (int)r->lights.size(); rc.n_infinite = r->n_inf;
    rc.medium_on = (p->integrator == PM_INTEGRATOR_VOL_PATH && r->have_medium) ? 1 : 0;
    rc.terminal_emission = (r->n_inf > 0 || r->have_area_lights) ? 1 : 0;
    rc.med_sigma_a = r->med_off_a; rc.med_sigma_s = r->med_off_s; rc.med_g = r->medium.g;
    std::memcpy(rc.med_bounds, r->medium.bounds, sizeof(rc.med_bounds));
    const bool shade_ext = rc.medium_on || r->have_textures || r->have_sphere_lights;      // (sphere emitters: EXT kernels only)
    const bool shade_area = r->have_area_lights && !shade_ext;
    bool force_ext = false;         // A/B and tests: PBRT_MI355_SHADE_EXT=1 routes area-light scenes through the EXT kernel as round 2 did
    if (const char* e = std::getenv("PBRT_MI355_SHADE_EXT")) force_ext = std::atoi(e) == 1 && r->have_area_lights;
    {   // SceneData::new -> light.preprocess(scene_bounds): Bounds3f::bounding_sphere (core/bounds.rs:588-597)
        const pm::Box& b = c->accel.bounds;
        float ctr[3], d2 = 0.0f; bool inside = true;
        for (int k = 0; k < 3; ++k) { ctr[k] = (b.lo[k] + b.hi[k]) / 2.0f; inside = inside && ctr[k] >= b.lo[k] && ctr[k] <= b.hi[k]; }
        const float dx = ctr[0] - b.hi[0], dy = ctr[1] - b.hi[1], dz = ctr[2] - b.hi[2];
        d2 = dx * dx + dy * dy + dz * dz;
        rc.scene_radius = inside ? std::sqrt(d2) : 0.0f;
    }
    rc.win_x0 = x0; rc.win_y0 = y0; rc.win_w = W; rc.win_h = H; rc.tiles_x = tiles_x; rc.npix_pad = (int)npix_pad;
    rc.film_x0 = pb[0]; rc.film_y0 = pb[1]; rc.film_w = pb[2] - pb[0];
    SceneArrays sa{c->d_nodes, c->d_prims, r->direct_records ? nullptr : r->d_prim_ref, r->d_patches, r->d_spheres, r->d_mats, r->d_lights,
                   r->d_inf, r->d_pool, r->have_area_lights ? 1 : 0};
    const int nb = p->max_depth + 2;
    const size_t cnt_words = 2 * (size_t)nb * PM_CNT_STRIDE;
    if (r->h_counts_words < cnt_words) {
        for (uint32_t*& h : r->h_counts) { if (h) (void)hipHostFree(h); h = nullptr; HIPCHK(c, hipHostMalloc((void**)&h, cnt_words * sizeof(uint32_t))); }
        r->h_counts_words = cnt_words;
    }
    const size_t max_launch = 3 * (size_t)(p->max_depth + 1) + 3;
    for (int l = 0; l < lanes; ++l)
        while (r->ev_pool[l].size() < max_launch + 1) { hipEvent_t e; HIPCHK(c, hipEventCreate(&e)); r->ev_pool[l].push_back(e); }
    unsigned long long cnt_before[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(c, hipMemcpy(cnt_before, c->d_counters, sizeof(cnt_before), hipMemcpyDeviceToHost));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    // PBRT_MI355_VERIFY=1: see verify_hits_kernel
    bool verify = false;
    if (const char* e = std::getenv("PBRT_MI355_VERIFY")) verify = std::atoi(e) == 1;
    float4* verify_hits = nullptr; uint32_t* verify_bad = nullptr; VerifyRec* verify_recs = nullptr;
    if (verify) {
        HIPCHK(c, hipMalloc((void**)&verify_hits, max_paths * sizeof(float4)));
        HIPCHK(c, hipMalloc((void**)&verify_bad, sizeof(uint32_t)));
        HIPCHK(c, hipMalloc((void**)&verify_recs, 64 * sizeof(VerifyRec)));
        HIPCHK(c, pm_memset_sync(c, verify_bad, sizeof(uint32_t)));
    }
    // Two lanes: pass k runs on lane k & 1.  What orders them: (i) lane 1 starts behind whatever the caller's last call left on
    // the context's stream (ev_go, recorded there - after the first `stagger` launches of pass 0, so that the two lanes are in
    // different phases of a bounce from the start: a sweep beside a shade kernel, not two of a kind); (ii) the film is added
    // to in sample-index order: the accumulate launch of pass k waits for that of pass k - 1 (ev_acc); (iii) a lane's buffers
    // are reused by its next pass, which the host issues once it has collected the counts of the previous one.
    bool coh_on = true;
    if (const char* e = std::getenv("PBRT_MI355_COH")) coh_on = std::atoi(e) != 0;
    int stagger = 2, grid_div = 1;
    if (const char* e = std::getenv("PBRT_MI355_STAGGER")) { int v = std::atoi(e); if (v >= 0 && v <= 64) stagger = v; }
    if (const char* e = std::getenv("PBRT_MI355_LANE_GRID_DIV")) { int v = std::atoi(e); if (v >= 1 && v <= 4) grid_div = v; }
    if (lanes == 1) grid_div = 1;
    struct LaneRun { bool busy = false; std::vector<int> ev_class; int s_count = 0; };
    LaneRun run[2];
    hipStream_t streams[2] = {c->stream, r->stream1};
    // whatever way this function is left (an error return in the middle of a frame included), nothing of it is still running on
    // either lane afterwards: the call is blocking, and the next call reuses the lanes' buffers
    struct Drain { hipStream_t a, b; ~Drain() { if (b) (void)hipStreamSynchronize(b); (void)hipStreamSynchronize(a); } } drain{c->stream, lanes == 2 ? r->stream1 : nullptr};
    uint32_t* works[2] = {c->d_work, r->d_work1};
    uint4* defers[2] = {c->d_defer, r->d_defer1};
    uint32_t* slows[2] = {c->d_slow, r->d_slow1};
    const size_t defer_caps[2] = {c->defer_cap, r->defer_cap1};
    // collect what a lane's finished pass left: per-class times, queue lengths
    auto harvest = [&](int l) -> int {
        LaneRun& lr = run[l];
        if (!lr.busy) return PM_OK;
        HIPCHK(c, hipStreamSynchronize(streams[l]));
        for (size_t e = 0; e < lr.ev_class.size(); ++e) {
            float ems = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ems, r->ev_pool[l][e], r->ev_pool[l][e + 1]));
            r->stats.ms_by_kernel[lr.ev_class[e]] += ems;
            r->stats.launches_by_kernel[lr.ev_class[e]] += 1;
        }
        const uint32_t* hcounts = r->h_counts[l];
        r->stats.passes++;
        r->stats.samples += (uint64_t)W * H * lr.s_count;
        // closest rays: every live path in the queue of bounce b (padded slots are not live)
        const uint64_t pad = (uint64_t)(npix_pad - (size_t)W * H) * lr.s_count;
        for (int b = 0; b <= p->max_depth; ++b) {
            r->q_closest[b] += hcounts[(size_t)b * PM_CNT_STRIDE] - (b == 0 ? pad : 0);
            r->q_shadow[b] += hcounts[(size_t)(nb + b) * PM_CNT_STRIDE];
        }
        for (int b = 0; b <= p->max_depth; ++b) r->stats.closest_rays += hcounts[(size_t)b * PM_CNT_STRIDE] - (b == 0 ? pad : 0);
        for (int b = 0; b < nb; ++b) r->stats.shadow_rays += hcounts[(size_t)(nb + b) * PM_CNT_STRIDE];
        lr.busy = false;
        return PM_OK;
    };
    int pass_index = 0;
    for (int sb = s0; sb < s1; sb += s_per_pass, ++pass_index) {
        const int l = lanes == 2 ? (pass_index & 1) : 0;
        if ((rcode = harvest(l))) return rcode;
        LaneRun& lr = run[l];
        std::vector<int>& ev_class = lr.ev_class;
        ev_class.clear();
        hipStream_t st = streams[l];
        PathArrays& pa = r->pb[l].pa;
        uint32_t* const d_work = works[l]; uint4* const d_defer = defers[l]; uint32_t* const d_slow = slows[l];
        const uint32_t defer_cap = (uint32_t)defer_caps[l];
        bool go_pending = lanes == 2 && pass_index == 0;
        auto mark = [&](int cls) {
            if (go_pending && (int)ev_class.size() == stagger) { (void)hipEventRecord(r->ev_go, st); go_pending = false; }
            (void)hipEventRecord(r->ev_pool[l][ev_class.size()], st); ev_class.push_back(cls);
        };
        if (lanes == 2 && pass_index == 1) HIPCHK(c, hipStreamWaitEvent(st, r->ev_go, 0));
        rc.s_base = sb; rc.s_count = std::min(s_per_pass, s1 - sb);
        lr.s_count = rc.s_count;
        const uint32_t npaths = (uint32_t)(npix_pad * (size_t)rc.s_count);
        HIPCHK(c, hipMemsetAsync(pa.counts, 0, 2 * (size_t)nb * 4 * PM_CNT_STRIDE, st));
        HIPCHK(c, hipMemsetAsync(d_work, 0, 3 * PM_WORK_SLOTS * sizeof(uint32_t), st));
        int wslot = 0;      // one work counter per traversal launch (re-zeroed when the slots run out)
        const dim3 gr((npaths + RB - 1) / RB);
        const dim3 gt(std::min<uint32_t>((npaths + TB - 1) / TB, (uint32_t)c->n_cu * PM_BLOCKS_PER_CU / grid_div));
        const dim3 gf(std::min<uint32_t>((npaths + TB - 1) / TB, (uint32_t)c->n_cu * PM_FLAT_BLOCKS_PER_CU / grid_div));
        const bool flat = c->flat.tris != nullptr, fpatch = c->flat.n_patch > 0;
        mark(0);
        hipLaunchKernelGGL(raygen_kernel, gr, dim3(RB), 0, st, rc, sa, pa, npaths);
        uint32_t gen = 0;       // generation of the path-state arrays that holds the current queue
        for (int b = 0; b <= p->max_depth; ++b) {
            uint32_t* cnt_in = pa.counts + (size_t)b * PM_CNT_STRIDE; uint32_t* cnt_out = pa.counts + (size_t)(b + 1) * PM_CNT_STRIDE;
            uint32_t* cnt_sh = pa.counts + (size_t)(nb + b) * PM_CNT_STRIDE;
            if (wslot + 2 > PM_WORK_SLOTS) {   // stream-ordered: every earlier traversal kernel has finished with its counter
                HIPCHK(c, hipMemsetAsync(d_work, 0, 3 * PM_WORK_SLOTS * sizeof(uint32_t), st));
                wslot = 0;
            }
            mark(1);
            const uint32_t* perm = nullptr;
            if (reorder_bits && b >= 1) {
                uint32_t n_q = 0;
                HIPCHK(c, hipMemcpyAsync(&n_q, cnt_in, sizeof(n_q), hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
                if (n_q >= 4096u) {
                    const pm::Box& bx = c->accel.bounds;
                    const float q = (float)(1u << reorder_bits);
                    auto sc1 = [&](int k) { const float e = bx.hi[k] - bx.lo[k]; return e > 0.0f ? q / e : 0.0f; };
                    hipLaunchKernelGGL(reorder_key_kernel, dim3((n_q + RB - 1) / RB), dim3(RB), 0, st, pa, gen, n_q, mk3(bx.lo[0], bx.lo[1], bx.lo[2]),
                                       mk3(sc1(0), sc1(1), sc1(2)), reorder_bits, reorder_oct_high, r->d_sort[0], r->d_sort[2]);
                    size_t tb = r->sort_tmp_bytes;
                    if (pm_sort_pairs_u32(st, r->d_sort_tmp, &tb, r->d_sort[0], r->d_sort[1], r->d_sort[2], r->d_sort[3], n_q, 3 * reorder_bits + 3))
                        return fail(c, PM_ERR_HIP, "pm_render: radix sort failed");
                    perm = r->d_sort[3];
                }
            }
            if (flat) {
                // bounce 0: the queue is the identity (entry i = slot i = pixel-in-tile x sample), so a wave holds the camera rays of
                // one 8 x 8 tile: the COH instantiation culls the table per wave (pm_device.hpp trace_flat; PBRT_MI355_COH=0: off)
                const bool coh = b == 0 && coh_on;
                auto kernel = c->counting ? (fpatch ? closest_flat_kernel<true, true> : closest_flat_kernel<true, false>)
                                          : coh ? (fpatch ? closest_flat_kernel<false, true, true> : closest_flat_kernel<false, false, true>)
                                                : (fpatch ? closest_flat_kernel<false, true> : closest_flat_kernel<false, false>);
                hipLaunchKernelGGL(kernel, gf, dim3(TB), 0, st, c->flat, sa, pa, gen, cnt_in, d_defer, defer_cap, d_work + PM_WORK_SLOTS + wslot, d_slow, c->d_counters);
            } else {
                auto kernel = c->counting ? (kind == PM_WALK_MIXED ? closest_kernel<true, PM_WALK_MIXED> : kind == PM_WALK_PATCH ? closest_kernel<true, PM_WALK_PATCH> : closest_kernel<true, PM_WALK_TRI>)
                                          : (kind == PM_WALK_MIXED ? closest_kernel<false, PM_WALK_MIXED> : kind == PM_WALK_PATCH ? closest_kernel<false, PM_WALK_PATCH> : closest_kernel<false, PM_WALK_TRI>);
                hipLaunchKernelGGL(kernel, gt, dim3(TB), 0, st, sa, pa, gen, cnt_in, d_work + wslot, c->refill, d_defer, defer_cap, c->d_counters, perm);
            }
            if (has_spheres || flat) {     // second pass: dense exact tests of what was deferred (timed with the closest-hit class)
                if (c->counting) hipLaunchKernelGGL(closest_fix_kernel<true>, gt, dim3(TB), 0, st, c->flat, sa, pa, gen, d_defer, defer_cap, d_work + PM_WORK_SLOTS + wslot, flat ? d_slow : nullptr, c->d_counters);
                else hipLaunchKernelGGL(closest_fix_kernel<false>, gt, dim3(TB), 0, st, c->flat, sa, pa, gen, d_defer, defer_cap, d_work + PM_WORK_SLOTS + wslot, flat ? d_slow : nullptr, c->d_counters);
            }
            ++wslot;
            if (verify && !c->counting && !flat && !has_spheres) {
                // the same queue through the counting walk (the hits it leaves in pa.hit are the ones the pass goes on with)
                HIPCHK(c, hipMemcpyAsync(verify_hits, pa.hit, (size_t)npaths * sizeof(float4), hipMemcpyDeviceToDevice, st));
                if (wslot + 2 > PM_WORK_SLOTS) { HIPCHK(c, hipMemsetAsync(d_work, 0, 3 * PM_WORK_SLOTS * sizeof(uint32_t), st)); wslot = 0; }
                auto kernel = kind == PM_WALK_MIXED ? closest_kernel<true, PM_WALK_MIXED> : kind == PM_WALK_PATCH ? closest_kernel<true, PM_WALK_PATCH> : closest_kernel<true, PM_WALK_TRI>;
                hipLaunchKernelGGL(kernel, gt, dim3(TB), 0, st, sa, pa, gen, cnt_in, d_work + wslot, c->refill, d_defer, defer_cap, c->d_counters, (const uint32_t*)nullptr);
                ++wslot;
                hipLaunchKernelGGL(verify_hits_kernel, gr, dim3(RB), 0, st, pa, gen, cnt_in, verify_hits, (uint32_t)b, verify_bad, verify_recs, 64u);
            }
            mark(2);
            if (random_walk) hipLaunchKernelGGL(shade_rw_kernel, gr, dim3(RB), 0, st, rc, sa, pa, gen, cnt_in, cnt_out);
            else if (r->have_sphere_lights) hipLaunchKernelGGL((shade_kernel<true, true>), gr, dim3(RB), 0, st, rc, sa, pa, gen | (b == 0 ? 2u : 0u), cnt_in, cnt_out, cnt_sh);
            else if (shade_ext || force_ext) hipLaunchKernelGGL((shade_kernel<true, false>), gr, dim3(RB), 0, st, rc, sa, pa, gen | (b == 0 ? 2u : 0u), cnt_in, cnt_out, cnt_sh);
            else if (shade_area) hipLaunchKernelGGL((shade_kernel<false, true>), gr, dim3(RB), 0, st, rc, sa, pa, gen | (b == 0 ? 2u : 0u), cnt_in, cnt_out, cnt_sh);
            else hipLaunchKernelGGL((shade_kernel<false, false>), gr, dim3(RB), 0, st, rc, sa, pa, gen | (b == 0 ? 2u : 0u), cnt_in, cnt_out, cnt_sh);
            if (!random_walk && p->sample_lights && b < p->max_depth) {
                mark(3);
                if (flat) {
                    auto kernel = c->counting ? (fpatch ? shadow_flat_kernel<true, true> : shadow_flat_kernel<true, false>)
                                              : (fpatch ? shadow_flat_kernel<false, true> : shadow_flat_kernel<false, false>);
                    hipLaunchKernelGGL(kernel, gf, dim3(TB), 0, st, c->flat, sa, pa, cnt_sh, d_defer, defer_cap, d_work + PM_WORK_SLOTS + wslot, d_slow, c->d_counters);
                } else {
                    auto kernel = c->counting ? (kind == PM_WALK_MIXED ? shadow_kernel<true, PM_WALK_MIXED> : kind == PM_WALK_PATCH ? shadow_kernel<true, PM_WALK_PATCH> : shadow_kernel<true, PM_WALK_TRI>)
                                              : (kind == PM_WALK_MIXED ? shadow_kernel<false, PM_WALK_MIXED> : kind == PM_WALK_PATCH ? shadow_kernel<false, PM_WALK_PATCH> : shadow_kernel<false, PM_WALK_TRI>);
                    hipLaunchKernelGGL(kernel, gt, dim3(TB), 0, st, sa, pa, cnt_sh, d_work + wslot, c->refill, d_defer, defer_cap, c->d_counters);
                }
                if (has_spheres || flat) {
                    if (c->counting) hipLaunchKernelGGL(shadow_fix_kernel<true>, gt, dim3(TB), 0, st, c->flat, sa, pa, d_defer, defer_cap, d_work + PM_WORK_SLOTS + wslot, flat ? d_slow : nullptr, c->d_counters);
                    else hipLaunchKernelGGL(shadow_fix_kernel<false>, gt, dim3(TB), 0, st, c->flat, sa, pa, d_defer, defer_cap, d_work + PM_WORK_SLOTS + wslot, flat ? d_slow : nullptr, c->d_counters);
                }
                ++wslot;
            }
            gen ^= 1u;
            r->stats.kernels_launched += 3;
        }
        if (go_pending) { HIPCHK(c, hipEventRecord(r->ev_go, st)); go_pending = false; }      // (a pass of fewer launches than `stagger`)
        if (!debug_one) {
            mark(4);        // (with two lanes the accumulate class also holds the wait for the other lane's previous pass)
            if (lanes == 2 && pass_index > 0) HIPCHK(c, hipStreamWaitEvent(st, r->ev_acc[l ^ 1], 0));
            hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)((npix_pad + RB - 1) / RB)), dim3(RB), 0, st, rc, sa, pa, r->d_film);
            if (lanes == 2) HIPCHK(c, hipEventRecord(r->ev_acc[l], st));
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(r->ev_pool[l][ev_class.size()], st));
        HIPCHK(c, hipMemcpyAsync(r->h_counts[l], pa.counts, cnt_words * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        lr.busy = true;
        if (lanes == 1 && (rcode = harvest(0))) return rcode;
    }
    if ((rcode = harvest(0)) || (rcode = harvest(1))) return rcode;
    if (lanes == 2 && pass_index > 1) {      // the context's stream - and every later call on it - comes after lane 1's work
        HIPCHK(c, hipEventRecord(r->ev_join, r->stream1));
        HIPCHK(c, hipStreamWaitEvent(c->stream, r->ev_join, 0));
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (verify) {
        uint32_t n_bad = 0; VerifyRec recs[64];
        HIPCHK(c, hipMemcpy(&n_bad, verify_bad, sizeof(n_bad), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(recs, verify_recs, sizeof(recs), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[pm verify] %u closest hits of the timed walk differ from the counting walk\n", n_bad);
        for (uint32_t k = 0; k < n_bad && k < 64; ++k) {
            const VerifyRec& v = recs[k];
            uint32_t ob[3], db[3]; std::memcpy(ob, v.o, 12); std::memcpy(db, v.d, 12);
            uint32_t ta, tb; std::memcpy(&ta, &v.timed.x, 4); std::memcpy(&tb, &v.counting.x, 4);
            std::fprintf(stderr, "[pm verify] bounce %u entry %u slot %u: o %08x %08x %08x d %08x %08x %08x  timed id %u t %.9g u %.9g v %.9g | counting id %u t %.9g u %.9g v %.9g\n",
                         v.bounce, v.index, *(const uint32_t*)&v.o[3], ob[0], ob[1], ob[2], db[0], db[1], db[2], ta, v.timed.y, v.timed.z, v.timed.w,
                         tb, v.counting.y, v.counting.z, v.counting.w);
        }
        (void)hipFree(verify_hits); (void)hipFree(verify_bad); (void)hipFree(verify_recs);
    }
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    r->stats.kernel_ms = ms; c->last_ms = ms;
    {
        unsigned long long cnt_after[8];
        HIPCHK(c, hipMemcpy(cnt_after, c->d_counters, sizeof(cnt_after), hipMemcpyDeviceToHost));
        r->stats.closest_nodes = cnt_after[1] - cnt_before[1]; r->stats.closest_prims = cnt_after[2] - cnt_before[2];
        r->stats.shadow_nodes = cnt_after[4] - cnt_before[4]; r->stats.shadow_prims = cnt_after[5] - cnt_before[5];
        r->stats.exact_tests = cnt_after[6] - cnt_before[6]; r->stats.fallback_walks = cnt_after[7] - cnt_before[7];
        if (cnt_after[3] != cnt_before[3]) return fail(c, PM_ERR_STACK_OVERFLOW, "pm_render: traversal stack overflow");
    }
    if (debug_one) {
        float4 h[3]; float fwv;
        const PathArrays& pa = r->pb[0].pa;
        HIPCHK(c, hipMemcpy(&h[0], pa.L, 16, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&h[1], pa.lambda_s, 16, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&h[2], pa.pdf, 16, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&fwv, pa.fw, 4, hipMemcpyDeviceToHost));
        std::memcpy(out13, h, 48);
        std::memcpy(out13 + 12, &fwv, 4);
    }
    return PM_OK;
}

extern "C" int pm_render(pm_ctx* c, const PmRenderParams* p, uint32_t struct_bytes) {
    if (c && p) PM_ABI_CHECK(c, PmRenderParams, struct_bytes, "pm_render");
    return render_impl(c, p, false, 0, 0, 0, nullptr);
}
extern "C" int pm_debug_sample(pm_ctx* c, const PmRenderParams* p, int32_t px, int32_t py, int32_t s, float* out13, uint32_t struct_bytes) {
    if (!out13) return fail(c, PM_ERR_INVALID, "pm_debug_sample: null out");
    if (c && p) PM_ABI_CHECK(c, PmRenderParams, struct_bytes, "pm_debug_sample");
    return render_impl(c, p, true, px, py, s, out13);
}
// test hook: the first n uniforms of the sample stream of (pixel, sample index, dimension), drawn on the device
__global__ void debug_uniforms_kernel(int px, int py, uint32_t seed, unsigned long long sample_index, unsigned long long dim, int n,
                                      float* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng rng = rng_start(px, py, seed, sample_index, dim);
    for (int i = 0; i < n; ++i) out[i] = rng_uniform(&rng);
}
extern "C" int pm_debug_uniforms(pm_ctx* c, int32_t px, int32_t py, uint32_t seed, uint64_t sample_index, uint64_t dimension,
                                 int32_t n, float* out) {
    if (!c || !out || n < 0) return fail(c, PM_ERR_INVALID, "pm_debug_uniforms: bad argument");
    if (c->device < 0) return fail(c, PM_ERR_INVALID, "pm_debug_uniforms: host-only context");
    if (n == 0) return PM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    float* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, (size_t)n * sizeof(float)));
    hipLaunchKernelGGL(debug_uniforms_kernel, dim3(1), dim3(64), 0, c->stream, px, py, seed, (unsigned long long)sample_index,
                       (unsigned long long)dimension, n, d);
    hipError_t e = hipMemcpyAsync(out, d, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, PM_ERR_HIP, std::string("pm_debug_uniforms: ") + hipGetErrorString(e));
    return PM_OK;
}
__global__ void debug_math_kernel(int which, const float* __restrict__ x, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float s = 0.0f, c = 0.0f, r = 0.0f;
    switch (which) {
        case 0: r = cr_atanh(v); break;
        case 1: r = cr_cosh(v); break;
        case 2: r = cr_exp(v); break;
        case 3: r = cr_log(v); break;
        case 4: cr_sincos_2pi(v, &s, &c); r = s; break;
        case 5: cr_sincos_2pi(v, &s, &c); r = c; break;
        case 6: cr_sincos_quarter(v, &s, &c); r = s; break;
        default: cr_sincos_quarter(v, &s, &c); r = c; break;
    }
    out[i] = r;
}
extern "C" int pm_debug_math(pm_ctx* c, int32_t which, const float* x, int32_t n, float* out) {
    if (!c || !x || !out || n < 0 || which < 0 || which > 7) return fail(c, PM_ERR_INVALID, "pm_debug_math: bad argument");
    if (c->device < 0) return fail(c, PM_ERR_INVALID, "pm_debug_math: host-only context");
    if (n == 0) return PM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    float *dx = nullptr, *dout = nullptr;
    HIPCHK(c, hipMalloc((void**)&dx, (size_t)n * sizeof(float)));
    hipError_t e = hipMalloc((void**)&dout, (size_t)n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(dx, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, (int)which, dx, (int)n, dout);
        e = hipMemcpyAsync(out, dout, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dx);
    if (dout) (void)hipFree(dout);
    if (e != hipSuccess) return fail(c, PM_ERR_HIP, std::string("pm_debug_math: ") + hipGetErrorString(e));
    return PM_OK;
}
// test hook: Sphere::sample_with_context of the sphere `prim_id` of the scene the last pm_render uploaded, item i: in8 = ctx_p[3],
// p_origin[3], u[2]; out8 = ok, p[3], n[3], pdf
__global__ void debug_sphere_sample_kernel(SceneArrays sc, uint32_t prim_id, const float* __restrict__ in8, int n, float* __restrict__ out8) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ref = shade_ref(sc, prim_id);
    const float* q = in8 + 8 * (size_t)i;
    f2 u; u.x = q[6]; u.y = q[7];
    ShapeSample ss; ss.p = mk3(0.f, 0.f, 0.f); ss.n = ss.p; ss.pdf = 0.f;
    const bool ok = sphere_sample_ctx(sc.spheres[ref & 0x3FFFFFFFu], mk3(q[0], q[1], q[2]), mk3(q[3], q[4], q[5]), u, &ss);
    float* o = out8 + 8 * (size_t)i;
    o[0] = ok ? 1.0f : 0.0f; o[1] = ss.p.x; o[2] = ss.p.y; o[3] = ss.p.z; o[4] = ss.n.x; o[5] = ss.n.y; o[6] = ss.n.z; o[7] = ss.pdf;
}
extern "C" int pm_debug_sphere_sample(pm_ctx* c, uint32_t prim_id, const float* in8, int32_t n, float* out8) {
    if (!c || !in8 || !out8 || n < 0) return fail(c, PM_ERR_INVALID, "pm_debug_sphere_sample: bad argument");
    if (c->device < 0 || !c->render) return fail(c, PM_ERR_INVALID, "pm_debug_sphere_sample: nothing rendered on this context");
    RenderState* r = c->render;
    if (r->scene_dirty || r->geom_gen != c->geom_gen || r->direct_records) return fail(c, PM_ERR_INVALID, "pm_debug_sphere_sample: render the scene (with spheres) first");
    bool is_sphere = false;
    for (const auto& sph : c->scene.spheres) if (sph.prim_id == prim_id) is_sphere = true;
    if (!is_sphere) return fail(c, PM_ERR_INVALID, "pm_debug_sphere_sample: prim_id is not a sphere");
    if (n == 0) return PM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    float *din = nullptr, *dout = nullptr;
    HIPCHK(c, hipMalloc((void**)&din, (size_t)n * 8 * sizeof(float)));
    hipError_t e = hipMalloc((void**)&dout, (size_t)n * 8 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(din, in8, (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        SceneArrays sa{c->d_nodes, c->d_prims, r->d_prim_ref, r->d_patches, r->d_spheres, r->d_mats, r->d_lights, r->d_inf, r->d_pool, 0};
        hipLaunchKernelGGL(debug_sphere_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, sa, prim_id, din, (int)n, dout);
        e = hipMemcpyAsync(out8, dout, (size_t)n * 8 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(din);
    if (dout) (void)hipFree(dout);
    if (e != hipSuccess) return fail(c, PM_ERR_HIP, std::string("pm_debug_sphere_sample: ") + hipGetErrorString(e));
    return PM_OK;
}
extern "C" int pm_get_queue_counts(pm_ctx* c, uint64_t* closest, uint64_t* shadow, uint32_t cap, uint32_t* n_bounces) {
    if (!c || !c->render) return fail(c, PM_ERR_INVALID, "pm_get_queue_counts: nothing rendered");
    const RenderState* r = c->render;
    if (n_bounces) *n_bounces = (uint32_t)r->q_closest.size();
    for (uint32_t b = 0; b < cap && b < r->q_closest.size(); ++b) {
        if (closest) closest[b] = r->q_closest[b];
        if (shadow) shadow[b] = r->q_shadow[b];
    }
    return PM_OK;
}
extern "C" int pm_get_render_stats(pm_ctx* c, PmRenderStats* out, uint32_t struct_bytes) {
    if (!c || !out || !c->render) return fail(c, PM_ERR_INVALID, "pm_get_render_stats: nothing rendered");
    PM_ABI_CHECK(c, PmRenderStats, struct_bytes, "pm_get_render_stats");
    *out = c->render->stats;
    return PM_OK;
}
