// C ABI of libpbrt_mi355.so (include/pbrt_mi355.h): context, geometry upload, wide-BVH build,
// batched closest-hit / any-hit kernels.  gfx950 only.
#include <cstdlib>
#include <mutex>

#include "pm_ctx.h"
#include "pm_device.hpp"

// ------------------------------------------------------------------------------------------ scratch of the no-alloc entry point
// pm_trace_closest_noalloc (include/pbrt_mi355.h) launches the same kernels as pm_trace_closest but allocates nothing: the
// device memory those kernels need beyond rays and hits - work counters, error counters, the deferred-primitive list and
// the slow-ray list - lives in this file-scope array.  Every GPU the library runs on holds its own copy until the process
// ends (15.8 MB).  In host code the NAME `scratch` is only a stand-in for the symbol: the address a kernel can use is looked
// up once per device with hipGetSymbolAddress (pm_create, with that device current) and kept in g_scratch_ptr[device] /
// pm_ctx::scratch; host code uses that pointer, never the name.
#define PM_NA_MAX_BLOCKS 1024u                                                      /* grid bound of the no-alloc launches */
#define PM_NA_LIST_CAP (PM_NOALLOC_MAX_RAYS + PM_NA_MAX_BLOCKS * (TRACE_BLOCK / 64) * PM_DEFER_CHUNK)   /* entries */
#define PM_NA_OFF_COUNTERS 1024u                                                    /* 8 x u64 */
#define PM_NA_OFF_DEFER 4096u                                                       /* uint4[PM_NA_LIST_CAP] */
#define PM_NA_OFF_SLOW (PM_NA_OFF_DEFER + (size_t)PM_NA_LIST_CAP * 16u)             /* uint32_t[PM_NA_LIST_CAP] */
#define PM_NA_SCRATCH_BYTES (PM_NA_OFF_SLOW + (size_t)PM_NA_LIST_CAP * 4u)
static __device__ __align__(256) unsigned char scratch[PM_NA_SCRATCH_BYTES];
#define PM_MAX_DEVICES 64
static unsigned char* g_scratch_ptr[PM_MAX_DEVICES];     // per device: each GPU has its own copy at its own address
static std::mutex g_scratch_mutex;                       // contexts may be created from several threads at once
static_assert(3 * PM_WORK_SLOTS * sizeof(uint32_t) <= PM_NA_OFF_COUNTERS, "work counters overlap the error counters");

// ------------------------------------------------------------------------------------------ kernels
template <bool ANY, bool COUNT, int KIND>
__global__ __launch_bounds__(TRACE_BLOCK) PM_TRAV_ATTR(ANY, KIND) void trace_kernel(const PmWideNode* __restrict__ nodes,
                                                            const uint4* __restrict__ prims,
                                                            const PmRay* __restrict__ rays, uint32_t n,
                                                            PmHit* __restrict__ hits, uint8_t* __restrict__ occ,
                                                            uint32_t* __restrict__ work, uint32_t refill,
                                                            uint4* __restrict__ defer_list, uint32_t defer_cap,
                                                            unsigned long long* __restrict__ counters) {
    __shared__ uint2 stack[PM_LDS_STACK * TRACE_BLOCK];
    __shared__ uint32_t cand[PM_CAND_WORDS(ANY, COUNT, KIND) * TRACE_BLOCK];     // units, then t lower bounds, then the ring of parked leaves
    pmd::TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    auto fetch = [&](uint32_t i, pmd::Ray* ray) -> bool {
        const float4* rp = reinterpret_cast<const float4*>(rays + i);
        const float4 a = rp[0], b = rp[1];
        ray->o = pmd::mk3(a.x, a.y, a.z); ray->tmax = a.w; ray->d = pmd::mk3(b.x, b.y, b.z);
        return true;
    };
    auto commit = [&](uint32_t i, bool found, const pmd::HitRec& h) {
        if (ANY) occ[i] = found ? 1 : 0;
        else {
            float4 o;
            o.x = __uint_as_float(h.prim_id); o.y = h.t; o.z = h.u; o.w = h.v;
            *reinterpret_cast<float4*>(hits + i) = o;
        }
    };
    pmd::trace_persistent<ANY, COUNT, KIND>(nodes, prims, n, work, refill, stack + threadIdx.x, cand + threadIdx.x, TRACE_BLOCK,
                                      defer_list, work + PM_WORK_SLOTS, defer_cap, &tc, fetch, commit);
#ifdef PM_DBG_UTIL
    if (true) {
#else
    if (COUNT) {
#endif
        // wave-level reduction, one atomic per wave per counter
        unsigned long long nn = tc.nodes, pp = tc.prims, ee = tc.exact, ff = tc.fallback;
        for (int off = 32; off > 0; off >>= 1) {
            nn += __shfl_down(nn, off); pp += __shfl_down(pp, off); ee += __shfl_down(ee, off); ff += __shfl_down(ff, off);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters[1], nn); atomicAdd(&counters[2], pp); atomicAdd(&counters[6], ee);
            if (ff) atomicAdd(&counters[7], ff);
        }
    }
    if (tc.overflow) atomicAdd(&counters[3], 1ull);
}

// the same entry points for scenes small enough to be swept without a tree (pm_device.hpp trace_flat)
template <bool ANY, bool COUNT, bool PATCH>
__global__ __launch_bounds__(TRACE_BLOCK) PM_FLAT_ATTR void trace_flat_kernel(PmFlatScene fs, const uint4* __restrict__ prims,
                                                                 const PmRay* __restrict__ rays, uint32_t n,
                                                                 PmHit* __restrict__ hits, uint8_t* __restrict__ occ,
                                                                 uint4* __restrict__ defer_list, uint32_t defer_cap, uint32_t* __restrict__ defer_count,
                                                                 uint32_t* __restrict__ slow_list,
                                                                 unsigned long long* __restrict__ counters) {
    __shared__ uint32_t cand[3 * PM_FLAT_CAND_MAX * TRACE_BLOCK];     // units, then the lower bounds of t as fractions (numerators, denominators)
    pmd::TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    auto fetch = [&](uint32_t i, pmd::Ray* ray) -> bool {
        const float4* rp = reinterpret_cast<const float4*>(rays + i);
        const float4 a = rp[0], b = rp[1];
        ray->o = pmd::mk3(a.x, a.y, a.z); ray->tmax = a.w; ray->d = pmd::mk3(b.x, b.y, b.z);
        return true;
    };
    auto commit = [&](uint32_t i, bool found, const pmd::HitRec& h) {
        if (ANY) occ[i] = found ? 1 : 0;
        else {
            float4 o;
            o.x = __uint_as_float(h.prim_id); o.y = h.t; o.z = h.u; o.w = h.v;
            *reinterpret_cast<float4*>(hits + i) = o;
        }
    };
    pmd::trace_flat<ANY, COUNT, PATCH>(fs, prims, n, cand + threadIdx.x, TRACE_BLOCK, defer_list, defer_count, slow_list,
                                defer_count + PM_WORK_SLOTS, defer_cap, &tc, fetch, commit);
    if (COUNT) {
        unsigned long long pp = tc.prims, ee = tc.exact, ff = tc.fallback;
        for (int off = 32; off > 0; off >>= 1) { pp += __shfl_down(pp, off); ee += __shfl_down(ee, off); ff += __shfl_down(ff, off); }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters[2], pp); atomicAdd(&counters[6], ee);
            if (ff) atomicAdd(&counters[7], ff);
        }
    }
    if (tc.overflow) atomicAdd(&counters[3], 1ull);
}

// second pass (pm_device.hpp deferred_fix): exact tests of what trace_kernel / trace_flat_kernel deferred, dense waves
template <bool ANY, bool COUNT>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_fix_kernel(PmFlatScene fs, const uint4* __restrict__ prims, const PmRay* __restrict__ rays,
                                                                PmHit* __restrict__ hits, uint8_t* __restrict__ occ,
                                                                const uint4* __restrict__ defer_list, uint32_t defer_cap,
                                                                const uint32_t* __restrict__ defer_count,
                                                                const uint32_t* __restrict__ slow_list,
                                                                unsigned long long* __restrict__ counters) {
    pmd::TravCounters tc = {0u, 0u, 0u, 0u, 0u};
    auto fetch = [&](uint32_t i, pmd::Ray* ray) -> bool {
        const float4* rp = reinterpret_cast<const float4*>(rays + i);
        const float4 a = rp[0], b = rp[1];
        ray->o = pmd::mk3(a.x, a.y, a.z); ray->tmax = a.w; ray->d = pmd::mk3(b.x, b.y, b.z);
        return true;
    };
    auto load_t = [&](uint32_t i) -> float { return hits[i].t; };
    auto commit = [&](uint32_t i, bool found, const pmd::HitRec& h) {
        if (ANY) occ[i] = found ? 1 : 0;
        else {
            float4 o;
            o.x = __uint_as_float(h.prim_id); o.y = h.t; o.z = h.u; o.w = h.v;
            *reinterpret_cast<float4*>(hits + i) = o;
        }
    };
    __shared__ uint4 gather[128 * (TRACE_BLOCK / 64)];
    pmd::deferred_fix<ANY, COUNT>(fs, prims, defer_list, defer_count, slow_list, defer_count + PM_WORK_SLOTS, defer_cap,
                                  gather + 128 * (threadIdx.x >> 6), &tc, fetch, load_t, commit);
    if (COUNT && tc.exact) atomicAdd(&counters[6], (unsigned long long)tc.exact);
    if (COUNT && tc.prims) atomicAdd(&counters[2], (unsigned long long)tc.prims);
}

int pm_gpu_build_tris(pm_ctx* c, int max_leaf, std::string& err);   // pm_gpu_build.hip

// ------------------------------------------------------------------------------------------ lifetime
extern "C" uint32_t pm_abi_version(void) { return PM_ABI_VERSION; }
extern "C" int pm_create(int device_id, pm_ctx** out) {
    if (!out) return PM_ERR_INVALID;
    *out = nullptr;
    if (device_id == -1) {   // host-only context: geometry + pm_build_accel + buffer export, no HIP call at all
        pm_ctx* hc = new pm_ctx();
        hc->device = -1;
        *out = hc;
        return PM_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return PM_ERR_HIP;
    pm_ctx* c = new pm_ctx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc(&c->d_counters, 8 * sizeof(unsigned long long)) != hipSuccess ||
        pm_memset_sync(c, c->d_counters, 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&c->d_work, 3 * PM_WORK_SLOTS * sizeof(uint32_t)) != hipSuccess ||
        hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess) {
        delete c;
        return PM_ERR_HIP;
    }
    // device address of this GPU's copy of `scratch` (hipSetDevice(device_id) above made it current)
    if (device_id < PM_MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device_id]) {
            unsigned char* ptr = nullptr;
            if (hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) { pm_destroy(c); return PM_ERR_HIP; }
            g_scratch_ptr[device_id] = ptr;
        }
        c->scratch = g_scratch_ptr[device_id];
    }
    *out = c;
    return PM_OK;
}
extern "C" void pm_destroy(pm_ctx* c) {
    if (!c) return;
    if (c->device < 0) { pm_render_state_destroy(c); delete c; return; }      // (host-only: the render state holds host tables only)
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    pm_render_state_destroy(c);
    if (c->d_nodes) (void)hipFree(c->d_nodes);
    if (c->d_prims) (void)hipFree(c->d_prims);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_work) (void)hipFree(c->d_work);
    if (c->d_defer) (void)hipFree(c->d_defer);
    if (c->d_slow) (void)hipFree(c->d_slow);
    if (c->flat.tris) (void)hipFree((void*)c->flat.tris);
    if (c->flat.units) (void)hipFree((void*)c->flat.units);
    if (c->d_stage_in) (void)hipFree(c->d_stage_in);
    if (c->d_stage_out) (void)hipFree(c->d_stage_out);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
extern "C" const char* pm_last_error(pm_ctx* c) { return c ? c->err.c_str() : "null context"; }

// ------------------------------------------------------------------------------------------ geometry
static pm::Box box_of_points(const float (*p)[3], int n) {
    pm::Box b;
    for (int k = 0; k < 3; ++k) { b.lo[k] = p[0][k]; b.hi[k] = p[0][k]; }
    for (int i = 1; i < n; ++i)
        for (int k = 0; k < 3; ++k) { b.lo[k] = std::fmin(b.lo[k], p[i][k]); b.hi[k] = std::fmax(b.hi[k], p[i][k]); }
    return b;
}
static float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

extern "C" int pm_add_spheres(pm_ctx* c, const PmSphere* s, uint32_t n, uint32_t* first, uint32_t struct_bytes) {
    if (!c || (!s && n)) return fail(c, PM_ERR_INVALID, "pm_add_spheres: null argument");
    PM_ABI_CHECK(c, PmSphere, struct_bytes, "pm_add_spheres");
    for (uint32_t i = 0; i < n; ++i) {
        const float* ofr = s[i].object_from_render;
        const float* rfo = s[i].render_from_object;
        if (!(ofr[12] == 0.f && ofr[13] == 0.f && ofr[14] == 0.f && ofr[15] == 1.f && rfo[12] == 0.f && rfo[13] == 0.f &&
              rfo[14] == 0.f && rfo[15] == 1.f))
            return fail(c, PM_ERR_UNSUPPORTED, "pm_add_spheres: only affine transforms (last row 0 0 0 1)");
        if (!(s[i].radius > 0.f) || s[i].z_min > s[i].z_max)
            return fail(c, PM_ERR_INVALID, "pm_add_spheres: radius must be > 0 and z_min <= z_max (sphere.rs:42-47)");
    }
    if (first) *first = c->scene.next_prim_id;
    for (uint32_t i = 0; i < n; ++i) {
        pm::HostSphere h;
        // Sphere::new, shapes/sphere.rs:34-68
        h.radius = s[i].radius;
        h.z_min = clampf(s[i].z_min, -h.radius, h.radius);
        h.z_max = clampf(s[i].z_max, -h.radius, h.radius);
        h.phi_max = clampf(s[i].phi_max_deg, 0.0f, 360.0f) * (3.14159265358979323846f / 180.0f);
        h.theta_z_min = std::acos(clampf(h.z_min / h.radius, -1.0f, 1.0f));
        h.theta_z_max = std::acos(clampf(h.z_max / h.radius, -1.0f, 1.0f));
        std::memcpy(h.render_from_object, s[i].render_from_object, 64);
        std::memcpy(h.object_from_render, s[i].object_from_render, 64);
        h.reverse_orientation = s[i].reverse_orientation != 0;
        h.prim_id = c->scene.next_prim_id++;
        // Sphere::bounds (sphere.rs:72-78) through Transform * Bounds3f (core/transform.rs:532-566)
        const float* m = h.render_from_object;
        const float blo[3] = {-h.radius, -h.radius, h.z_min}, bhi[3] = {h.radius, h.radius, h.z_max};
        for (int r = 0; r < 3; ++r) {
            float lo = m[r * 4 + 3], hi = m[r * 4 + 3];
            for (int k = 0; k < 3; ++k) {
                float a = m[r * 4 + k] * blo[k], b = m[r * 4 + k] * bhi[k];
                lo += std::fmin(a, b); hi += std::fmax(a, b);
            }
            h.bounds.lo[r] = lo; h.bounds.hi[r] = hi;
        }
        c->scene.spheres.push_back(h);
    }
    c->accel_valid = false; c->geom_gen++;
    return PM_OK;
}

extern "C" int pm_add_patch_mesh(pm_ctx* c, const float* pos, uint32_t n_vert, const uint32_t* idx4, uint32_t n_patch,
                                 const float* normals, int reverse_orientation, int swaps, uint32_t* first) {
    return pm_add_patch_mesh_uv(c, pos, n_vert, idx4, n_patch, normals, nullptr, reverse_orientation, swaps, first);
}
extern "C" int pm_add_patch_mesh_uv(pm_ctx* c, const float* pos, uint32_t n_vert, const uint32_t* idx4, uint32_t n_patch,
                                    const float* normals, const float* uv, int reverse_orientation, int swaps, uint32_t* first) {
    if (!c || !pos || (!idx4 && n_patch)) return fail(c, PM_ERR_INVALID, "pm_add_patch_mesh: null argument");
    for (uint64_t i = 0; i < (uint64_t)n_patch * 4; ++i)
        if (idx4[i] >= n_vert) return fail(c, PM_ERR_INVALID, "pm_add_patch_mesh: vertex index out of range");
    if (first) *first = c->scene.next_prim_id;
    c->scene.patches.reserve(c->scene.patches.size() + n_patch);
    for (uint32_t i = 0; i < n_patch; ++i) {
        pm::HostPatch h;
        for (int k = 0; k < 4; ++k) std::memcpy(h.p[k], pos + 3 * (size_t)idx4[4 * i + k], 12);
        h.has_normals = normals != nullptr;
        if (normals) for (int k = 0; k < 4; ++k) std::memcpy(h.n[k], normals + 3 * (size_t)idx4[4 * i + k], 12);
        h.has_uv = uv != nullptr;
        if (uv) for (int k = 0; k < 4; ++k) std::memcpy(h.uv[k], uv + 2 * (size_t)idx4[4 * i + k], 8);
        h.flip_normal = (reverse_orientation != 0) != (swaps != 0);
        h.prim_id = c->scene.next_prim_id++;
        h.bounds = box_of_points(h.p, 4);   // bilinear_patch.rs:176-181
        c->scene.patches.push_back(h);
    }
    c->accel_valid = false; c->geom_gen++;
    return PM_OK;
}

extern "C" int pm_add_triangle_mesh(pm_ctx* c, const float* pos, uint32_t n_vert, const uint32_t* idx3, uint32_t n_tri,
                                    uint32_t* first) {
    if (!c || !pos || (!idx3 && n_tri)) return fail(c, PM_ERR_INVALID, "pm_add_triangle_mesh: null argument");
    for (uint64_t i = 0; i < (uint64_t)n_tri * 3; ++i)
        if (idx3[i] >= n_vert) return fail(c, PM_ERR_INVALID, "pm_add_triangle_mesh: vertex index out of range");
    if (first) *first = c->scene.next_prim_id;
    c->scene.tris.reserve(c->scene.tris.size() + n_tri);
    for (uint32_t i = 0; i < n_tri; ++i) {
        pm::HostTri h;
        for (int k = 0; k < 3; ++k) std::memcpy(h.v[k], pos + 3 * (size_t)idx3[3 * i + k], 12);
        h.prim_id = c->scene.next_prim_id++;
        h.bounds = box_of_points(h.v, 3);
        c->scene.tris.push_back(h);
    }
    c->accel_valid = false; c->geom_gen++;
    return PM_OK;
}

extern "C" int pm_clear_geometry(pm_ctx* c) {
    if (!c) return PM_ERR_INVALID;
    c->scene = pm::HostScene();
    c->accel_valid = false; c->geom_gen++;
    return PM_OK;
}

extern "C" int pm_build_accel(pm_ctx* c, int split_method, int max_prims_in_node, PmAccelInfo* info, uint32_t struct_bytes) {
    if (!c) return PM_ERR_INVALID;
    if (info) PM_ABI_CHECK(c, PmAccelInfo, struct_bytes, "pm_build_accel");
    if (split_method < PM_SPLIT_HLBVH || split_method > PM_SPLIT_EQUAL_COUNTS)
        return fail(c, PM_ERR_INVALID, "pm_build_accel: unknown split method");
    if (c->device >= 0) HIPCHK(c, hipSetDevice(c->device));
    c->split_method = split_method;
    std::string err;
    // PM_SPLIT_HLBVH on a GPU context: linear BVH built on the device (pm_gpu_build.hip) for triangle scenes
    bool built_on_gpu = false;
    if (split_method == PM_SPLIT_HLBVH && c->device >= 0) {
        const int rc = pm_gpu_build_tris(c, max_prims_in_node, err);
        if (rc == PM_OK) built_on_gpu = true;
        else if (rc != PM_ERR_UNSUPPORTED) return fail(c, rc, "pm_build_accel: " + err);
    }
    // the reference's max_prims_in_node is u8 (main.rs passes 255); wide leaves hold at most PM_MAX_LEAF
    if (!built_on_gpu) {
        c->accel.dev_nodes = 0; c->accel.dev_prim_bytes = 0;
        if (pm::build_accel(c->scene, max_prims_in_node, c->accel, err) != 0) return fail(c, PM_ERR_INVALID, "pm_build_accel: " + err);
    }
    const size_t n_nodes = built_on_gpu ? c->accel.dev_nodes : c->accel.nodes.size();
    const size_t nb = n_nodes * sizeof(PmWideNode), pb = built_on_gpu ? c->accel.dev_prim_bytes : c->accel.prims.size();
    if (c->device >= 0 && !built_on_gpu) {
        if (c->d_nodes) { HIPCHK(c, hipFree(c->d_nodes)); c->d_nodes = nullptr; }
        if (c->d_prims) { HIPCHK(c, hipFree(c->d_prims)); c->d_prims = nullptr; }
        HIPCHK(c, hipMalloc((void**)&c->d_nodes, nb));
        HIPCHK(c, hipMalloc((void**)&c->d_prims, pb ? pb : 16));
        HIPCHK(c, hipMemcpy(c->d_nodes, c->accel.nodes.data(), nb, hipMemcpyHostToDevice));
        if (pb) HIPCHK(c, hipMemcpy(c->d_prims, c->accel.prims.data(), pb, hipMemcpyHostToDevice));
    }
    // tiny scenes are swept flat (pm_device.hpp trace_flat); PBRT_MI355_FLAT_MAX=0 keeps the tree walk for them
    if (c->flat.tris) { (void)hipFree((void*)c->flat.tris); }
    if (c->flat.units) { (void)hipFree((void*)c->flat.units); }
    c->flat = PmFlatScene{nullptr, nullptr, 0, 0, 0};
    uint32_t flat_max = c->flat_max;
    if (const char* e = std::getenv("PBRT_MI355_FLAT_MAX")) { int v = std::atoi(e); if (v >= 0 && v <= PM_FLAT_MAX_PRIMS) flat_max = (uint32_t)v; }
    pm::build_flat_table(c->accel, flat_max);
    if (c->device >= 0 && c->accel.flat_n[0] + c->accel.flat_n[1] + c->accel.flat_n[2] > 0) {
        const size_t tb = std::max<size_t>(1, c->accel.flat_tris.size()) * sizeof(PmFlatTri);
        const size_t ub = std::max<size_t>(1, c->accel.flat_units.size()) * sizeof(uint32_t);
        PmFlatTri* dt = nullptr; uint32_t* du = nullptr;
        HIPCHK(c, hipMalloc((void**)&dt, tb));
        HIPCHK(c, hipMalloc((void**)&du, ub));
        if (!c->accel.flat_tris.empty()) HIPCHK(c, hipMemcpy(dt, c->accel.flat_tris.data(), c->accel.flat_tris.size() * sizeof(PmFlatTri), hipMemcpyHostToDevice));
        if (!c->accel.flat_units.empty()) HIPCHK(c, hipMemcpy(du, c->accel.flat_units.data(), c->accel.flat_units.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->flat = PmFlatScene{dt, du, c->accel.flat_n[0], c->accel.flat_n[1], c->accel.flat_n[2]};
    }
    c->accel_valid = true; c->geom_gen++;
    // refill threshold of the persistent traversal waves: walks through a tiny tree all have about the same (short)
    // length, so refilling early only multiplies the exact-test rounds; deep trees have long-tailed walks.
    c->refill = n_nodes <= 64 ? 64u : 24u;     // measured: scripts/refill_sweep.sh, profiles/r01/refill_sweep*.log
    if (const char* e = std::getenv("PBRT_MI355_REFILL")) { int v = std::atoi(e); if (v >= 1 && v <= 64) c->refill = (uint32_t)v; }
    if (info) {
        info->n_prims = c->scene.next_prim_id;
        info->n_nodes = (uint32_t)n_nodes;
        info->node_bytes = sizeof(PmWideNode);
        info->prim_record_bytes[0] = sizeof(PmSphereRecord);
        info->prim_record_bytes[1] = sizeof(PmPatchRecord);
        info->prim_record_bytes[2] = sizeof(PmTriRecord);
        info->node_buffer_bytes = nb; info->prim_buffer_bytes = pb;
        for (int k = 0; k < 3; ++k) { info->bounds[k] = c->accel.bounds.lo[k]; info->bounds[3 + k] = c->accel.bounds.hi[k]; }
        info->build_seconds = c->accel.build_seconds;
        info->flat_prims = c->accel.flat_n[0] + c->accel.flat_n[1] + c->accel.flat_n[2];
        info->reserved = 0;
    }
    return PM_OK;
}

// ------------------------------------------------------------------------------------------ trace
// the kernels of one batch of at most 2^30 rays that are resident in HBM: walk (or flat sweep) + second pass.  `work` holds
// 3 * PM_WORK_SLOTS zeroed counters (work items, deferred entries, slow entries), `slot` picks this launch's; max_blocks
// bounds the grids (the lists must hold items + waves * PM_DEFER_CHUNK entries, pm_device.hpp DeferWriter).
template <bool ANY>
static void launch_trace(pm_ctx* c, const PmRay* d_rays, uint32_t m, PmHit* oh, uint8_t* oo, uint32_t* work, uint32_t slot,
                         uint4* defer, uint32_t defer_cap, uint32_t* slow, unsigned long long* counters, bool counting,
                         uint32_t max_blocks) {
    const bool has_spheres = !c->scene.spheres.empty();
    const uint32_t blocks = (m + TRACE_BLOCK - 1) / TRACE_BLOCK;
    const uint32_t grid = std::min(std::min<uint32_t>(blocks, (uint32_t)c->n_cu * PM_BLOCKS_PER_CU), max_blocks);
    if (c->flat.tris) {
        const uint32_t fgrid = std::min(std::min<uint32_t>(blocks, (uint32_t)c->n_cu * PM_FLAT_BLOCKS_PER_CU), max_blocks);
        const bool patch = c->flat.n_patch > 0;
        auto kernel = counting ? (patch ? trace_flat_kernel<ANY, true, true> : trace_flat_kernel<ANY, true, false>)
                               : (patch ? trace_flat_kernel<ANY, false, true> : trace_flat_kernel<ANY, false, false>);
        hipLaunchKernelGGL(kernel, dim3(fgrid), dim3(TRACE_BLOCK), 0, c->stream, c->flat, c->d_prims, d_rays, m, oh, oo,
                           defer, defer_cap, work + PM_WORK_SLOTS + slot, slow, counters);
    } else {
        const int kind = pm_walk_kind(c);
        auto kernel = counting ? (kind == PM_WALK_MIXED ? trace_kernel<ANY, true, PM_WALK_MIXED> : kind == PM_WALK_PATCH ? trace_kernel<ANY, true, PM_WALK_PATCH> : trace_kernel<ANY, true, PM_WALK_TRI>)
                               : (kind == PM_WALK_MIXED ? trace_kernel<ANY, false, PM_WALK_MIXED> : kind == PM_WALK_PATCH ? trace_kernel<ANY, false, PM_WALK_PATCH> : trace_kernel<ANY, false, PM_WALK_TRI>);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(TRACE_BLOCK), 0, c->stream, c->d_nodes, c->d_prims, d_rays, m, oh, oo,
                           work + slot, c->refill, defer, defer_cap, counters);
    }
    if (has_spheres || c->flat.tris) {      // the flat sweep defers patches and rare cases too
        if (counting)
            hipLaunchKernelGGL((trace_fix_kernel<ANY, true>), dim3(grid), dim3(TRACE_BLOCK), 0, c->stream, c->flat, c->d_prims,
                               d_rays, oh, oo, defer, defer_cap, work + PM_WORK_SLOTS + slot, c->flat.tris ? slow : nullptr, counters);
        else
            hipLaunchKernelGGL((trace_fix_kernel<ANY, false>), dim3(grid), dim3(TRACE_BLOCK), 0, c->stream, c->flat, c->d_prims,
                               d_rays, oh, oo, defer, defer_cap, work + PM_WORK_SLOTS + slot, c->flat.tris ? slow : nullptr, counters);
    }
}

template <bool ANY>
static int trace_impl(pm_ctx* c, const PmRay* rays, uint64_t n, void* out, size_t out_elem) {
    if (!c) return PM_ERR_INVALID;
    if (c->device < 0) return fail(c, PM_ERR_INVALID, "trace: host-only context (pm_create(-1)) cannot launch kernels");
    if (!c->accel_valid) return fail(c, PM_ERR_NO_ACCEL, "trace: pm_build_accel has not been called");
    if (n == 0) { c->last_ms = 0.f; return PM_OK; }
    if (!rays || !out) return fail(c, PM_ERR_INVALID, "trace: null rays/out");
    HIPCHK(c, hipSetDevice(c->device));
    const bool in_dev = is_device_ptr(rays), out_dev = is_device_ptr(out);
    const PmRay* d_rays = rays;
    void* d_out = out;
    if (!in_dev) {
        int r = ensure(c, &c->d_stage_in, &c->stage_in_cap, n * sizeof(PmRay));
        if (r) return r;
        HIPCHK(c, hipMemcpyAsync(c->d_stage_in, rays, n * sizeof(PmRay), hipMemcpyHostToDevice, c->stream));
        d_rays = (const PmRay*)c->d_stage_in;
    }
    if (!out_dev) {
        int r = ensure(c, &c->d_stage_out, &c->stage_out_cap, n * out_elem);
        if (r) return r;
        d_out = c->d_stage_out;
    }
    unsigned long long before[4];
    HIPCHK(c, hipMemcpyAsync(before, c->d_counters, sizeof(before), hipMemcpyDeviceToHost, c->stream));
    // persistent waves pull ray indices from a work counter; one launch covers at most 2^30 rays (2^24 when the scene
    // holds spheres: the deferred-sphere list has one 16-byte entry per ray of the launch)
    const bool has_spheres = !c->scene.spheres.empty();
    const uint64_t chunk = (has_spheres || c->flat.tris) ? (1ull << 24) : (1ull << 30);
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    { int r = ensure_defer(c, (size_t)std::min<uint64_t>(chunk, n)); if (r) return r; }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    for (uint64_t k = 0; k < n_chunks; ++k) {
        const uint32_t slot = (uint32_t)(k % PM_WORK_SLOTS);     // stream-ordered: a slot is free again when it comes round
        if (slot == 0) HIPCHK(c, hipMemsetAsync(c->d_work, 0, 3 * PM_WORK_SLOTS * sizeof(uint32_t), c->stream));
        const uint64_t b = k * chunk;
        const uint32_t m = (uint32_t)std::min<uint64_t>(chunk, n - b);
        PmHit* oh = ANY ? nullptr : (PmHit*)d_out + b;
        uint8_t* oo = ANY ? (uint8_t*)d_out + b : nullptr;
        launch_trace<ANY>(c, d_rays + b, m, oh, oo, c->d_work, slot, c->d_defer, (uint32_t)c->defer_cap, c->d_slow, c->d_counters,
                          c->counting, 0xFFFFFFFFu);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    if (!out_dev) HIPCHK(c, hipMemcpyAsync(out, d_out, n * out_elem, hipMemcpyDeviceToHost, c->stream));
    unsigned long long after[4];
    HIPCHK(c, hipMemcpyAsync(after, c->d_counters, sizeof(after), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    if (c->counting) c->rays_counted += n;
    if (after[3] != before[3]) return fail(c, PM_ERR_STACK_OVERFLOW, "trace: traversal stack / deferred list overflow");
    return PM_OK;
}
extern "C" int pm_trace_closest(pm_ctx* c, const PmRay* rays, uint64_t n, PmHit* out) {
    return trace_impl<false>(c, rays, n, out, sizeof(PmHit));
}
extern "C" int pm_trace_any(pm_ctx* c, const PmRay* rays, uint64_t n, uint8_t* out) {
    return trace_impl<true>(c, rays, n, out, 1);
}

// pm_trace_closest for callers that cannot allocate: same kernels, same results, no hipMalloc / hipFree (nor any allocating
// library routine) - everything the kernels need beyond rays and hits comes from this GPU's `scratch`.  It is NOT capturable
// into a HIP graph and it is a blocking call: like pm_trace_closest it launches on the context's own stream, waits for it,
// reads the overflow word back and times the launch with events.  Both buffers must be
// device memory (a host buffer would need staging, i.e. an allocation).  Calls on one GPU share that GPU's scratch, which
// keeps the last call's contents: the first 4 KB (work counters, error counters) are cleared here, the lists are rewritten
// up to the counts the kernels themselves produce.  Calls that overlap on one GPU (two contexts, two threads) race.
extern "C" int pm_trace_closest_noalloc(pm_ctx* c, const PmRay* rays, uint64_t n, PmHit* out) {
    if (!c) return PM_ERR_INVALID;
    if (c->device < 0) return fail(c, PM_ERR_INVALID, "pm_trace_closest_noalloc: host-only context (pm_create(-1)) cannot launch kernels");
    if (!c->accel_valid) return fail(c, PM_ERR_NO_ACCEL, "pm_trace_closest_noalloc: pm_build_accel has not been called");
    if (n > PM_NOALLOC_MAX_RAYS)
        return fail(c, PM_ERR_INVALID, "pm_trace_closest_noalloc: " + std::to_string(n) + " rays exceed PM_NOALLOC_MAX_RAYS (" +
                                       std::to_string(PM_NOALLOC_MAX_RAYS) + "); split the batch or use pm_trace_closest");
    if (n == 0) { c->last_ms = 0.f; return PM_OK; }
    if (!rays || !out) return fail(c, PM_ERR_INVALID, "pm_trace_closest_noalloc: null rays/out");
    if (!c->scratch) return fail(c, PM_ERR_UNSUPPORTED, "pm_trace_closest_noalloc: no scratch on this device");
    HIPCHK(c, hipSetDevice(c->device));
    if (!is_device_ptr(rays) || !is_device_ptr(out))
        return fail(c, PM_ERR_INVALID, "pm_trace_closest_noalloc: rays and out must be device memory (host buffers need staging)");
    unsigned char* sp = c->scratch;                                 // the looked-up device address, never the symbol's name
    uint32_t* work = reinterpret_cast<uint32_t*>(sp);
    unsigned long long* counters = reinterpret_cast<unsigned long long*>(sp + PM_NA_OFF_COUNTERS);
    uint4* defer = reinterpret_cast<uint4*>(sp + PM_NA_OFF_DEFER);
    uint32_t* slow = reinterpret_cast<uint32_t*>(sp + PM_NA_OFF_SLOW);
    HIPCHK(c, hipMemsetAsync(sp, 0, PM_NA_OFF_DEFER, c->stream));    // what must start at zero: work / list / error counters
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    launch_trace<false>(c, rays, (uint32_t)n, out, nullptr, work, 0u, defer, (uint32_t)PM_NA_LIST_CAP, slow, counters, false,
                        PM_NA_MAX_BLOCKS);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    unsigned long long overflow = 0;
    HIPCHK(c, hipMemcpyAsync(&overflow, counters + 3, sizeof(overflow), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    if (overflow) return fail(c, PM_ERR_STACK_OVERFLOW, "pm_trace_closest_noalloc: traversal stack / deferred list overflow");
    return PM_OK;
}

// ------------------------------------------------------------------------------------------ measurement
extern "C" int pm_set_counting(pm_ctx* c, int enabled) { if (!c) return PM_ERR_INVALID; c->counting = enabled != 0; return PM_OK; }
extern "C" int pm_reset_counters(pm_ctx* c) {
    if (!c || c->device < 0) return PM_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, pm_memset_sync(c, c->d_counters, 8 * sizeof(unsigned long long)));
    c->rays_counted = 0;
    return PM_OK;
}
extern "C" int pm_get_counters(pm_ctx* c, PmCounters* out, uint32_t struct_bytes) {
    if (!c || !out || c->device < 0) return PM_ERR_INVALID;
    PM_ABI_CHECK(c, PmCounters, struct_bytes, "pm_get_counters");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long v[8];
    HIPCHK(c, hipMemcpy(v, c->d_counters, sizeof(v), hipMemcpyDeviceToHost));
    out->rays = c->rays_counted; out->nodes_visited = v[1] + v[4]; out->prims_tested = v[2] + v[5]; out->stack_overflows = v[3];
    out->exact_tests = v[6]; out->fallback_walks = v[7];
    return PM_OK;
}
extern "C" int pm_last_kernel_ms(pm_ctx* c, float* ms) { if (!c || !ms) return PM_ERR_INVALID; *ms = c->last_ms; return PM_OK; }
// host copies of a tree that was built on the device, fetched when first asked for
static int fetch_device_tree(pm_ctx* c) {
    if (c->accel.dev_nodes == 0 || !c->accel.nodes.empty()) return PM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    c->accel.nodes.resize(c->accel.dev_nodes);
    c->accel.prims.resize(c->accel.dev_prim_bytes);
    HIPCHK(c, hipMemcpy(c->accel.nodes.data(), c->d_nodes, c->accel.dev_nodes * sizeof(PmWideNode), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(c->accel.prims.data(), c->d_prims, c->accel.dev_prim_bytes, hipMemcpyDeviceToHost));
    return PM_OK;
}
extern "C" int pm_get_node_buffer(pm_ctx* c, void* data, uint64_t cap, uint64_t* needed) {
    if (!c || !c->accel_valid) return fail(c, PM_ERR_NO_ACCEL, "no accel");
    if (data) { const int rc = fetch_device_tree(c); if (rc) return rc; }
    const uint64_t nb = (c->accel.dev_nodes ? c->accel.dev_nodes : c->accel.nodes.size()) * sizeof(PmWideNode);
    if (needed) *needed = nb;
    if (data) { if (cap < nb) return fail(c, PM_ERR_INVALID, "buffer too small"); std::memcpy(data, c->accel.nodes.data(), nb); }
    return PM_OK;
}
extern "C" int pm_get_prim_buffer(pm_ctx* c, void* data, uint64_t cap, uint64_t* needed) {
    if (!c || !c->accel_valid) return fail(c, PM_ERR_NO_ACCEL, "no accel");
    if (data) { const int rc = fetch_device_tree(c); if (rc) return rc; }
    const uint64_t nb = c->accel.dev_nodes ? c->accel.dev_prim_bytes : c->accel.prims.size();
    if (needed) *needed = nb;
    if (data) { if (cap < nb) return fail(c, PM_ERR_INVALID, "buffer too small"); std::memcpy(data, c->accel.prims.data(), nb); }
    return PM_OK;
}
