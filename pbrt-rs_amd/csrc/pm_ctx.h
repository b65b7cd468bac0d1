// Internal: context object shared by the translation units of libpbrt_mi355.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pbrt_mi355.h"
#include "pm_build.h"

struct RenderState;   // pm_render.hip

#define TRACE_BLOCK 128
#define PM_WORK_SLOTS 64
#ifndef PM_BLOCKS_PER_CU
#define PM_BLOCKS_PER_CU 10     // persistent traversal grids: 10 blocks x 2 waves per CU = 5 waves per SIMD, all resident
                                // (96 VGPRs, 14 KB of LDS per block: pm_layout.h PM_LDS_STACK)
#endif
#ifndef PM_FLAT_BLOCKS_PER_CU
#define PM_FLAT_BLOCKS_PER_CU 16    // flat-sweep grids (grid-stride over 64-item groups): small kernels, high occupancy
#endif
#ifndef PM_FLAT_WAVES_PER_EU
#define PM_FLAT_WAVES_PER_EU 8      // measured (scripts/ab_path.sh): with the SLP vectoriser off the sweep fits 64 VGPRs, and
                                    // 8 waves/SIMD (= the 16 x 2 waves per CU of the grid) beat 5 by 9 %
#endif
#if PM_FLAT_WAVES_PER_EU > 0
#define PM_FLAT_ATTR __attribute__((amdgpu_waves_per_eu(PM_FLAT_WAVES_PER_EU, PM_FLAT_WAVES_PER_EU)))
#else
#define PM_FLAT_ATTR
#endif
#ifndef PM_WAVES_PER_EU
#define PM_WAVES_PER_EU 5           // tree-walk kernels, closest hit: VGPR budget of 5 waves/SIMD (96)
#endif
#ifndef PM_ANY_WAVES_PER_EU
#define PM_ANY_WAVES_PER_EU 5       // tree-walk kernels, any hit (6 = 80 VGPRs measured slower, 7 spills into the node step)
#endif
#ifndef PM_MIXED_WAVES_PER_EU
#define PM_MIXED_WAVES_PER_EU 4     // ... of scenes that hold spheres or patches as well: their walk carries more code, and 5
                                    // waves (96 VGPRs) spill into it (scripts/mixed_probe.py: 4 beats 5 by 39 %, 3 by 3 %)
#endif
#ifndef PM_PATCH_WAVES_PER_EU
#define PM_PATCH_WAVES_PER_EU 4     // ... of patch-only scenes (KIND == 2: 117 VGPRs with the bilinear test inlined)
#endif
#define PM_TRAV_WAVES(any, mixed) ((mixed) == 2 ? PM_PATCH_WAVES_PER_EU : (mixed) ? PM_MIXED_WAVES_PER_EU : (any) ? PM_ANY_WAVES_PER_EU : PM_WAVES_PER_EU)
#define PM_TRAV_ATTR(any, mixed) __attribute__((amdgpu_waves_per_eu(PM_TRAV_WAVES(any, mixed), PM_TRAV_WAVES(any, mixed))))

struct pm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    pm::HostScene scene;
    pm::Accel accel;
    bool accel_valid = false;
    uint64_t geom_gen = 0;                      // bumped by every geometry mutation and every pm_build_accel: the render state
                                                // re-uploads its per-primitive tables when it has seen an older generation
    int split_method = PM_SPLIT_SAH;
    PmWideNode* d_nodes = nullptr;
    uint4* d_prims = nullptr;
    unsigned long long* d_counters = nullptr;   // [1] nodes [2] prims [3] overflow [4] shadow nodes [5] shadow prims [6] exact [7] fallback
    uint32_t* d_work = nullptr;                 // PM_WORK_SLOTS work-item counters of the persistent traversal kernels, then
                                                // PM_WORK_SLOTS entry counts of the deferred list, then of the slow list
    PmFlatScene flat = {nullptr, nullptr, 0, 0, 0};   // device flat table; n_* all zero = walk the tree
    uint32_t flat_max = PM_FLAT_MAX_PRIMS;      // scenes of at most this many primitives are swept flat (0: never)
    uint32_t* d_slow = nullptr;                 // slow rays of ONE flat-sweep launch (work items; same capacity as d_defer)
    uint4* d_defer = nullptr;                   // deferred (work item, sphere, sphere) entries of ONE traversal launch
    size_t defer_cap = 0;                       // ... in entries; stays null for scenes without spheres
    int n_cu = 256;
    uint32_t refill = 24;                       // idle lanes that trigger a wave refill (chosen in pm_build_accel)
    bool counting = false;
    uint64_t rays_counted = 0;
    float last_ms = 0.0f;
    // staging for host-pointer callers
    void* d_stage_in = nullptr; size_t stage_in_cap = 0;
    void* d_stage_out = nullptr; size_t stage_out_cap = 0;
    unsigned char* scratch = nullptr;   // device address of this GPU's file-scope scratch array (pm_api.hip; not owned)
    RenderState* render = nullptr;   // owned; created lazily by the render entry points
};
void pm_render_state_destroy(pm_ctx* c);
// which walk kernel a scene gets (pm_device.hpp PM_WALK_*): 0 triangles only, 2 bilinear patches only, 1 anything else
static inline int pm_walk_kind(const pm_ctx* c) {
    if (!c->scene.spheres.empty()) return 1;
    if (c->scene.patches.empty()) return 0;
    if (const char* e = std::getenv("PBRT_MI355_PATCH_WALK")) if (std::atoi(e) == 0) return 1;     // A/B: round 2's routing
    return c->scene.tris.empty() ? 2 : 1;
}

static inline int fail(pm_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
// ABI handshake (pbrt_mi355.h): the caller's sizeof(T) must be the library's
#define PM_ABI_CHECK(c, T, bytes, fn)                                                                                          \
    do {                                                                                                                       \
        if ((bytes) != (uint32_t)sizeof(T))                                                                                    \
            return fail(c, PM_ERR_INVALID, std::string(fn) + ": struct size mismatch for " #T " (caller " + std::to_string(bytes) + \
                                               " bytes, library " + std::to_string(sizeof(T)) + ", ABI version " +              \
                                               std::to_string(PM_ABI_VERSION) + "): the caller was built against another pbrt_mi355.h"); \
    } while (0)
#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail(c, PM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Zeroing device memory: on the context's own stream and waited for.  (Round 3: pm_film_clear used hipMemset - which for device
// memory may return before it has run, on the null stream, which does not order against this non-blocking stream - and with
// four ranks sharing one card the clear landed AFTER the film_add that followed it.)
static inline hipError_t pm_memset_sync(pm_ctx* c, void* p, size_t bytes) {
    hipError_t e = hipMemsetAsync(p, 0, bytes, c->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);
}
static inline bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}
static inline int ensure(pm_ctx* c, void** buf, size_t* cap, size_t need);
// room for one second-pass entry per work item of a traversal launch (scenes with spheres, and every flat-swept scene)
static inline int ensure_defer_buf(pm_ctx* c, size_t items, uint4** d_defer, uint32_t** d_slow, size_t* cap) {
    // + what the waves of the largest traversal grid can leave unused of their reservations (pm_device.hpp DeferWriter)
    items += (size_t)c->n_cu * PM_FLAT_BLOCKS_PER_CU * (TRACE_BLOCK / 64) * 256;
    if ((c->scene.spheres.empty() && !c->flat.tris) || *cap >= items) return PM_OK;
    if (*d_defer) { HIPCHK(c, hipFree(*d_defer)); *d_defer = nullptr; *cap = 0; }
    if (*d_slow) { HIPCHK(c, hipFree(*d_slow)); *d_slow = nullptr; }
    HIPCHK(c, hipMalloc((void**)d_defer, items * sizeof(uint4)));
    HIPCHK(c, hipMalloc((void**)d_slow, items * sizeof(uint32_t)));
    *cap = items;
    return PM_OK;
}
static inline int ensure_defer(pm_ctx* c, size_t items) { return ensure_defer_buf(c, items, &c->d_defer, &c->d_slow, &c->defer_cap); }
static inline int ensure(pm_ctx* c, void** buf, size_t* cap, size_t need) {
    if (*cap >= need) return PM_OK;
    if (*buf) HIPCHK(c, hipFree(*buf));
    *buf = nullptr; *cap = 0;
    HIPCHK(c, hipMalloc(buf, need));
    *cap = need;
    return PM_OK;
}

