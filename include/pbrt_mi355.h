/* pbrt_mi355.h — C ABI of libpbrt_mi355.so, the MI355X (gfx950) ray-intersection and
 * wavefront path-tracing core that replaces the CPU hot path of erxma/pbrt-rs.
 *
 * The reference has no FFI layer; the seams this ABI replaces are two Rust traits over
 * closed enums (paths relative to the reference tree):
 *   trait Primitive  { bounds, intersect, intersect_p }      src/primitives/mod.rs:21-27
 *   trait Integrate  { render, intersect, intersect_p, .. }  src/integrators/base.rs:32-45
 * Per-ray virtual calls are the wrong granularity for a GPU, so the boundary is batched:
 * whole ray arrays for intersect/intersect_p, whole image windows for render().
 *
 * Conventions: every function returns an int status (PM_OK == 0) and never unwinds;
 * the caller owns all host buffers; the library owns device memory behind pm_ctx;
 * one pm_ctx per GPU; calls on one ctx are serialised by the caller.  Pointers named
 * rays / out / film may be HOST or DEVICE pointers (detected with
 * hipPointerGetAttributes); device pointers are used in place with no copy.
 * Matrices are row-major float[16], m[r*4+c], as SquareMatrix<4> (src/core/square_matrix.rs).
 */
#ifndef PBRT_MI355_H
#define PBRT_MI355_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_OK 0
#define PM_ERR_INVALID 1      /* bad argument / state */
#define PM_ERR_HIP 2          /* a HIP runtime call failed; see pm_last_error */
#define PM_ERR_NO_ACCEL 3     /* trace/render before pm_build_accel */
#define PM_ERR_UNSUPPORTED 4
#define PM_ERR_STACK_OVERFLOW 5 /* traversal stack exhausted (results invalid) */

#define PM_MISS 0xFFFFFFFFu

typedef struct pm_ctx pm_ctx;

/* core/ray.rs:7-14 (Ray{o,dir,time}; the medium pointer is not carried) + the t_max argument of
 * Primitive::intersect (None == +inf). 32 bytes. */
typedef struct PmRay { float o[3]; float tmax; float d[3]; float time; } PmRay;
/* shapes/base.rs:59-63 ShapeIntersection reduced to what identifies the hit: primitive id (insertion
 * order), t_hit, parametric (u,v).  prim_id == PM_MISS and t == ray.tmax on a miss. 16 bytes. */
typedef struct PmHit { uint32_t prim_id; float t; float u; float v; } PmHit;

/* shapes/sphere.rs:19-68 Sphere::builder arguments. */
typedef struct PmSphere {
    float radius, z_min, z_max, phi_max_deg;
    float render_from_object[16];
    float object_from_render[16];
    uint32_t reverse_orientation;
    uint32_t _pad;
} PmSphere;

/* primitives/aggregates.rs:664-676 BVHSplitMethod.  The GPU tree is always a wide (8-ary) quantised
 * BVH built with binned SAH; the value is accepted for interface parity and recorded only. */
enum { PM_SPLIT_HLBVH = 0, PM_SPLIT_SAH = 1, PM_SPLIT_MIDDLE = 2, PM_SPLIT_EQUAL_COUNTS = 3 };

typedef struct PmAccelInfo {
    uint32_t n_prims, n_nodes;
    uint32_t node_bytes;          /* S_node: bytes fetched per node visit (80) */
    uint32_t prim_record_bytes[3];/* S_prim per record: sphere, patch, triangle */
    uint64_t node_buffer_bytes, prim_buffer_bytes;
    float bounds[6];              /* aggregate bounds(): min xyz, max xyz */
    double build_seconds;
} PmAccelInfo;

/* deterministic traversal counters: totals since pm_reset_counters */
typedef struct PmCounters {
    uint64_t rays, nodes_visited, prims_tested, stack_overflows;
} PmCounters;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* device_id >= 0: a GPU context.  device_id == -1: host-only context (geometry, pm_build_accel and the
 * buffer getters work; every call that would launch a kernel returns PM_ERR_INVALID) — used to check the
 * tree on machines without a GPU. */
int pm_create(int device_id, pm_ctx** out);
void pm_destroy(pm_ctx* ctx);
const char* pm_last_error(pm_ctx* ctx);   /* valid until the next call on ctx */

/* ---- scene geometry: replaces Vec<Arc<PrimitiveEnum>> handed to BVHAggregate::new
 * (primitives/aggregates.rs:25-47).  Primitive ids are assigned in insertion order across calls;
 * *first_prim_id receives the id of the first primitive added by the call. */
int pm_add_spheres(pm_ctx* ctx, const PmSphere* spheres, uint32_t n, uint32_t* first_prim_id);
/* shapes/bilinear_patch.rs:630-685 BilinearPatchMesh::new — positions already in render space,
 * 4 indices per patch in (p00,p10,p01,p11) order. Vertex normals are not supported yet (must be NULL). */
int pm_add_patch_mesh(pm_ctx* ctx, const float* pos_xyz, uint32_t n_vert, const uint32_t* idx4, uint32_t n_patch,
                      const float* normals_or_null, int reverse_orientation, int transform_swaps_handedness,
                      uint32_t* first_prim_id);
/* No triangle shape exists in the reference (shapes/base.rs:8-13). A triangle (v0,v1,v2) is intersected
 * exactly as the reference intersects the degenerate patch p00=v0,p10=v1,p01=v2,p11=v2
 * (shapes/bilinear_patch.rs:506-612). */
int pm_add_triangle_mesh(pm_ctx* ctx, const float* pos_xyz, uint32_t n_vert, const uint32_t* idx3, uint32_t n_tri,
                         uint32_t* first_prim_id);
int pm_clear_geometry(pm_ctx* ctx);

/* BVHAggregate::new (primitives/aggregates.rs:25-47): builds the wide BVH on the host and uploads it. */
int pm_build_accel(pm_ctx* ctx, int split_method, int max_prims_in_node, PmAccelInfo* info_or_null);

/* ---- batched Primitive::intersect / intersect_p --------------------------------------------- */
/* BVHAggregate::intersect (primitives/aggregates.rs:537-600), n rays -> n hits (closest hit). */
int pm_trace_closest(pm_ctx* ctx, const PmRay* rays, uint64_t n, PmHit* out);
/* BVHAggregate::intersect_p (primitives/aggregates.rs:602-655), n rays -> n bytes (1 = occluded). */
int pm_trace_any(pm_ctx* ctx, const PmRay* rays, uint64_t n, uint8_t* out);

/* ---- measurement ---------------------------------------------------------------------------- */
int pm_set_counting(pm_ctx* ctx, int enabled);      /* counting kernels are slower; off by default */
int pm_reset_counters(pm_ctx* ctx);
int pm_get_counters(pm_ctx* ctx, PmCounters* out);
/* device time (HIP events on the library's stream) of the kernels launched by the last trace/render call */
int pm_last_kernel_ms(pm_ctx* ctx, float* ms);
/* copies of the device-layout buffers, so a CPU checker can walk the very same tree
 * (cap in bytes; *needed receives the full size; data may be NULL to query) */
int pm_get_node_buffer(pm_ctx* ctx, void* data, uint64_t cap, uint64_t* needed);
int pm_get_prim_buffer(pm_ctx* ctx, void* data, uint64_t cap, uint64_t* needed);

#ifdef __cplusplus
}
#endif
#endif
