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
 * hipPointerGetAttributes); device pointers are used in place with no copy.  Every call is BLOCKING and works on the context's
 * own (non-blocking) stream: work the caller has queued on other streams for a device buffer it passes in must have finished
 * (synchronise that stream first), and results are complete when the call returns.
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

/* ---- ABI handshake ---------------------------------------------------------------------------
 * Every entry point that reads or writes one of the structs below takes the caller's sizeof() of that struct as
 * `struct_bytes`; a caller compiled against another revision of this header gets PM_ERR_INVALID ("struct size mismatch")
 * instead of the library reading past its array.  (Round 2: a C++ harness built before PmLight / PmMaterial grew ran
 * against the rebuilt library and died of SIGSEGV inside pm_set_lights.)  PM_ABI_VERSION is bumped whenever a struct, an
 * enum value or a signature changes; pm_abi_version() returns the library's. */
#define PM_ABI_VERSION 3u

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

/* primitives/aggregates.rs:664-676 BVHSplitMethod.  The tree is always a wide (8-ary) quantised BVH.  SAH, MIDDLE and
 * EQUAL_COUNTS build it on the host with binned SAH (the value is recorded only: closest-hit results do not depend on
 * the tree).  HLBVH builds a linear BVH ON THE DEVICE (Morton codes, radix sort, radix tree, refit, collapse) when the
 * scene holds only triangles (>= 4096 of them) and the context owns a GPU; otherwise it falls back to the host build. */
enum { PM_SPLIT_HLBVH = 0, PM_SPLIT_SAH = 1, PM_SPLIT_MIDDLE = 2, PM_SPLIT_EQUAL_COUNTS = 3 };

typedef struct PmAccelInfo {
    uint32_t n_prims, n_nodes;
    uint32_t node_bytes;          /* S_node: bytes fetched per node visit (80) */
    uint32_t prim_record_bytes[3];/* S_prim per record: sphere, patch, triangle */
    uint64_t node_buffer_bytes, prim_buffer_bytes;
    float bounds[6];              /* aggregate bounds(): min xyz, max xyz */
    double build_seconds;
    uint32_t flat_prims;          /* != 0: the scene is small enough (<= 64 primitives) that rays sweep all of them in lock
                                   * step instead of walking the tree; then nodes_visited stays 0 and prims_tested counts
                                   * every primitive the sweep offers a ray.  PBRT_MI355_FLAT_MAX=0 in the environment
                                   * disables it (checked by pm_build_accel). */
    uint32_t reserved;
} PmAccelInfo;

/* deterministic traversal counters: totals since pm_reset_counters */
typedef struct PmCounters {
    uint64_t rays, nodes_visited, prims_tested, stack_overflows;
    uint64_t exact_tests;      /* reference-arithmetic primitive tests actually run (prims_tested counts pre-filter tests too) */
    uint64_t fallback_walks;   /* rays re-walked with immediate exact tests because culling was not confirmed (expected 0) */
} PmCounters;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* device_id >= 0: a GPU context.  device_id == -1: host-only context (geometry, pm_build_accel and the
 * buffer getters work; every call that would launch a kernel returns PM_ERR_INVALID) — used to check the
 * tree on machines without a GPU. */
uint32_t pm_abi_version(void);            /* == PM_ABI_VERSION of the header the library was built from */
int pm_create(int device_id, pm_ctx** out);
void pm_destroy(pm_ctx* ctx);
const char* pm_last_error(pm_ctx* ctx);   /* valid until the next call on ctx */

/* ---- scene geometry: replaces Vec<Arc<PrimitiveEnum>> handed to BVHAggregate::new
 * (primitives/aggregates.rs:25-47).  Primitive ids are assigned in insertion order across calls;
 * *first_prim_id receives the id of the first primitive added by the call. */
int pm_add_spheres(pm_ctx* ctx, const PmSphere* spheres, uint32_t n, uint32_t* first_prim_id, uint32_t struct_bytes /* sizeof(PmSphere) */);
/* shapes/bilinear_patch.rs:630-685 BilinearPatchMesh::new — positions already in render space,
 * 4 indices per patch in (p00,p10,p01,p11) order.  normals_or_null: one shading normal per vertex (n_vert x 3), already in
 * render space and already negated for reverse_orientation, as BilinearPatchMesh::new leaves them (bilinear_patch.rs:660-671;
 * the reference asserts normals.len() == indices.len() but indexes them by vertex id - here they are per vertex). */
int pm_add_patch_mesh(pm_ctx* ctx, const float* pos_xyz, uint32_t n_vert, const uint32_t* idx4, uint32_t n_patch,
                      const float* normals_or_null, int reverse_orientation, int transform_swaps_handedness,
                      uint32_t* first_prim_id);
/* ... with per-vertex texture coordinates (BilinearPatchMesh.uv, shapes/bilinear_patch.rs:621-622; n_vert x 2 floats, may be
 * NULL).  The reference stores them but never uses them ("TODO: Skipping this part for now", bilinear_patch.rs:103-112); here
 * a textured material is looked up at st = lerp(lerp(uv00, uv01, v), lerp(uv10, uv11, v), u) (pbrt-v4's formula); without
 * them st = the patch's own (u, v). */
int pm_add_patch_mesh_uv(pm_ctx* ctx, const float* pos_xyz, uint32_t n_vert, const uint32_t* idx4, uint32_t n_patch,
                         const float* normals_or_null, const float* uv_or_null, int reverse_orientation,
                         int transform_swaps_handedness, uint32_t* first_prim_id);
/* No triangle shape exists in the reference (shapes/base.rs:8-13). A triangle (v0,v1,v2) is intersected
 * exactly as the reference intersects the degenerate patch p00=v0,p10=v1,p01=v2,p11=v2
 * (shapes/bilinear_patch.rs:506-612). */
int pm_add_triangle_mesh(pm_ctx* ctx, const float* pos_xyz, uint32_t n_vert, const uint32_t* idx3, uint32_t n_tri,
                         uint32_t* first_prim_id);
int pm_clear_geometry(pm_ctx* ctx);

/* BVHAggregate::new (primitives/aggregates.rs:25-47): builds the wide BVH (on the host, or on the device for
 * PM_SPLIT_HLBVH) and uploads it.  max_prims_in_node is an upper bound, as in the reference (whose SAH may stop earlier):
 * wide leaves hold at most 4 primitives, and triangle leaves always hold one (measured faster on 64-lane waves). */
int pm_build_accel(pm_ctx* ctx, int split_method, int max_prims_in_node, PmAccelInfo* info_or_null, uint32_t struct_bytes /* sizeof(PmAccelInfo) */);

/* ---- batched Primitive::intersect / intersect_p --------------------------------------------- */
/* BVHAggregate::intersect (primitives/aggregates.rs:537-600), n rays -> n hits (closest hit). */
int pm_trace_closest(pm_ctx* ctx, const PmRay* rays, uint64_t n, PmHit* out);
/* BVHAggregate::intersect_p (primitives/aggregates.rs:602-655), n rays -> n bytes (1 = occluded). */
int pm_trace_any(pm_ctx* ctx, const PmRay* rays, uint64_t n, uint8_t* out);

/* NO COUNTERPART IN THE REFERENCE.  pm_trace_closest for callers that cannot allocate device memory: harnesses that forbid
 * hipMalloc once set-up is over, processes on a device whose memory is spoken for.  Same kernels and bit-identical results,
 * but it calls no hipMalloc / hipFree or allocating library routine: the work counters and the deferred-primitive / slow-ray
 * lists of the kernels live in a file-scope device array of the library (15.8 MB per GPU, held until the process ends).
 * It is a BLOCKING call like pm_trace_closest (own stream, stream synchronise, overflow word read back, event timing), so it
 * cannot be captured into a HIP graph.  Limits: at most
 * PM_NOALLOC_MAX_RAYS rays per call (more -> PM_ERR_INVALID); rays and out must be DEVICE pointers (else PM_ERR_INVALID).
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (from two contexts or threads) race. */
#define PM_NOALLOC_MAX_RAYS (1u << 18)
int pm_trace_closest_noalloc(pm_ctx* ctx, const PmRay* d_rays, uint64_t n, PmHit* d_out);

/* ---- measurement ---------------------------------------------------------------------------- */
int pm_set_counting(pm_ctx* ctx, int enabled);      /* counting kernels are slower; off by default */
int pm_reset_counters(pm_ctx* ctx);
int pm_get_counters(pm_ctx* ctx, PmCounters* out, uint32_t struct_bytes /* sizeof(PmCounters) */);
/* device time (HIP events on the library's stream) of the kernels launched by the last trace/render call */
int pm_last_kernel_ms(pm_ctx* ctx, float* ms);
/* copies of the device-layout buffers, so a CPU checker can walk the very same tree
 * (cap in bytes; *needed receives the full size; data may be NULL to query) */
int pm_get_node_buffer(pm_ctx* ctx, void* data, uint64_t cap, uint64_t* needed);
int pm_get_prim_buffer(pm_ctx* ctx, void* data, uint64_t cap, uint64_t* needed);


/* ============================================================================================
 * Integrate::render() (integrators/base.rs:32-45,82-219; integrators/simple_path.rs:105-269)
 * The host (scene_parsing / camera / lights / materials of the reference) flattens what it built
 * into the structs below; the library runs the wavefront pipeline
 *   raygen -> [closest-hit -> shade (+NEE ray) -> any-hit shadow] x depth -> accumulate
 * and keeps the f64 film (camera/film.rs:302-307) in HBM.
 * ============================================================================================ */

/* sampling/spectrum.rs SpectrumEnum, evaluated exactly as the reference evaluates each variant.
 * `table` = 471 floats for integer wavelengths 360..830 nm (DenselySampledSpectrum::at, :225-239). */
enum { PM_SPEC_CONSTANT = 0,        /* a = c                                   (:163-186) */
       PM_SPEC_DENSE = 1,           /* table                                   (:188-244) */
       PM_SPEC_RGB_ALBEDO = 2,      /* sigmoid(c0=a,c1=b,c2=c)                 (:922-948) */
       PM_SPEC_RGB_UNBOUNDED = 3,   /* d * sigmoid(a,b,c)                      (:950-979) */
       PM_SPEC_RGB_ILLUMINANT = 4,  /* d * sigmoid(a,b,c) * table (illuminant) (:981-1019) */
       PM_SPEC_BLACKBODY = 5 };     /* a = T [K], b = normalisation factor     (:376-437) */
typedef struct PmSpectrum { int32_t kind; float a, b, c, d; const float* table; } PmSpectrum;

/* materials/material.rs:111-201.  The reference has constant textures only (materials/textures.rs:63-90).  EXTENSION for
 * scenes/test.pbrt: texture == PM_TEX_CHECKERBOARD makes a DIFFUSE material's reflectance a 2-D checkerboard of `spectrum`
 * (tex1) and `spectrum2` (tex2) over st = (su * u + du, sv * v + dv) (UvMapping::map, materials/mappings.rs:66-84):
 * tex1 where floor(s) + floor(t) is even, tex2 where it is odd - point-sampled, as no ray differentials are carried. */
enum { PM_MAT_DIFFUSE = 0, PM_MAT_DIELECTRIC = 1 };
enum { PM_TEX_NONE = 0, PM_TEX_CHECKERBOARD = 1 };
typedef struct PmMaterial {
    int32_t type;
    PmSpectrum spectrum;             /* DIFFUSE: reflectance (tex1); DIELECTRIC: eta */
    float u_roughness, v_roughness;  /* DIELECTRIC */
    int32_t remap_roughness;
    int32_t texture;                 /* PM_TEX_* (DIFFUSE only) */
    float su, sv, du, dv;            /* UvMapping */
    PmSpectrum spectrum2;            /* tex2 */
} PmMaterial;

/* lights/{point,directional,infinite}.rs; `spectrum` is the light's DenselySampledSpectrum (471 floats) */
enum { PM_LIGHT_POINT = 0, PM_LIGHT_DIRECTIONAL = 1, PM_LIGHT_UNIFORM_INFINITE = 2,
       /* EXTENSION (the reference has no area light: lights/base.rs:18-22, although GeometricPrimitive carries an `area_light`
        * field, primitives/single.rs:13-19): a diffuse emitter on the patch / triangle / sphere `prim_id`, radiance scale * spectrum on
        * the side its surface normal faces (both sides if two_sided).  Sampled through BilinearPatch::sample_with_context's
        * area-sampling branch (shapes/bilinear_patch.rs:260-352; the spherical-rectangle branch is todo!() upstream) or - round 3 -
        * Sphere::sample_with_context (shapes/sphere.rs:135-213: by area from inside, within the subtended cone from outside,
        * restated operation for operation INCLUDING upstream's cos_alpha expression, which differs from pbrt-v4's and crowds the
        * samples onto the point of the sphere nearest to the receiver); seen by rays through
        * SurfaceInteraction::emitted_radiance (core/interaction.rs:179-184, simple_path.rs:156-158). */
       PM_LIGHT_DIFFUSE_AREA = 3 };
typedef struct PmLight {
    int32_t type; float scale; float render_from_light[16]; const float* spectrum;
    uint32_t prim_id; int32_t two_sided;      /* PM_LIGHT_DIFFUSE_AREA only */
} PmLight;

/* media/medium.rs:41-69 HomogeneousMedium::new(sigma_a, sigma_s, sigma_scale, emission = 0, g): the two coefficient spectra
 * densely sampled (471 floats, 360..830 nm) and ALREADY multiplied by sigma_scale; Henyey-Greenstein asymmetry g
 * (media/phase_function.rs:27-80).  EXTENSION: no reference integrator consumes a medium, and instead of MediumInterface
 * (media/medium_interface.rs) the medium fills the axis-aligned box `bounds` (min xyz, max xyz, render space; +-INFINITY =
 * all of space).  Only PM_INTEGRATOR_VOL_PATH looks at it. */
typedef struct PmMedium { const float* sigma_a; const float* sigma_s; float g; float bounds[6]; } PmMedium;

/* camera/perspective.rs:20-25 + camera/base.rs:260-327: the two matrices generate_ray uses */
typedef struct PmCamera {
    float camera_from_raster[16];
    float render_from_camera[16];
    float lens_radius, focal_distance;
} PmCamera;

/* camera/film.rs:108-121 RGBFilm + camera/sensor.rs:14-20 PixelSensor + imaging/filter.rs */
enum { PM_FILTER_BOX = 0, PM_FILTER_TRIANGLE = 1, PM_FILTER_GAUSSIAN = 2 };
typedef struct PmFilm {
    int32_t full_resolution[2];
    int32_t pixel_bounds[4];             /* x0, y0, x1, y1 */
    float max_component_value, imaging_ratio;
    float output_rgb_from_sensor_rgb[9];
    const float* sensor_rgb_bar;         /* r_bar, g_bar, b_bar: 3 x 471 floats */
    int32_t filter_kind; float filter_radius[2];
    int32_t filter_nx, filter_ny;        /* GAUSSIAN: FilterSampler table (imaging/filter.rs:187-216), row-major */
    const float* filter_vals;
} PmFilm;

enum { PM_INTEGRATOR_SIMPLE_PATH = 0, PM_INTEGRATOR_RANDOM_WALK = 1,
       /* EXTENSION (integrators/base.rs:26-30 has no volumetric integrator; scenes/test.pbrt asks for "volpath"):
        * SimplePathIntegrator whose rays are first sampled for a real collision inside the medium of pm_set_medium
        * (distance ~ sigma_t at the first wavelength; absorbed -> the path ends; scattered -> a path vertex with next-event
        * estimation through the phase function and an HG-sampled continuation) and whose shadow rays are attenuated by the
        * transmittance exp(-sigma_t d).  Without a medium it is SimplePath. */
       PM_INTEGRATOR_VOL_PATH = 2 };
typedef struct PmRenderParams {
    int32_t integrator, max_depth, sample_lights, sample_bsdf;
    int32_t spp;              /* sampler.samples_per_pixel() */
    uint32_t seed;            /* IndependentSampler seed (sampler.rs:77: None -> 0) */
    int32_t window[4];        /* pixel window x0,y0,x1,y1 inside pixel_bounds (tile split across GPUs) */
    int32_t s0, s1;           /* sample-index window [s0,s1) (sample split across GPUs) */
} PmRenderParams;

typedef struct PmRenderStats {
    uint64_t samples, closest_rays, shadow_rays;
    uint64_t passes, kernels_launched;
    double kernel_ms;          /* device time of all kernels of the last pm_render (HIP events) */
    double state_bytes_per_path;
    /* per kernel class: 0 raygen, 1 closest_hit, 2 shade, 3 shadow (any-hit), 4 accumulate */
    double ms_by_kernel[5];
    uint64_t launches_by_kernel[5];
    /* traversal counters of this render (only with pm_set_counting(ctx, 1)) */
    uint64_t closest_nodes, closest_prims, shadow_nodes, shadow_prims;
    uint64_t exact_tests, fallback_walks;   /* both ray kinds together */
} PmRenderStats;

int pm_set_materials(pm_ctx* ctx, const PmMaterial* mats, uint32_t n, uint32_t struct_bytes /* sizeof(PmMaterial) */);
int pm_set_prim_materials(pm_ctx* ctx, const uint32_t* material_id_per_prim, uint32_t n);
int pm_set_lights(pm_ctx* ctx, const PmLight* lights, uint32_t n, uint32_t struct_bytes /* sizeof(PmLight) */);
/* NULL removes the medium.  A rejected medium (negative / NaN coefficient) leaves the previous state untouched. */
int pm_set_medium(pm_ctx* ctx, const PmMedium* medium_or_null, uint32_t struct_bytes /* sizeof(PmMedium) */);
int pm_set_camera(pm_ctx* ctx, const PmCamera* cam, uint32_t struct_bytes /* sizeof(PmCamera) */);
int pm_set_film(pm_ctx* ctx, const PmFilm* film, uint32_t struct_bytes /* sizeof(PmFilm) */);   /* (re)allocates and clears the film */
/* Integrate::render() restricted to a pixel/sample window; accumulates into the film in sample-index order.
 * The window is rendered in passes of at most 2^26 paths (about 200 bytes of device memory per path, allocated on first
 * use and kept; never more than half of what is free on the device; PBRT_MI355_PASS_PATHS lowers it).  The film does not
 * depend on how the window is cut into passes.  A call of at least 2^22 paths and two sample indices runs its passes
 * alternately on two streams of the library's own (same memory: the passes are half as large; PBRT_MI355_LANES=1 keeps one
 * stream): the call is blocking all the same, the film and every statistic but the per-class times - which then overlap - are
 * bit-identical, and PmRenderStats.passes counts the halves. */
int pm_render(pm_ctx* ctx, const PmRenderParams* params, uint32_t struct_bytes /* sizeof(PmRenderParams) */);
int pm_film_clear(pm_ctx* ctx);
/* RGBPixel{rgb_sum[3], weight_sum} as 4 doubles per pixel, row-major over pixel_bounds; host or device pointer */
int pm_film_read(pm_ctx* ctx, double* rgbw);
/* film += rgbw (merging partial films from other GPUs: sample split) */
int pm_film_add(pm_ctx* ctx, const double* rgbw);
/* RGBFilm::get_pixel_rgb for every pixel (camera/film.rs:261-279): 3 floats per pixel */
int pm_film_image(pm_ctx* ctx, float* rgb);
int pm_get_render_stats(pm_ctx* ctx, PmRenderStats* out, uint32_t struct_bytes /* sizeof(PmRenderStats) */);
/* queue occupancy of the last pm_render: entries of the closest-hit queue and of the shadow queue at bounce 0 .. max_depth
 * (summed over passes); *n_bounces receives max_depth + 1, at most cap entries are written to each array (either may be NULL) */
int pm_get_queue_counts(pm_ctx* ctx, uint64_t* closest, uint64_t* shadow, uint32_t cap, uint32_t* n_bounces);
/* test hook: unweighted radiance L[4] + lambda[4] + pdf[4] + filter weight of ONE (pixel, sample), recomputed
 * on the device with the same kernels (13 floats) */
int pm_debug_sample(pm_ctx* ctx, const PmRenderParams* params, int32_t px, int32_t py, int32_t sample_index, float* out13,
                    uint32_t struct_bytes /* sizeof(PmRenderParams) */);

/* test hook: the first n uniforms that IndependentSampler::get_1d returns after start_pixel_sample((px,py), sample_index,
 * dimension) with this seed (sampling/sampler.rs:104-124, util/rng.rs:15-50), drawn by the device's PCG32 / murmur3 code */
int pm_debug_uniforms(pm_ctx* ctx, int32_t px, int32_t py, uint32_t seed, uint64_t sample_index, uint64_t dimension, int32_t n,
                      float* out);

/* test hook: the device's transcendentals at n arguments (host pointers).  The reference calls f32 libm functions
 * (sampling/spectrum.rs:1004-1018 atanh / cosh, sampling/routines.rs sin / cos of 2 pi u, media/medium.rs exp / ln); the device
 * evaluates them in f64 and rounds once (csrc/pm_crmath.hpp).  which: 0 atanh (|x| < 1), 1 cosh (|x| <= 2.2), 2 exp, 3 log,
 * 4 / 5 sin / cos of an azimuth in [0, 2 pi], 6 / 7 sin / cos of the concentric-disk angle in [-pi/4, 3 pi/4] */
int pm_debug_math(pm_ctx* ctx, int32_t which, const float* x, int32_t n, float* out);
/* Test hook (round 3): Sphere::sample_with_context (shapes/sphere.rs:135-213) of the sphere `prim_id` of the scene the last
 * pm_render uploaded, evaluated on the device.  Item i: in8 = ctx_p[3], p_origin[3] (ctx.offset_ray_origin_towards(centre), or
 * ctx_p for a point in a medium), u[2]; out8 = ok (1 / 0), p[3], n[3], pdf.  Host pointers. */
int pm_debug_sphere_sample(pm_ctx* ctx, uint32_t prim_id, const float* in8, int32_t n, float* out8);

/* test hook (host state only, works on a host-only context): out4 = { medium set ? 1 : 0, g, sigma_a[0], sigma_s[0] } of the
 * medium the next render would use - shows that a rejected pm_set_medium left the previous state in place */
int pm_debug_medium_state(pm_ctx* ctx, float* out4);

#ifdef __cplusplus
}
#endif

/* Layout contract of the structs above (LP64): bindings in other languages (rust/pbrt-mi355-sys, pbrt-rs_amd/_capi.py) are
 * checked against these sizes by tests/test_rust_bindings.py. */
#ifdef __cplusplus
#define PM_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define PM_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
PM_STATIC_ASSERT(sizeof(PmRay) == 32, "PmRay");
PM_STATIC_ASSERT(sizeof(PmHit) == 16, "PmHit");
PM_STATIC_ASSERT(sizeof(PmSphere) == 152, "PmSphere");
PM_STATIC_ASSERT(sizeof(PmAccelInfo) == 80, "PmAccelInfo");
PM_STATIC_ASSERT(sizeof(PmCounters) == 48, "PmCounters");
PM_STATIC_ASSERT(sizeof(PmSpectrum) == 32, "PmSpectrum");
PM_STATIC_ASSERT(sizeof(PmMaterial) == 104, "PmMaterial");
PM_STATIC_ASSERT(sizeof(PmLight) == 88, "PmLight");
PM_STATIC_ASSERT(sizeof(PmMedium) == 48, "PmMedium");
PM_STATIC_ASSERT(sizeof(PmCamera) == 136, "PmCamera");
PM_STATIC_ASSERT(sizeof(PmFilm) == 112, "PmFilm");
PM_STATIC_ASSERT(sizeof(PmRenderParams) == 48, "PmRenderParams");
PM_STATIC_ASSERT(sizeof(PmRenderStats) == 184, "PmRenderStats");
#endif
