//! Raw declarations of `include/pbrt_mi355.h` — every struct, constant and entry point, field for field.
//!
//! This crate cannot be compiled in the image the library is built in (no Rust toolchain there), so
//! `tests/test_rust_bindings.py` parses this file and the C header and checks that struct names, field order, field
//! types, computed sizes and every function signature agree.  Keep one item per line group as below: the checker reads
//! `pub struct Name { pub field: Type, ... }` and `pub fn name(arg: Type, ...) -> Ret;`.
#![allow(non_camel_case_types, non_snake_case)]

use std::os::raw::c_char;

pub const PM_OK: i32 = 0;
pub const PM_ERR_INVALID: i32 = 1;
pub const PM_ERR_HIP: i32 = 2;
pub const PM_ERR_NO_ACCEL: i32 = 3;
pub const PM_ERR_UNSUPPORTED: i32 = 4;
pub const PM_ERR_STACK_OVERFLOW: i32 = 5;
pub const PM_MISS: u32 = 0xFFFF_FFFF;
pub const PM_ABI_VERSION: u32 = 3;
pub const PM_NOALLOC_MAX_RAYS: u32 = 1 << 18;

pub const PM_SPLIT_HLBVH: i32 = 0;
pub const PM_SPLIT_SAH: i32 = 1;
pub const PM_SPLIT_MIDDLE: i32 = 2;
pub const PM_SPLIT_EQUAL_COUNTS: i32 = 3;

pub const PM_SPEC_CONSTANT: i32 = 0;
pub const PM_SPEC_DENSE: i32 = 1;
pub const PM_SPEC_RGB_ALBEDO: i32 = 2;
pub const PM_SPEC_RGB_UNBOUNDED: i32 = 3;
pub const PM_SPEC_RGB_ILLUMINANT: i32 = 4;
pub const PM_SPEC_BLACKBODY: i32 = 5;

pub const PM_MAT_DIFFUSE: i32 = 0;
pub const PM_MAT_DIELECTRIC: i32 = 1;

pub const PM_TEX_NONE: i32 = 0;
pub const PM_TEX_CHECKERBOARD: i32 = 1;

pub const PM_LIGHT_POINT: i32 = 0;
pub const PM_LIGHT_DIRECTIONAL: i32 = 1;
pub const PM_LIGHT_UNIFORM_INFINITE: i32 = 2;
pub const PM_LIGHT_DIFFUSE_AREA: i32 = 3;

pub const PM_FILTER_BOX: i32 = 0;
pub const PM_FILTER_TRIANGLE: i32 = 1;
pub const PM_FILTER_GAUSSIAN: i32 = 2;

pub const PM_INTEGRATOR_SIMPLE_PATH: i32 = 0;
pub const PM_INTEGRATOR_RANDOM_WALK: i32 = 1;
pub const PM_INTEGRATOR_VOL_PATH: i32 = 2;

/// Opaque context (`typedef struct pm_ctx pm_ctx`): one per GPU.
#[repr(C)]
pub struct pm_ctx {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PmRay {
    pub o: [f32; 3],
    pub tmax: f32,
    pub d: [f32; 3],
    pub time: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PmHit {
    pub prim_id: u32,
    pub t: f32,
    pub u: f32,
    pub v: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PmSphere {
    pub radius: f32,
    pub z_min: f32,
    pub z_max: f32,
    pub phi_max_deg: f32,
    pub render_from_object: [f32; 16],
    pub object_from_render: [f32; 16],
    pub reverse_orientation: u32,
    pub _pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PmAccelInfo {
    pub n_prims: u32,
    pub n_nodes: u32,
    pub node_bytes: u32,
    pub prim_record_bytes: [u32; 3],
    pub node_buffer_bytes: u64,
    pub prim_buffer_bytes: u64,
    pub bounds: [f32; 6],
    pub build_seconds: f64,
    pub flat_prims: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PmCounters {
    pub rays: u64,
    pub nodes_visited: u64,
    pub prims_tested: u64,
    pub stack_overflows: u64,
    pub exact_tests: u64,
    pub fallback_walks: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PmSpectrum {
    pub kind: i32,
    pub a: f32,
    pub b: f32,
    pub c: f32,
    pub d: f32,
    pub table: *const f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PmMaterial {
    pub type_: i32,
    pub spectrum: PmSpectrum,
    pub u_roughness: f32,
    pub v_roughness: f32,
    pub remap_roughness: i32,
    pub texture: i32,
    pub su: f32,
    pub sv: f32,
    pub du: f32,
    pub dv: f32,
    pub spectrum2: PmSpectrum,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PmLight {
    pub type_: i32,
    pub scale: f32,
    pub render_from_light: [f32; 16],
    pub spectrum: *const f32,
    pub prim_id: u32,
    pub two_sided: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PmMedium {
    pub sigma_a: *const f32,
    pub sigma_s: *const f32,
    pub g: f32,
    pub bounds: [f32; 6],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PmCamera {
    pub camera_from_raster: [f32; 16],
    pub render_from_camera: [f32; 16],
    pub lens_radius: f32,
    pub focal_distance: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PmFilm {
    pub full_resolution: [i32; 2],
    pub pixel_bounds: [i32; 4],
    pub max_component_value: f32,
    pub imaging_ratio: f32,
    pub output_rgb_from_sensor_rgb: [f32; 9],
    pub sensor_rgb_bar: *const f32,
    pub filter_kind: i32,
    pub filter_radius: [f32; 2],
    pub filter_nx: i32,
    pub filter_ny: i32,
    pub filter_vals: *const f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PmRenderParams {
    pub integrator: i32,
    pub max_depth: i32,
    pub sample_lights: i32,
    pub sample_bsdf: i32,
    pub spp: i32,
    pub seed: u32,
    pub window: [i32; 4],
    pub s0: i32,
    pub s1: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PmRenderStats {
    pub samples: u64,
    pub closest_rays: u64,
    pub shadow_rays: u64,
    pub passes: u64,
    pub kernels_launched: u64,
    pub kernel_ms: f64,
    pub state_bytes_per_path: f64,
    pub ms_by_kernel: [f64; 5],
    pub launches_by_kernel: [u64; 5],
    pub closest_nodes: u64,
    pub closest_prims: u64,
    pub shadow_nodes: u64,
    pub shadow_prims: u64,
    pub exact_tests: u64,
    pub fallback_walks: u64,
}

extern "C" {
    pub fn pm_abi_version() -> u32;
    pub fn pm_create(device_id: i32, out: *mut *mut pm_ctx) -> i32;
    pub fn pm_destroy(ctx: *mut pm_ctx);
    pub fn pm_last_error(ctx: *mut pm_ctx) -> *const c_char;
    pub fn pm_add_spheres(ctx: *mut pm_ctx, spheres: *const PmSphere, n: u32, first_prim_id: *mut u32, struct_bytes: u32) -> i32;
    pub fn pm_add_patch_mesh(ctx: *mut pm_ctx, pos_xyz: *const f32, n_vert: u32, idx4: *const u32, n_patch: u32, normals_or_null: *const f32, reverse_orientation: i32, transform_swaps_handedness: i32, first_prim_id: *mut u32) -> i32;
    pub fn pm_add_patch_mesh_uv(ctx: *mut pm_ctx, pos_xyz: *const f32, n_vert: u32, idx4: *const u32, n_patch: u32, normals_or_null: *const f32, uv_or_null: *const f32, reverse_orientation: i32, transform_swaps_handedness: i32, first_prim_id: *mut u32) -> i32;
    pub fn pm_add_triangle_mesh(ctx: *mut pm_ctx, pos_xyz: *const f32, n_vert: u32, idx3: *const u32, n_tri: u32, first_prim_id: *mut u32) -> i32;
    pub fn pm_clear_geometry(ctx: *mut pm_ctx) -> i32;
    pub fn pm_build_accel(ctx: *mut pm_ctx, split_method: i32, max_prims_in_node: i32, info_or_null: *mut PmAccelInfo, struct_bytes: u32) -> i32;
    pub fn pm_trace_closest(ctx: *mut pm_ctx, rays: *const PmRay, n: u64, out: *mut PmHit) -> i32;
    pub fn pm_trace_any(ctx: *mut pm_ctx, rays: *const PmRay, n: u64, out: *mut u8) -> i32;
    pub fn pm_trace_closest_noalloc(ctx: *mut pm_ctx, d_rays: *const PmRay, n: u64, d_out: *mut PmHit) -> i32;
    pub fn pm_set_counting(ctx: *mut pm_ctx, enabled: i32) -> i32;
    pub fn pm_reset_counters(ctx: *mut pm_ctx) -> i32;
    pub fn pm_get_counters(ctx: *mut pm_ctx, out: *mut PmCounters, struct_bytes: u32) -> i32;
    pub fn pm_last_kernel_ms(ctx: *mut pm_ctx, ms: *mut f32) -> i32;
    pub fn pm_get_node_buffer(ctx: *mut pm_ctx, data: *mut std::ffi::c_void, cap: u64, needed: *mut u64) -> i32;
    pub fn pm_get_prim_buffer(ctx: *mut pm_ctx, data: *mut std::ffi::c_void, cap: u64, needed: *mut u64) -> i32;
    pub fn pm_set_materials(ctx: *mut pm_ctx, mats: *const PmMaterial, n: u32, struct_bytes: u32) -> i32;
    pub fn pm_set_prim_materials(ctx: *mut pm_ctx, material_id_per_prim: *const u32, n: u32) -> i32;
    pub fn pm_set_lights(ctx: *mut pm_ctx, lights: *const PmLight, n: u32, struct_bytes: u32) -> i32;
    pub fn pm_set_medium(ctx: *mut pm_ctx, medium_or_null: *const PmMedium, struct_bytes: u32) -> i32;
    pub fn pm_set_camera(ctx: *mut pm_ctx, cam: *const PmCamera, struct_bytes: u32) -> i32;
    pub fn pm_set_film(ctx: *mut pm_ctx, film: *const PmFilm, struct_bytes: u32) -> i32;
    pub fn pm_render(ctx: *mut pm_ctx, params: *const PmRenderParams, struct_bytes: u32) -> i32;
    pub fn pm_film_clear(ctx: *mut pm_ctx) -> i32;
    pub fn pm_film_read(ctx: *mut pm_ctx, rgbw: *mut f64) -> i32;
    pub fn pm_film_add(ctx: *mut pm_ctx, rgbw: *const f64) -> i32;
    pub fn pm_film_image(ctx: *mut pm_ctx, rgb: *mut f32) -> i32;
    pub fn pm_get_render_stats(ctx: *mut pm_ctx, out: *mut PmRenderStats, struct_bytes: u32) -> i32;
    pub fn pm_get_queue_counts(ctx: *mut pm_ctx, closest: *mut u64, shadow: *mut u64, cap: u32, n_bounces: *mut u32) -> i32;
    pub fn pm_debug_sample(ctx: *mut pm_ctx, params: *const PmRenderParams, px: i32, py: i32, sample_index: i32, out13: *mut f32, struct_bytes: u32) -> i32;
    pub fn pm_debug_uniforms(ctx: *mut pm_ctx, px: i32, py: i32, seed: u32, sample_index: u64, dimension: u64, n: i32, out: *mut f32) -> i32;
    pub fn pm_debug_math(ctx: *mut pm_ctx, which: i32, x: *const f32, n: i32, out: *mut f32) -> i32;
    pub fn pm_debug_sphere_sample(ctx: *mut pm_ctx, prim_id: u32, in8: *const f32, n: i32, out8: *mut f32) -> i32;
    pub fn pm_debug_medium_state(ctx: *mut pm_ctx, out4: *mut f32) -> i32;
}

// Layout guards: the sizes the C header asserts (include/pbrt_mi355.h, PM_STATIC_ASSERT block).  A field added on one side
// only fails the build here as it fails tests/test_rust_bindings.py there.
const _: () = {
    use core::mem::size_of;
    assert!(size_of::<PmRay>() == 32);
    assert!(size_of::<PmHit>() == 16);
    assert!(size_of::<PmSphere>() == 152);
    assert!(size_of::<PmAccelInfo>() == 80);
    assert!(size_of::<PmCounters>() == 48);
    assert!(size_of::<PmSpectrum>() == 32);
    assert!(size_of::<PmMaterial>() == 104);
    assert!(size_of::<PmLight>() == 88);
    assert!(size_of::<PmMedium>() == 48);
    assert!(size_of::<PmCamera>() == 136);
    assert!(size_of::<PmFilm>() == 112);
    assert!(size_of::<PmRenderParams>() == 48);
    assert!(size_of::<PmRenderStats>() == 184);
};
