//! `src/primitives/gpu.rs` — `GpuAggregate`: the MI355X wide-BVH aggregate behind pbrt-rs's `Primitive` trait
//! (`src/primitives/mod.rs:21-27`), a drop-in for `BVHAggregate` (`src/primitives/aggregates.rs:25-47,532-655`).
//!
//! SOURCE ONLY: the image this repository is built in has no Rust toolchain, so this file has not been compiled.  It is
//! written against the reference tree as it stands plus the small additions listed under "needs" (accessors for fields
//! that are private today, and the new enum variant).  The C++ mirror `include/pbrt_mi355.hpp::GpuAggregate` and the Python
//! mirror `pbrt-rs_amd/aggregate.py` are the same adapter and ARE exercised by the tests.
//!
//! needs:
//!   * `PrimitiveEnum::Gpu(GpuAggregate)` added to the closed enum at `src/primitives/mod.rs:14-19`;
//!   * `SimplePrimitive::shape()` / `GeometricPrimitive::shape()` returning `&Arc<ShapeEnum>` (fields are private,
//!     `src/primitives/single.rs:13-19,72-75`);
//!   * `Sphere::{radius,z_min,z_max,phi_max_degrees,render_from_object,object_from_render,reverse_orientation}()` getters
//!     (`src/shapes/sphere.rs:19-30`), `BilinearPatch::{mesh_idx,blp_idx}()` (`src/shapes/bilinear_patch.rs:24-28`).
use std::mem::size_of;
use std::sync::Arc;

use pbrt_mi355_sys as sys;

use crate::{
    core::{Bounds3f, Float, Point3f, Ray},
    primitives::{BVHSplitMethod, Primitive, PrimitiveEnum},
    shapes::{BilinearPatchMesh, ShapeEnum, ShapeIntersection},
};

pub struct GpuAggregate {
    pub(crate) ctx: *mut sys::pm_ctx,
    bounds: Bounds3f,
    /// prim_id (insertion order across pm_add_* calls) -> the reference primitive it came from
    prims: Vec<Arc<PrimitiveEnum>>,
}
// immutable after new(); the C ABI asks callers to serialise calls on one context (see intersect below)
unsafe impl Send for GpuAggregate {}
unsafe impl Sync for GpuAggregate {}

fn check(ctx: *mut sys::pm_ctx, rc: i32) {
    if rc != sys::PM_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(sys::pm_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("libpbrt_mi355: status {rc}: {msg}");      // the reference panics on invariant violations too
    }
}

impl GpuAggregate {
    /// Same arguments as `BVHAggregate::new`.  Primitives are handed to the library in their order in `prims`, so that a hit's
    /// `prim_id` indexes `prims`; consecutive patches of one mesh go down as one `pm_add_patch_mesh` call.
    pub fn new(prims: Vec<Arc<PrimitiveEnum>>, max_prims_in_node: u8, split_method: BVHSplitMethod, device_id: i32) -> Self {
        assert!(!prims.is_empty(), "aggregate needs at least one primitive");       // aggregates.rs:60-64
        let mut ctx = std::ptr::null_mut();
        assert_eq!(unsafe { sys::pm_abi_version() }, sys::PM_ABI_VERSION, "libpbrt_mi355.so and pbrt-mi355-sys disagree on the ABI");
        assert_eq!(unsafe { sys::pm_create(device_id, &mut ctx) }, sys::PM_OK, "no MI355X visible (there is no CPU fallback)");
        let mut first = 0u32;
        let mut i = 0;
        while i < prims.len() {
            match &**prims[i].shape() {
                ShapeEnum::Sphere(s) => {
                    let sp = sys::PmSphere {
                        radius: s.radius(), z_min: s.z_min(), z_max: s.z_max(), phi_max_deg: s.phi_max_degrees(),
                        render_from_object: s.render_from_object().matrix().to_row_major(),
                        object_from_render: s.object_from_render().matrix().to_row_major(),
                        reverse_orientation: s.reverse_orientation() as u32, _pad: 0,
                    };
                    check(ctx, unsafe { sys::pm_add_spheres(ctx, &sp, 1, &mut first, size_of::<sys::PmSphere>() as u32) });
                    i += 1;
                }
                ShapeEnum::BilinearPatch(p0) => {
                    // the run of patches that belong to the same mesh
                    let mesh: &BilinearPatchMesh = BilinearPatchMesh::get(p0.mesh_idx()).unwrap();
                    let mut idx4: Vec<u32> = Vec::new();
                    let mut j = i;
                    while j < prims.len() {
                        match &**prims[j].shape() {
                            ShapeEnum::BilinearPatch(p) if p.mesh_idx() == p0.mesh_idx() => {
                                idx4.extend(mesh.indices[4 * p.blp_idx()..4 * p.blp_idx() + 4].iter().map(|&v| v as u32));
                                j += 1;
                            }
                            _ => break,
                        }
                    }
                    let pos: Vec<f32> = mesh.positions.iter().flat_map(|p: &Point3f| [p.x(), p.y(), p.z()]).collect();
                    let nrm: Option<Vec<f32>> = mesh.normals.as_ref().map(|n| n.iter().flat_map(|n| [n.x(), n.y(), n.z()]).collect());
                    check(ctx, unsafe {
                        sys::pm_add_patch_mesh(ctx, pos.as_ptr(), mesh.positions.len() as u32, idx4.as_ptr(), (j - i) as u32,
                                               nrm.as_ref().map_or(std::ptr::null(), |n| n.as_ptr()),
                                               mesh.reverse_orientation as i32, mesh.transform_swaps_handedness as i32, &mut first)
                    });
                    i = j;
                }
            }
        }
        let split = match split_method {
            BVHSplitMethod::HLBVH => sys::PM_SPLIT_HLBVH,
            BVHSplitMethod::SAH => sys::PM_SPLIT_SAH,
            BVHSplitMethod::Middle => sys::PM_SPLIT_MIDDLE,
            BVHSplitMethod::EqualCounts => sys::PM_SPLIT_EQUAL_COUNTS,
        };
        let mut info = sys::PmAccelInfo::default();
        check(ctx, unsafe { sys::pm_build_accel(ctx, split, max_prims_in_node as i32, &mut info, size_of::<sys::PmAccelInfo>() as u32) });
        let b = info.bounds;
        Self { ctx, bounds: Bounds3f::new(Point3f::new(b[0], b[1], b[2]), Point3f::new(b[3], b[4], b[5])), prims }
    }

    /// The hot call: `Primitive::intersect` for a whole batch (host slices; device pointers work through the raw call).
    pub fn intersect_batch(&self, rays: &[sys::PmRay], out: &mut [sys::PmHit]) {
        assert_eq!(rays.len(), out.len());
        check(self.ctx, unsafe { sys::pm_trace_closest(self.ctx, rays.as_ptr(), rays.len() as u64, out.as_mut_ptr()) });
    }
    pub fn intersect_p_batch(&self, rays: &[sys::PmRay], out: &mut [u8]) {
        assert_eq!(rays.len(), out.len());
        check(self.ctx, unsafe { sys::pm_trace_any(self.ctx, rays.as_ptr(), rays.len() as u64, out.as_mut_ptr()) });
    }
    fn to_pm(ray: &Ray, t_max: Option<Float>) -> sys::PmRay {
        sys::PmRay { o: [ray.o.x(), ray.o.y(), ray.o.z()], tmax: t_max.unwrap_or(Float::INFINITY),
                     d: [ray.dir.x(), ray.dir.y(), ray.dir.z()], time: ray.time }
    }
}

impl Primitive for GpuAggregate {
    fn bounds(&self) -> Bounds3f { self.bounds }

    /// Per-ray compatibility shim: a one-ray batch (correct, but a GPU launch per ray - render through
    /// `WavefrontIntegrator`, which keeps whole passes of paths on the device).  The winning primitive then rebuilds its
    /// `SurfaceInteraction` with the reference's own code, so callers see exactly what `BVHAggregate::intersect` returns.
    fn intersect<'a>(&'a self, ray: &'a Ray, t_max: Option<Float>) -> Option<ShapeIntersection<'a>> {
        let r = Self::to_pm(ray, t_max);
        let mut h = sys::PmHit { prim_id: sys::PM_MISS, t: 0.0, u: 0.0, v: 0.0 };
        {
            let _serialise = CTX_LOCK.lock().unwrap();        // rayon workers call this concurrently; one context = one stream
            self.intersect_batch(std::slice::from_ref(&r), std::slice::from_mut(&mut h));
        }
        if h.prim_id == sys::PM_MISS { return None; }
        // the hit distance is bit-identical to the reference's, so re-intersecting the ONE winning primitive with
        // t_max just above it reproduces the reference's ShapeIntersection (material, area light, medium interface attached)
        self.prims[h.prim_id as usize].intersect(ray, Some(crate::core::next_float_up(h.t)))
    }

    fn intersect_p(&self, ray: &Ray, t_max: Option<Float>) -> bool {
        let r = Self::to_pm(ray, t_max);
        let mut o = 0u8;
        let _serialise = CTX_LOCK.lock().unwrap();
        self.intersect_p_batch(std::slice::from_ref(&r), std::slice::from_mut(&mut o));
        o != 0
    }
}

static CTX_LOCK: std::sync::Mutex<()> = std::sync::Mutex::new(());

impl Drop for GpuAggregate {
    fn drop(&mut self) { unsafe { sys::pm_destroy(self.ctx) } }
}
