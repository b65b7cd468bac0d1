//! `src/integrators/wavefront.rs` — `WavefrontIntegrator`: `Integrate::render()` (`src/integrators/base.rs:32-45`) on the
//! MI355X.  Constructor arguments are `SimplePathIntegrator::new`'s (`src/integrators/simple_path.rs:40-66`); `render()`
//! replaces `image_tile_render` (`base.rs:82-142`): the library runs raygen -> [closest-hit -> shade -> shadow] x depth ->
//! accumulate over the whole image and returns the f64 film sums, which are loaded into the reference's `RGBFilm` so that
//! `write_image` (`camera/film.rs:255-259`) produces the EXR exactly as before.
//!
//! SOURCE ONLY (no Rust toolchain in the build image).  `pbrt-rs_amd/integrator.py` and `include/pbrt_mi355.hpp` are the same
//! adapter, exercised by the tests; `pbrt-rs_amd/scenes.py::main_rs_scene` flattens the `main.rs` scene field by field.
//!
//! needs:
//!   * `IntegratorEnum::Wavefront(WavefrontIntegrator)` at `src/integrators/base.rs:26-30`;
//!   * flatteners `MaterialEnum::to_pm()`, `LightEnum::to_pm()` (each variant's fields + its spectrum sampled at 360..830 nm
//!     with `DenselySampledSpectrum::new`), `PerspectiveCamera::{camera_from_raster, render_from_camera, lens_radius,
//!     focal_distance}()` getters, `FilterSampler::table()`, `RGBFilm::load_sums(&[f64])` writing `RGBPixel{rgb_sum,
//!     weight_sum}` (`camera/film.rs:302-317`).
use std::mem::size_of;
use std::sync::Arc;

use pbrt_mi355_sys as sys;

use crate::{
    camera::{Camera, CameraEnum},
    core::{Float, Ray},
    integrators::Integrate,
    lights::LightEnum,
    primitives::{gpu::GpuAggregate, Primitive, PrimitiveEnum},
    sampling::sampler::{Sampler, SamplerEnum},
    shapes::ShapeIntersection,
};

pub struct WavefrontIntegrator {
    aggregate: GpuAggregate,
    camera: CameraEnum,
    params: sys::PmRenderParams,
    // host copies that the flattened structs point into (spectrum tables, filter table, sensor curves)
    _tables: Vec<Vec<f32>>,
}

impl WavefrontIntegrator {
    pub fn new(max_depth: usize, sample_lights: bool, sample_bsdf: bool, camera: CameraEnum, sampler: SamplerEnum,
               aggregate: PrimitiveEnum, lights: Vec<Arc<LightEnum>>) -> Self {
        let aggregate = match aggregate {
            PrimitiveEnum::Gpu(g) => g,
            _ => panic!("WavefrontIntegrator needs a GpuAggregate (GpuAggregate::new takes BVHAggregate::new's arguments)"),
        };
        let ctx = aggregate.ctx;
        let mut tables: Vec<Vec<f32>> = Vec::new();
        // materials: one per primitive, in primitive order (material.rs:111-201; constant textures only, as upstream)
        let (mats, prim_mat) = aggregate.flatten_materials(&mut tables);
        chk(ctx, unsafe { sys::pm_set_materials(ctx, mats.as_ptr(), mats.len() as u32, size_of::<sys::PmMaterial>() as u32) });
        chk(ctx, unsafe { sys::pm_set_prim_materials(ctx, prim_mat.as_ptr(), prim_mat.len() as u32) });
        let pm_lights: Vec<sys::PmLight> = lights.iter().map(|l| l.to_pm(&mut tables)).collect();
        chk(ctx, unsafe { sys::pm_set_lights(ctx, pm_lights.as_ptr(), pm_lights.len() as u32, size_of::<sys::PmLight>() as u32) });
        let cam = camera.to_pm();                                   // perspective.rs:20-25, base.rs:260-327
        chk(ctx, unsafe { sys::pm_set_camera(ctx, &cam, size_of::<sys::PmCamera>() as u32) });
        let film = camera.film().to_pm(&mut tables);                // film.rs:108-121, sensor.rs:14-20, filter.rs:177-216
        chk(ctx, unsafe { sys::pm_set_film(ctx, &film, size_of::<sys::PmFilm>() as u32) });
        let b = camera.film().pixel_bounds();
        let spp = sampler.samples_per_pixel() as i32;
        let params = sys::PmRenderParams {
            integrator: sys::PM_INTEGRATOR_SIMPLE_PATH, max_depth: max_depth as i32, sample_lights: sample_lights as i32,
            sample_bsdf: sample_bsdf as i32, spp, seed: sampler.seed() as u32,
            window: [b.min.x(), b.min.y(), b.max.x(), b.max.y()], s0: 0, s1: spp,
        };
        Self { aggregate, camera, params, _tables: tables }
    }
}

fn chk(ctx: *mut sys::pm_ctx, rc: i32) {
    if rc != sys::PM_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(sys::pm_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("libpbrt_mi355: status {rc}: {msg}");
    }
}

impl Integrate for WavefrontIntegrator {
    fn render(&mut self) {
        let ctx = self.aggregate.ctx;
        chk(ctx, unsafe { sys::pm_render(ctx, &self.params, size_of::<sys::PmRenderParams>() as u32) });                 // the whole image, every sample index
        let film = self.camera.film_mut();
        let b = film.pixel_bounds();
        let n = ((b.max.x() - b.min.x()) * (b.max.y() - b.min.y())) as usize;
        let mut rgbw = vec![0f64; 4 * n];                                        // RGBPixel { rgb_sum[3], weight_sum } per pixel
        chk(ctx, unsafe { sys::pm_film_read(ctx, rgbw.as_mut_ptr()) });
        film.load_sums(&rgbw);
        film.write_image(Default::default(), 1.0 / self.params.spp as Float);    // base.rs:138-141
    }

    fn intersect<'a>(&'a self, ray: &'a Ray, t_max: Option<Float>) -> Option<ShapeIntersection<'a>> {
        self.aggregate.intersect(ray, t_max)
    }

    fn intersect_p(&self, ray: &Ray, t_max: Option<Float>) -> bool {
        self.aggregate.intersect_p(ray, t_max)
    }
}
