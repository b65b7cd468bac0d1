"""Host-side mirror of the reference's integrator interface over the C ABI.

Reference: `SimplePathIntegrator::new(max_depth, sample_lights, sample_bsdf, camera, sampler, aggregate, lights)`
and `impl Integrate { fn render(&mut self) }` (src/integrators/simple_path.rs:40-66, src/integrators/base.rs:32-45).
`render()` runs the whole image (or a pixel/sample window of it: the multi-GPU split) through the wavefront
kernels and leaves the f64 film in HBM; `film()` / `image()` read it back (RGBFilm::get_image, film.rs:236-253).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import PM_INTEGRATOR_SIMPLE_PATH, PmRenderParams, PmRenderStats
from .aggregate import GpuAggregate


class WavefrontIntegrator:
    def __init__(self, max_depth, sample_lights, sample_bsdf, scene, spp, seed=0, device_id=0,
                 integrator=PM_INTEGRATOR_SIMPLE_PATH, split_method=_capi.PM_SPLIT_SAH, max_prims_in_node=4):
        self.scene = scene
        self.aggregate = GpuAggregate(scene.prims, max_prims_in_node=max_prims_in_node, device_id=device_id,
                                      split_method=split_method)
        self._lib = self.aggregate._lib
        self._h = self.aggregate._h
        ck = self.aggregate._check
        mats = scene.c_materials()
        ck(self._lib.pm_set_materials(self._h, mats, len(scene.materials), C.sizeof(_capi.PmMaterial)))
        pmat = scene.c_prim_materials()
        ck(self._lib.pm_set_prim_materials(self._h, pmat.ctypes.data, len(pmat)))
        lights = scene.c_lights()
        ck(self._lib.pm_set_lights(self._h, lights, len(scene.lights), C.sizeof(_capi.PmLight)))
        med = scene.c_medium() if getattr(scene, "medium", None) is not None else None
        if med is not None:
            ck(self._lib.pm_set_medium(self._h, C.byref(med), C.sizeof(med)))
        ck(self._lib.pm_set_camera(self._h, C.byref(scene.camera), C.sizeof(scene.camera)))
        film = scene.c_film()
        ck(self._lib.pm_set_film(self._h, C.byref(film), C.sizeof(film)))
        self.params = PmRenderParams()
        self.params.integrator = integrator
        self.params.max_depth = max_depth
        self.params.sample_lights = int(bool(sample_lights))
        self.params.sample_bsdf = int(bool(sample_bsdf))
        self.params.spp = spp
        self.params.seed = seed
        self.bounds = tuple(scene.film["pixel_bounds"])
        self.width = self.bounds[2] - self.bounds[0]
        self.height = self.bounds[3] - self.bounds[1]

    def _params(self, window, samples):
        p = PmRenderParams.from_buffer_copy(self.params)
        p.window[:] = list(window if window is not None else self.bounds)
        s0, s1 = samples if samples is not None else (0, self.params.spp)
        p.s0, p.s1 = s0, s1
        return p

    def render(self, window=None, samples=None):
        """Integrate::render() over `window` (x0,y0,x1,y1; default all pixels) and sample indices [s0,s1)."""
        p = self._params(window, samples)
        self.aggregate._check(self._lib.pm_render(self._h, C.byref(p), C.sizeof(p)))
        return self.stats()

    @staticmethod
    def wave_schedule(spp):
        """The sample-index waves of ImageTileIntegrate::image_tile_render (integrators/base.rs:85-127): [0,1), [1,2), [2,4),
        [4,8) ... the wave size doubles up to 64, the last wave ends at spp."""
        waves = []
        wave_start, wave_end, next_wave_size = 0, 1, 1
        while wave_start < spp:
            waves.append((wave_start, wave_end))
            wave_start = wave_end
            wave_end = min(spp, wave_end + next_wave_size)
            next_wave_size = min(2 * next_wave_size, 64)
        return waves

    def render_waves(self, window=None, on_wave=None):
        """Integrate::render() in the reference's waves.  After each wave the film holds every sample index below wave_end,
        so `on_wave(wave_start, wave_end, self)` may read a progressive image (`image()`, `film()`, `write_exr`).  The final
        film is bit-identical to one `render()` call: samples are accumulated in sample-index order either way."""
        stats = None
        for w0, w1 in self.wave_schedule(self.params.spp):
            stats = self.render(window, (w0, w1))
            if on_wave is not None:
                on_wave(w0, w1, self)
        return stats

    def clear(self):
        self.aggregate._check(self._lib.pm_film_clear(self._h))

    def film(self, out=None):
        """RGBPixel {rgb_sum[3], weight_sum} per pixel as float64 [H, W, 4]."""
        if out is None:
            out = np.empty((self.height, self.width, 4), np.float64)
            self.aggregate._check(self._lib.pm_film_read(self._h, out.ctypes.data))
        else:
            self.aggregate._check(self._lib.pm_film_read(self._h, C.c_void_p(out.data_ptr())))
        return out

    def film_add(self, other):
        ptr = other.ctypes.data if isinstance(other, np.ndarray) else C.c_void_p(other.data_ptr())
        self.aggregate._check(self._lib.pm_film_add(self._h, ptr))

    def image(self):
        """RGBFilm::get_pixel_rgb for every pixel: float32 [H, W, 3] in the output colour space."""
        out = np.empty((self.height, self.width, 3), np.float32)
        self.aggregate._check(self._lib.pm_film_image(self._h, out.ctypes.data))
        return out

    def write_exr(self, path):
        """RGBFilm::write_image -> Image::write_exr (camera/film.rs:255-259, imaging/image.rs:48-57)"""
        img = np.ascontiguousarray(self.image(), dtype=np.float32)
        if self._lib.pmh_write_exr(str(path).encode(), self.width, self.height, img.ctypes.data) != 0:
            raise OSError(f"could not write {path}")

    def stats(self):
        s = PmRenderStats()
        self.aggregate._check(self._lib.pm_get_render_stats(self._h, C.byref(s), C.sizeof(s)))
        out = {}
        for k, _ in PmRenderStats._fields_:
            v = getattr(s, k)
            out[k] = list(v) if hasattr(v, "__len__") else v
        return out

    def queue_counts(self):
        """(closest-hit queue entries, shadow queue entries) per bounce of the last render()"""
        n = C.c_uint32()
        self.aggregate._check(self._lib.pm_get_queue_counts(self._h, None, None, 0, C.byref(n)))
        a, b = np.zeros(n.value, np.uint64), np.zeros(n.value, np.uint64)
        self.aggregate._check(self._lib.pm_get_queue_counts(self._h, a.ctypes.data, b.ctypes.data, n.value, None))
        return a, b

    def debug_sample(self, px, py, sample_index):
        """(L[4], lambda[4], pdf[4], filter_weight) of one (pixel, sample) — test hook."""
        out = np.zeros(13, np.float32)
        p = self._params(None, None)
        self.aggregate._check(self._lib.pm_debug_sample(self._h, C.byref(p), px, py, sample_index, out.ctypes.data, C.sizeof(p)))
        return out
