"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes wrapper of oracle/liboracle.so, the CPU restatement of the reference hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under pbrt-rs_amd/ does.  Parity status: pinned by the reference's own known-answer
tests where they exist (tests/test_oracle_golden.py); BVH traversal, patch intersection,
interval arithmetic and the RNG/hash stream have no reference test => "parity unpinned"
 — the reference is Rust and cannot be built in this image.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

RAY_DTYPE = np.dtype([("o", "<f4", 3), ("tmax", "<f4"), ("d", "<f4", 3), ("time", "<f4")])
HIT_DTYPE = np.dtype([("prim_id", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])

SPLIT_SAH, SPLIT_MIDDLE, SPLIT_EQUAL_COUNTS = 1, 2, 3


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        f, i, u32, u64, p = C.c_float, C.c_int, C.c_uint32, C.c_uint64, C.c_void_p
        sig = {
            "ora_scene_create": (p, []), "ora_scene_destroy": (None, [p]),
            "ora_add_sphere": (u32, [p, f, f, f, f, p, p, i]),
            "ora_add_patch_mesh": (u32, [p, p, u32, p, u32, i, i]),
            "ora_add_patch_mesh_n": (u32, [p, p, u32, p, u32, p, i, i]),
            "ora_add_patch_mesh_uv": (u32, [p, p, u32, p, u32, p, p, i, i]),
            "ora_add_triangle_mesh": (u32, [p, p, u32, p, u32]),
            "ora_build": (i, [p, i, i]), "ora_scene_bounds": (None, [p, p]),
            "ora_trace_closest": (None, [p, p, u64, p, p, i, i]),
            "ora_trace_any": (None, [p, p, u64, p, p, i]),
            "ora_wide_trace": (None, [p, p, p, u64, p, i, p, i, i]),
            "ora_next_float_up": (f, [f]), "ora_next_float_down": (f, [f]), "ora_gamma": (f, [i]),
            "ora_exponent": (i, [f]), "ora_fast_exp": (f, [f]), "ora_gaussian": (f, [f, f, f]),
            "ora_erf_inv": (f, [f]), "ora_evaluate_polynomial": (f, [f, p, i]),
            "ora_difference_of_products": (f, [f, f, f, f]), "ora_solve_quadratic": (i, [f, f, f, p]),
            "ora_mat4_inverse": (i, [p, p]), "ora_xf_bounds": (None, [p, p, p]),
            "ora_visible_wavelengths_pdf": (f, [f]), "ora_sample_visible_wavelengths": (f, [f]),
            "ora_sphere_basic_intersect": (i, [f, f, f, f, p, p, p, p, f, p]),
            "ora_intersect_patch": (i, [p, p, f, p, p]),
            "ora_hash_pixel": (u64, [C.c_int32, C.c_int32, u32]),
            "ora_sampler_stream": (None, [C.c_int32, C.c_int32, u32, u64, u64, i, p]),
            "ora_pcg32_next": (u32, [p]), "ora_pcg32_new": (None, [u64, u64, p]),
            "ora_set_transcendental_mode": (None, [i]), "ora_get_transcendental_mode": (i, []),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class correctly_rounded:
    """with oracle.correctly_rounded(): sin / cos / atan2 / atanh / cosh / exp of the per-sample path are evaluated in f64 and
    rounded once (as the device kernels evaluate them) instead of by glibc's f32 routines (the faithful default)."""

    def __enter__(self):
        self._old = lib().ora_get_transcendental_mode()
        lib().ora_set_transcendental_mode(1)
        return self

    def __exit__(self, *exc):
        lib().ora_set_transcendental_mode(self._old)
        return False


class Scene:
    """Vec<Arc<PrimitiveEnum>> + BVHAggregate of the reference, on the CPU."""

    def __init__(self):
        self._l = lib()
        self._h = C.c_void_p(self._l.ora_scene_create())

    def add_sphere(self, radius, z_min, z_max, phi_max_deg, rfo, ofr, reverse=False):
        a, b = _f32(rfo).reshape(16), _f32(ofr).reshape(16)
        return self._l.ora_add_sphere(self._h, radius, z_min, z_max, phi_max_deg, a.ctypes.data, b.ctypes.data, int(reverse))

    def add_patch_mesh(self, positions, indices, reverse=False, swaps=False, normals=None, uv=None):
        pos = _f32(positions).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 4)
        nrm = None if normals is None else _f32(normals).reshape(-1, 3)
        tex = None if uv is None else _f32(uv).reshape(-1, 2)
        return self._l.ora_add_patch_mesh_uv(self._h, pos.ctypes.data, len(pos), idx.ctypes.data, len(idx),
                                             None if nrm is None else nrm.ctypes.data, None if tex is None else tex.ctypes.data,
                                             int(reverse), int(swaps))

    def add_triangle_mesh(self, positions, indices):
        pos = _f32(positions).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        return self._l.ora_add_triangle_mesh(self._h, pos.ctypes.data, len(pos), idx.ctypes.data, len(idx))

    def build(self, method=SPLIT_MIDDLE, max_prims_in_node=255):
        return self._l.ora_build(self._h, method, max_prims_in_node)

    def bounds(self):
        out = np.zeros(6, np.float32)
        self._l.ora_scene_bounds(self._h, out.ctypes.data)
        return out[:3], out[3:]

    def intersect(self, rays, nthreads=8, brute_force=False):
        assert rays.dtype == RAY_DTYPE
        out = np.empty(len(rays), HIT_DTYPE)
        ctr = np.zeros(2, np.uint64)
        self._l.ora_trace_closest(self._h, rays.ctypes.data, len(rays), out.ctypes.data, ctr.ctypes.data, nthreads, int(brute_force))
        return out, {"nodes_visited": int(ctr[0]), "prims_tested": int(ctr[1])}

    def intersect_p(self, rays, nthreads=8):
        assert rays.dtype == RAY_DTYPE
        out = np.empty(len(rays), np.uint8)
        ctr = np.zeros(2, np.uint64)
        self._l.ora_trace_any(self._h, rays.ctypes.data, len(rays), out.ctypes.data, ctr.ctypes.data, nthreads)
        return out, {"nodes_visited": int(ctr[0]), "prims_tested": int(ctr[1])}

    def __del__(self):
        try:
            self._l.ora_scene_destroy(self._h)
        except Exception:
            pass


def wide_trace(node_buf, prim_buf, rays, any_hit=False, nthreads=8, flat=False):
    """Walk the GPU-layout tree on the CPU; returns (hits or occlusion, counters).  flat: sweep all primitives without
    the tree, as the kernels do for scenes with PmAccelInfo.flat_prims != 0."""
    assert rays.dtype == RAY_DTYPE
    out = np.empty(len(rays), np.uint8 if any_hit else HIT_DTYPE)
    ctr = np.zeros(4, np.uint64)
    lib().ora_wide_trace(node_buf.ctypes.data, prim_buf.ctypes.data, rays.ctypes.data, len(rays), out.ctypes.data,
                         int(any_hit), ctr.ctypes.data, nthreads, int(bool(flat)))
    return out, {"nodes_visited": int(ctr[0]), "prims_tested": int(ctr[1]), "exact_tests": int(ctr[2]),
                 "fallback_walks": int(ctr[3])}


class Renderer:
    """Integrate::render() on the CPU.  Takes the flattened scene as ctypes objects laid out like the C-ABI
    structs of include/pbrt_mi355.h (the oracle re-declares the layouts in ref_integrator.h)."""

    def __init__(self, scene, materials, n_materials, prim_materials, lights, n_lights, camera, film):
        L = lib()
        p, u32 = C.c_void_p, C.c_uint32
        L.ora_renderer_create.restype = p
        L.ora_renderer_create.argtypes = [p, p, u32, p, u32, p, u32, p, p]
        L.ora_renderer_destroy.argtypes = [p]
        L.ora_renderer_render.argtypes = [p, p, C.c_int, p]
        L.ora_renderer_clear.argtypes = [p]
        L.ora_renderer_film.argtypes = [p, p]
        L.ora_renderer_image.argtypes = [p, p]
        L.ora_renderer_scene_radius.restype = C.c_float
        L.ora_renderer_scene_radius.argtypes = [p]
        L.ora_renderer_sample.argtypes = [p, p, C.c_int, C.c_int, C.c_int, p]
        L.ora_renderer_set_medium.argtypes = [p, p, p, C.c_float, p]
        self._l = L
        self._scene = scene
        pm = np.ascontiguousarray(prim_materials, dtype=np.uint32)
        self._h = C.c_void_p(L.ora_renderer_create(scene._h, C.addressof(materials), n_materials, pm.ctypes.data, len(pm),
                                                   C.addressof(lights), n_lights, C.addressof(camera), C.addressof(film)))
        b = film.pixel_bounds
        self.width, self.height = b[2] - b[0], b[3] - b[1]

    def set_medium(self, sigma_a, sigma_s, g, bounds):
        """HomogeneousMedium for the VOL_PATH integrator (tables already scaled); bounds = 6 floats"""
        a, s = _f32(sigma_a), _f32(sigma_s)
        b = _f32(bounds)
        self._l.ora_renderer_set_medium(self._h, a.ctypes.data, s.ctypes.data, g, b.ctypes.data)

    def render(self, params, nthreads=8):
        ctr = np.zeros(4, np.uint64)
        self._l.ora_renderer_render(self._h, C.addressof(params), nthreads, ctr.ctypes.data)
        self._l.ora_renderer_dead_final_rays.restype = C.c_uint64
        self._l.ora_renderer_dead_final_rays.argtypes = [C.c_void_p]
        dead = int(self._l.ora_renderer_dead_final_rays(self._h))
        # useful_closest_rays: what the device traces (it elides rays towards a last vertex that cannot add to the film)
        return {"samples": int(ctr[0]), "closest_rays": int(ctr[1]), "shadow_rays": int(ctr[2]), "bounces": int(ctr[3]),
                "dead_final_rays": dead, "useful_closest_rays": int(ctr[1]) - dead}

    def clear(self):
        self._l.ora_renderer_clear(self._h)

    def film(self):
        out = np.empty((self.height, self.width, 4), np.float64)
        self._l.ora_renderer_film(self._h, out.ctypes.data)
        return out

    def image(self):
        out = np.empty((self.height, self.width, 3), np.float32)
        self._l.ora_renderer_image(self._h, out.ctypes.data)
        return out

    def scene_radius(self):
        return float(self._l.ora_renderer_scene_radius(self._h))

    def sample(self, params, px, py, sample_index):
        out = np.zeros(13, np.float32)
        self._l.ora_renderer_sample(self._h, C.addressof(params), px, py, sample_index, out.ctypes.data)
        return out

    def __del__(self):
        try:
            self._l.ora_renderer_destroy(self._h)
        except Exception:
            pass
