"""Host-side mirror of the reference's aggregate interface over the C ABI.

Reference: `BVHAggregate::new(prims, max_prims_in_node, split_method)` and
`impl Primitive for BVHAggregate { bounds, intersect, intersect_p }`
(src/primitives/aggregates.rs:25-47, 532-655; trait at src/primitives/mod.rs:21-27).
Same names and argument meaning; the only change is granularity: `intersect` / `intersect_p` take
an array of rays (numpy structured array, or a torch CUDA tensor viewed as raw bytes) instead of one.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import HIT_DTYPE, RAY_DTYPE, PmAccelInfo, PmCounters, PmSphere


class PmError(RuntimeError):
    pass


class Sphere:
    """Sphere::builder() arguments (src/shapes/sphere.rs:34-41)."""

    def __init__(self, radius, z_min, z_max, phi_max, render_from_object, object_from_render=None,
                 reverse_orientation=False):
        self.radius, self.z_min, self.z_max, self.phi_max = radius, z_min, z_max, phi_max
        self.render_from_object = np.asarray(render_from_object, dtype=np.float32).reshape(4, 4)
        if object_from_render is None:
            raise ValueError("object_from_render (Transform::inverse) must be supplied by the host")
        self.object_from_render = np.asarray(object_from_render, dtype=np.float32).reshape(4, 4)
        self.reverse_orientation = bool(reverse_orientation)


class BilinearPatchMesh:
    """BilinearPatchMesh::new (src/shapes/bilinear_patch.rs:631-685); positions in render space."""

    def __init__(self, positions, indices, reverse_orientation=False, transform_swaps_handedness=False, normals=None, uv=None):
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 4)
        # one shading normal per vertex, in render space, already negated when reverse_orientation (as the
        # reference's constructor leaves them)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        if self.normals is not None and len(self.normals) != len(self.positions):
            raise ValueError("normals must hold one normal per vertex")
        # one (u, v) texture coordinate per vertex (BilinearPatchMesh.uv), optional
        self.uv = None if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        if self.uv is not None and len(self.uv) != len(self.positions):
            raise ValueError("uv must hold one coordinate pair per vertex")
        self.reverse_orientation = bool(reverse_orientation)
        self.transform_swaps_handedness = bool(transform_swaps_handedness)


class TriangleMesh:
    """Extension (no triangle shape in the reference): each triangle is the degenerate patch (v0,v1,v2,v2)."""

    def __init__(self, positions, indices):
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)


def _ptr(a):
    """(address, keepalive) of a numpy array or a torch tensor."""
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return C.c_void_p(a.ctypes.data), a
    if hasattr(a, "data_ptr"):
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return C.c_void_p(a.data_ptr()), a
    raise TypeError(type(a))


class GpuAggregate:
    """Drop-in for BVHAggregate: built once, immutable afterwards, one per GPU."""

    def __init__(self, prims, max_prims_in_node=4, split_method=_capi.PM_SPLIT_SAH, device_id=0):
        self._lib = _capi.lib()
        h = C.c_void_p()
        rc = self._lib.pm_create(device_id, C.byref(h))
        if rc != _capi.PM_OK:
            raise PmError(f"pm_create(device {device_id}) failed with status {rc} (is a gfx950 GPU visible?)")
        self._h = h
        self.rebuild(prims, max_prims_in_node, split_method, clear=False)

    def rebuild(self, prims, max_prims_in_node=4, split_method=_capi.PM_SPLIT_SAH, clear=True):
        """pm_clear_geometry + pm_add_* + pm_build_accel inside the same context (a new BVHAggregate::new over new
        primitives; materials, lights, camera and film of the context stay)."""
        if clear:
            self._check(self._lib.pm_clear_geometry(self._h))
        self.first_prim_id = []
        first = C.c_uint32()
        for p in prims:
            if isinstance(p, Sphere):
                s = PmSphere(p.radius, p.z_min, p.z_max, p.phi_max)
                s.render_from_object[:] = p.render_from_object.reshape(-1).tolist()
                s.object_from_render[:] = p.object_from_render.reshape(-1).tolist()
                s.reverse_orientation = 1 if p.reverse_orientation else 0
                self._check(self._lib.pm_add_spheres(self._h, C.byref(s), 1, C.byref(first), C.sizeof(s)))
            elif isinstance(p, BilinearPatchMesh):
                self._check(self._lib.pm_add_patch_mesh_uv(
                    self._h, p.positions.ctypes.data, len(p.positions), p.indices.ctypes.data, len(p.indices),
                    None if p.normals is None else p.normals.ctypes.data, None if p.uv is None else p.uv.ctypes.data,
                    int(p.reverse_orientation), int(p.transform_swaps_handedness), C.byref(first)))
            elif isinstance(p, TriangleMesh):
                self._check(self._lib.pm_add_triangle_mesh(
                    self._h, p.positions.ctypes.data, len(p.positions), p.indices.ctypes.data, len(p.indices),
                    C.byref(first)))
            else:
                raise TypeError(f"unknown primitive {type(p)}")
            self.first_prim_id.append(first.value)
        self.info = PmAccelInfo()
        self._check(self._lib.pm_build_accel(self._h, split_method, max_prims_in_node, C.byref(self.info), C.sizeof(self.info)))

    # -- Primitive trait -----------------------------------------------------------------------
    def bounds(self):
        b = np.array(list(self.info.bounds), dtype=np.float32)
        return b[:3].copy(), b[3:].copy()

    def intersect(self, rays, out=None):
        """Closest hit for every ray -> HIT_DTYPE array (prim_id == PM_MISS on a miss)."""
        n = self._count(rays)
        if out is None:
            out = np.empty(n, dtype=HIT_DTYPE)
        rp, k1 = _ptr(rays)
        op, k2 = _ptr(out)
        self._check(self._lib.pm_trace_closest(self._h, rp, n, op))
        return out

    def intersect_noalloc(self, d_rays, d_out):
        """pm_trace_closest_noalloc: closest hit, device buffers only, at most PM_NOALLOC_MAX_RAYS rays, no allocation."""
        n = self._count(d_rays)
        rp, k1 = _ptr(d_rays)
        op, k2 = _ptr(d_out)
        self._check(self._lib.pm_trace_closest_noalloc(self._h, rp, n, op))
        return d_out

    def intersect_p(self, rays, out=None):
        """Any hit for every ray -> uint8 array (1 = occluded)."""
        n = self._count(rays)
        if out is None:
            out = np.empty(n, dtype=np.uint8)
        rp, k1 = _ptr(rays)
        op, k2 = _ptr(out)
        self._check(self._lib.pm_trace_any(self._h, rp, n, op))
        return out

    # -- measurement ---------------------------------------------------------------------------
    def set_counting(self, on):
        self._check(self._lib.pm_set_counting(self._h, int(bool(on))))

    def reset_counters(self):
        self._check(self._lib.pm_reset_counters(self._h))

    def counters(self):
        c = PmCounters()
        self._check(self._lib.pm_get_counters(self._h, C.byref(c), C.sizeof(c)))
        return {"rays": c.rays, "nodes_visited": c.nodes_visited, "prims_tested": c.prims_tested,
                "stack_overflows": c.stack_overflows, "exact_tests": c.exact_tests, "fallback_walks": c.fallback_walks}

    def debug_uniforms(self, px, py, seed, sample_index, dimension, n):
        """first n uniforms of the device's sample stream for (pixel, sample index, dimension) — test hook"""
        out = np.zeros(n, np.float32)
        self._check(self._lib.pm_debug_uniforms(self._h, px, py, seed, sample_index, dimension, n, out.ctypes.data))
        return out

    def debug_math(self, which, x):
        """the device's f64-then-round transcendental `which` (include/pbrt_mi355.h pm_debug_math) at the f32 arguments x — test hook"""
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(len(x), np.float32)
        self._check(self._lib.pm_debug_math(self._h, which, x.ctypes.data, len(x), out.ctypes.data))
        return out

    def debug_sphere_sample(self, prim_id, in8):
        """Sphere::sample_with_context of sphere `prim_id` on the device; rows of in8: ctx_p[3], p_origin[3], u[2]; rows of the result:
        ok, p[3], n[3], pdf (include/pbrt_mi355.h pm_debug_sphere_sample; after a pm_render of the scene) — test hook"""
        in8 = np.ascontiguousarray(in8, np.float32).reshape(-1, 8)
        out = np.zeros((len(in8), 8), np.float32)
        self._check(self._lib.pm_debug_sphere_sample(self._h, prim_id, in8.ctypes.data, len(in8), out.ctypes.data))
        return out

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self._lib.pm_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def node_buffer(self):
        need = C.c_uint64()
        self._check(self._lib.pm_get_node_buffer(self._h, None, 0, C.byref(need)))
        buf = np.empty(need.value, dtype=np.uint8)
        self._check(self._lib.pm_get_node_buffer(self._h, buf.ctypes.data, need.value, None))
        return buf

    def prim_buffer(self):
        need = C.c_uint64()
        self._check(self._lib.pm_get_prim_buffer(self._h, None, 0, C.byref(need)))
        buf = np.empty(max(need.value, 16), dtype=np.uint8)
        self._check(self._lib.pm_get_prim_buffer(self._h, buf.ctypes.data, need.value, None))
        return buf

    # -- plumbing ------------------------------------------------------------------------------
    @staticmethod
    def _count(rays):
        if isinstance(rays, np.ndarray):
            if rays.dtype != RAY_DTYPE:
                raise TypeError("rays must have dtype RAY_DTYPE")
            return rays.shape[0]
        nbytes = rays.numel() * rays.element_size()
        if nbytes % 32:
            raise ValueError("ray tensor size must be a multiple of 32 bytes")
        return nbytes // 32

    def _check(self, rc):
        if rc != _capi.PM_OK:
            msg = self._lib.pm_last_error(self._h)
            raise PmError(f"status {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
