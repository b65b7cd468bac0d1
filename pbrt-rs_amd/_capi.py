"""ctypes binding of libpbrt_mi355.so (include/pbrt_mi355.h).

This is the binding a host application adds on its side of the C ABI; the Rust `-sys` crate in
INTEGRATION.md declares exactly the same symbols.  The library is REQUIRED: importing this module
raises if the HIP extension has not been built — there is no CPU fallback in the product path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpbrt_mi355.so")

PM_OK = 0
PM_ERR_INVALID, PM_ERR_HIP, PM_ERR_NO_ACCEL, PM_ERR_UNSUPPORTED, PM_ERR_STACK_OVERFLOW = 1, 2, 3, 4, 5
PM_ABI_VERSION = 3
PM_MISS = 0xFFFFFFFF
PM_NOALLOC_MAX_RAYS = 1 << 18
PM_SPLIT_HLBVH, PM_SPLIT_SAH, PM_SPLIT_MIDDLE, PM_SPLIT_EQUAL_COUNTS = 0, 1, 2, 3

RAY_DTYPE = np.dtype([("o", "<f4", 3), ("tmax", "<f4"), ("d", "<f4", 3), ("time", "<f4")])
HIT_DTYPE = np.dtype([("prim_id", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
assert RAY_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 16


class PmSphere(C.Structure):
    _fields_ = [("radius", C.c_float), ("z_min", C.c_float), ("z_max", C.c_float), ("phi_max_deg", C.c_float),
                ("render_from_object", C.c_float * 16), ("object_from_render", C.c_float * 16),
                ("reverse_orientation", C.c_uint32), ("_pad", C.c_uint32)]


class PmAccelInfo(C.Structure):
    _fields_ = [("n_prims", C.c_uint32), ("n_nodes", C.c_uint32), ("node_bytes", C.c_uint32),
                ("prim_record_bytes", C.c_uint32 * 3), ("node_buffer_bytes", C.c_uint64),
                ("prim_buffer_bytes", C.c_uint64), ("bounds", C.c_float * 6), ("build_seconds", C.c_double),
                ("flat_prims", C.c_uint32), ("reserved", C.c_uint32)]


class PmCounters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("stack_overflows", C.c_uint64), ("exact_tests", C.c_uint64), ("fallback_walks", C.c_uint64)]


class PmSpectrum(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_float), ("b", C.c_float), ("c", C.c_float), ("d", C.c_float),
                ("table", C.POINTER(C.c_float))]


class PmMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("spectrum", PmSpectrum), ("u_roughness", C.c_float), ("v_roughness", C.c_float),
                ("remap_roughness", C.c_int32), ("texture", C.c_int32), ("su", C.c_float), ("sv", C.c_float),
                ("du", C.c_float), ("dv", C.c_float), ("spectrum2", PmSpectrum)]


class PmLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("scale", C.c_float), ("render_from_light", C.c_float * 16),
                ("spectrum", C.POINTER(C.c_float)), ("prim_id", C.c_uint32), ("two_sided", C.c_int32)]


class PmMedium(C.Structure):
    _fields_ = [("sigma_a", C.POINTER(C.c_float)), ("sigma_s", C.POINTER(C.c_float)), ("g", C.c_float),
                ("bounds", C.c_float * 6)]


class PmCamera(C.Structure):
    _fields_ = [("camera_from_raster", C.c_float * 16), ("render_from_camera", C.c_float * 16),
                ("lens_radius", C.c_float), ("focal_distance", C.c_float)]


class PmFilm(C.Structure):
    _fields_ = [("full_resolution", C.c_int32 * 2), ("pixel_bounds", C.c_int32 * 4),
                ("max_component_value", C.c_float), ("imaging_ratio", C.c_float),
                ("output_rgb_from_sensor_rgb", C.c_float * 9), ("sensor_rgb_bar", C.POINTER(C.c_float)),
                ("filter_kind", C.c_int32), ("filter_radius", C.c_float * 2),
                ("filter_nx", C.c_int32), ("filter_ny", C.c_int32), ("filter_vals", C.POINTER(C.c_float))]


class PmRenderParams(C.Structure):
    _fields_ = [("integrator", C.c_int32), ("max_depth", C.c_int32), ("sample_lights", C.c_int32),
                ("sample_bsdf", C.c_int32), ("spp", C.c_int32), ("seed", C.c_uint32), ("window", C.c_int32 * 4),
                ("s0", C.c_int32), ("s1", C.c_int32)]


class PmRenderStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("passes", C.c_uint64), ("kernels_launched", C.c_uint64), ("kernel_ms", C.c_double),
                ("state_bytes_per_path", C.c_double), ("ms_by_kernel", C.c_double * 5),
                ("launches_by_kernel", C.c_uint64 * 5), ("closest_nodes", C.c_uint64), ("closest_prims", C.c_uint64),
                ("shadow_nodes", C.c_uint64), ("shadow_prims", C.c_uint64), ("exact_tests", C.c_uint64),
                ("fallback_walks", C.c_uint64)]


PM_SPEC_CONSTANT, PM_SPEC_DENSE, PM_SPEC_RGB_ALBEDO, PM_SPEC_RGB_UNBOUNDED, PM_SPEC_RGB_ILLUMINANT, PM_SPEC_BLACKBODY = range(6)
PM_MAT_DIFFUSE, PM_MAT_DIELECTRIC = 0, 1
PM_TEX_NONE, PM_TEX_CHECKERBOARD = 0, 1
PM_LIGHT_POINT, PM_LIGHT_DIRECTIONAL, PM_LIGHT_UNIFORM_INFINITE, PM_LIGHT_DIFFUSE_AREA = 0, 1, 2, 3
PM_FILTER_BOX, PM_FILTER_TRIANGLE, PM_FILTER_GAUSSIAN = 0, 1, 2
PM_INTEGRATOR_SIMPLE_PATH, PM_INTEGRATOR_RANDOM_WALK, PM_INTEGRATOR_VOL_PATH = 0, 1, 2

# every symbol include/pbrt_mi355.h and include/pbrt_mi355_host.h declare: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "pm_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "pm_destroy": (None, [_P]),
    "pm_last_error": (C.c_char_p, [_P]),
    "pm_abi_version": (C.c_uint32, []),
    "pm_add_spheres": (C.c_int, [_P, C.POINTER(PmSphere), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]),
    "pm_add_patch_mesh": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_int, C.c_int, C.POINTER(C.c_uint32)]),
    "pm_add_patch_mesh_uv": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_uint32)]),
    "pm_add_triangle_mesh": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "pm_clear_geometry": (C.c_int, [_P]),
    "pm_build_accel": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(PmAccelInfo), C.c_uint32]),
    "pm_trace_closest": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "pm_trace_any": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "pm_trace_closest_noalloc": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "pm_set_counting": (C.c_int, [_P, C.c_int]),
    "pm_reset_counters": (C.c_int, [_P]),
    "pm_get_counters": (C.c_int, [_P, C.POINTER(PmCounters), C.c_uint32]),
    "pm_last_kernel_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "pm_get_node_buffer": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pm_get_prim_buffer": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pm_set_materials": (C.c_int, [_P, C.POINTER(PmMaterial), C.c_uint32, C.c_uint32]),
    "pm_set_prim_materials": (C.c_int, [_P, _P, C.c_uint32]),
    "pm_set_lights": (C.c_int, [_P, C.POINTER(PmLight), C.c_uint32, C.c_uint32]),
    "pm_set_medium": (C.c_int, [_P, C.POINTER(PmMedium), C.c_uint32]),
    "pm_set_camera": (C.c_int, [_P, C.POINTER(PmCamera), C.c_uint32]),
    "pm_set_film": (C.c_int, [_P, C.POINTER(PmFilm), C.c_uint32]),
    "pm_render": (C.c_int, [_P, C.POINTER(PmRenderParams), C.c_uint32]),
    "pm_film_clear": (C.c_int, [_P]),
    "pm_film_read": (C.c_int, [_P, _P]),
    "pm_film_add": (C.c_int, [_P, _P]),
    "pm_film_image": (C.c_int, [_P, _P]),
    "pm_get_render_stats": (C.c_int, [_P, C.POINTER(PmRenderStats), C.c_uint32]),
    "pm_get_queue_counts": (C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "pm_debug_sample": (C.c_int, [_P, C.POINTER(PmRenderParams), C.c_int32, C.c_int32, C.c_int32, _P, C.c_uint32]),
    "pm_debug_uniforms": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32, _P]),
    "pm_debug_math": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P]),
    "pm_debug_sphere_sample": (C.c_int, [_P, C.c_uint32, _P, C.c_int32, _P]),
    "pm_debug_medium_state": (C.c_int, [_P, C.POINTER(C.c_float)]),
    # include/pbrt_mi355_host.h
    "pmh_mat4_inverse": (C.c_int, [_P, _P]),
    "pmh_mat4_mul": (None, [_P, _P, _P]),
    "pmh_look_at": (C.c_int, [_P, _P, _P, _P, _P]),
    "pmh_perspective_camera": (C.c_int, [_P, _P, C.c_float, _P, C.c_int, C.c_int, C.c_float, C.c_float,
                                         C.POINTER(PmCamera), _P, _P]),
    "pmh_gaussian_filter_table": (C.c_int, [C.c_float, C.c_float, C.c_float, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pmh_gaussian": (C.c_float, [C.c_float, C.c_float, C.c_float]),
    "pmh_fast_exp": (C.c_float, [C.c_float]),
    "pmh_d65_table": (None, [_P, _P, C.c_int, _P]),
    "pmh_piecewise_linear_table": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "pmh_photometric": (C.c_float, [_P, _P]),
    "pmh_blackbody_table": (C.c_float, [C.c_float, _P]),
    "pmh_rgb_spectrum_table": (None, [_P, C.c_float, _P, _P]),
    "pmh_sigmoid_poly": (C.c_float, [_P, C.c_float]),
    "pmh_rgb_to_coeffs": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "pmh_srgb_matrices": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "pmh_write_exr": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _P]),
    "pmh_rgb_table_generate": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int]),
    "pmh_rgb_table_lookup": (None, [_P, _P, _P, _P]),
}


def load_library(path=None):
    path = path or os.environ.get("PBRT_MI355_LIB") or LIB_PATH     # override: development builds of the same ABI
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C pbrt-rs_amd/csrc).  There is no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process do not coexist
    # ("No HIP GPUs are available" from whichever initialises second).  Loading torch first makes the
    # dynamic loader bind this library to the same runtime torch uses.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib
