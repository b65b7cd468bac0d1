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
PM_MISS = 0xFFFFFFFF
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
                ("prim_buffer_bytes", C.c_uint64), ("bounds", C.c_float * 6), ("build_seconds", C.c_double)]


class PmCounters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("stack_overflows", C.c_uint64)]


# every symbol include/pbrt_mi355.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "pm_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "pm_destroy": (None, [_P]),
    "pm_last_error": (C.c_char_p, [_P]),
    "pm_add_spheres": (C.c_int, [_P, C.POINTER(PmSphere), C.c_uint32, C.POINTER(C.c_uint32)]),
    "pm_add_patch_mesh": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_int, C.c_int, C.POINTER(C.c_uint32)]),
    "pm_add_triangle_mesh": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "pm_clear_geometry": (C.c_int, [_P]),
    "pm_build_accel": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(PmAccelInfo)]),
    "pm_trace_closest": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "pm_trace_any": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "pm_set_counting": (C.c_int, [_P, C.c_int]),
    "pm_reset_counters": (C.c_int, [_P]),
    "pm_get_counters": (C.c_int, [_P, C.POINTER(PmCounters)]),
    "pm_last_kernel_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "pm_get_node_buffer": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pm_get_prim_buffer": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
}


def load_library(path=LIB_PATH):
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
