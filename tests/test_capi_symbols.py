"""The C-ABI library loads without a GPU and exports every function include/*.h declares (no compute calls here)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECL = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_\s\*]*?\b(pmh?_[a-z0-9_]+)\s*\(", re.M)


def declared_symbols():
    names = set()
    for h in ("pbrt_mi355.h", "pbrt_mi355_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        names.update(m for m in DECL.findall(text) if m != "pm_ctx")
    return names


def test_headers_declare_the_expected_entry_points():
    names = declared_symbols()
    for must in ("pm_create", "pm_destroy", "pm_build_accel", "pm_trace_closest", "pm_trace_any", "pm_render",
                 "pm_film_read", "pm_film_add", "pm_get_counters", "pmh_perspective_camera", "pmh_rgb_to_coeffs"):
        assert must in names
    assert len(names) >= 40


def test_library_exports_every_declared_symbol(pm):
    lib = C.CDLL(os.path.join(ROOT, "pbrt-rs_amd", "libpbrt_mi355.so"))
    missing = [n for n in sorted(declared_symbols()) if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding table covers the same set
    assert set(pm._capi.SYMBOLS) == declared_symbols()


def test_no_device_gives_a_loud_error(pm):
    """pm_create(device) on a box without that GPU must fail (no CPU fallback); device -1 is the host-only context used
    for tree inspection, which refuses to trace."""
    import numpy as np
    import pytest
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(pm.PmError):
            pm.GpuAggregate([], device_id=0)
    pos = np.array([[0, 0, 1], [1, 0, 1], [0, 1, 1]], np.float32)
    agg = pm.GpuAggregate([pm.TriangleMesh(pos, [[0, 1, 2]])], device_id=-1)
    rays = pm.scenes.primary_rays(8, 8)
    with pytest.raises(pm.PmError):
        agg.intersect(rays)
