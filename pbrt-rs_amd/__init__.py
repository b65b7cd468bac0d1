"""pbrt-rs_amd — MI355X (gfx950) ray-intersection / wavefront path-tracing core for pbrt-rs.

The directory name is not a Python identifier; import it with
    importlib.import_module("pbrt-rs_amd")
(tests/conftest.py and bench.py do).  The HIP extension (libpbrt_mi355.so) is required:
there is no CPU fallback anywhere in this package.
"""
from . import _capi, distributed, host, pbrt_scene, scenes  # noqa: F401
from ._capi import HIT_DTYPE, PM_MISS, RAY_DTYPE  # noqa: F401
from .aggregate import BilinearPatchMesh, GpuAggregate, PmError, Sphere, TriangleMesh  # noqa: F401
from .integrator import WavefrontIntegrator  # noqa: F401
