"""Development aid: the PCIe-inclusive rate of pm_trace_closest when the caller hands over HOST buffers (C3 primary rays):
the library stages rays (32 B/ray) in and hits (16 B/ray) out.  bench.py's `value` is the device-resident rate."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")
pos, idx = pm.scenes.blob_mesh(1024, 1024, seed=1)
agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)])
rays = pm.scenes.primary_rays(4096, 4096)
hits = np.empty(len(rays), dtype=pm.HIT_DTYPE)
agg.intersect(rays, out=hits)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    agg.intersect(rays, out=hits)
    best = min(best, time.perf_counter() - t0)
print(f"host buffers: {len(rays) / best / 1e6:.0f} Mrays/s ({best * 1e3:.1f} ms for {len(rays)} rays, kernel {agg.last_kernel_ms():.2f} ms); "
      f"{len(rays) * 48 / best / 1e9:.1f} GB/s over PCIe")
