"""quick GPU probe: times context creation and one closest/any trace (diagnostic, not a test)"""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
pm = importlib.import_module("pbrt-rs_amd")
print("import", time.time() - t0, flush=True)
pos, idx = pm.scenes.blob_mesh(64, 256, seed=1)
t0 = time.time()
agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)])
print("create+build", time.time() - t0, flush=True)
rays = pm.scenes.incoherent_rays(100_000, seed=2)
t0 = time.time(); h = agg.intersect(rays); print("closest", time.time() - t0, agg.last_kernel_ms(), (h["prim_id"] != 0xFFFFFFFF).sum(), flush=True)
sh = pm.scenes.shadow_rays(rays, h)
t0 = time.time(); a = agg.intersect_p(sh); print("any", time.time() - t0, agg.last_kernel_ms(), a.sum(), flush=True)
