"""Dev probe (GPU): closest / any-hit kernel time on the Cornell geometry for random interior rays, flat sweep vs tree."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")
n = 1 << 23
rng = np.random.default_rng(5)
for glass in (True, False):
    sd = pm.scenes.cornell_scene((64, 64), glass_sphere=glass)
    for flat_max in ("64", "0"):
        os.environ["PBRT_MI355_FLAT_MAX"] = flat_max
        agg = pm.GpuAggregate(sd.prims)
        lo, hi = agg.bounds()
        rays = np.zeros(n, dtype=pm.RAY_DTYPE)
        rng = np.random.default_rng(5)
        rays["o"] = (lo + (hi - lo) * (0.02 + 0.96 * rng.random((n, 3)))).astype(np.float32)
        d = rng.normal(size=(n, 3))
        rays["d"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        rays["tmax"] = np.inf
        d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
        d_h = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        d_o = torch.empty(n, dtype=torch.uint8, device="cuda")
        agg.set_counting(True); agg.reset_counters(); agg.intersect(d_r, out=d_h); c = agg.counters(); agg.set_counting(False)
        ms_c, ms_a = [], []
        for _ in range(4):
            agg.intersect(d_r, out=d_h); ms_c.append(agg.last_kernel_ms())
            agg.intersect_p(d_r, out=d_o); ms_a.append(agg.last_kernel_ms())
        print(f"glass={glass} flat_max={flat_max} flat={agg.info.flat_prims} closest {min(ms_c):.3f} ms ({n/min(ms_c)/1e3:.0f} Mrays/s) "
              f"any {min(ms_a):.3f} ms | nodes/ray {c['nodes_visited']/n:.2f} prims/ray {c['prims_tested']/n:.2f} exact/ray {c['exact_tests']/n:.2f}",
              flush=True)
        del agg
