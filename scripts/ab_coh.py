"""A/B (GPU): per-wave table culling for the camera rays of a flat-swept scene (PBRT_MI355_COH=0|1, read at call time), one lane."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")
os.environ["PBRT_MI355_LANES"] = os.environ.get("LANES", "1")
C = pm._capi
for name, sd, spp, depth, integ in (("c2", pm.scenes.c2_scene((1024, 1024)), 64, 5, C.PM_INTEGRATOR_SIMPLE_PATH),
                                    ("c2pl", pm.scenes.cornell_scene((1024, 1024)), 64, 5, C.PM_INTEGRATOR_SIMPLE_PATH),
                                    ("c5", pm.scenes.c5_scene((1024, 1024)), 8, 16, C.PM_INTEGRATOR_VOL_PATH),
                                    ("main_rs", pm.scenes.main_rs_scene((1024, 1024)), 16, 5, C.PM_INTEGRATOR_SIMPLE_PATH)):
    it = pm.WavefrontIntegrator(depth, True, True, sd, spp, seed=0, integrator=integ)
    ref = None
    for coh in ("0", "1", "0", "1"):
        os.environ["PBRT_MI355_COH"] = coh
        it.clear(); it.render()
        ms = np.zeros(5); tot = 0.0
        for _ in range(4):
            it.clear(); st = it.render(); ms += np.asarray(st["ms_by_kernel"]); tot += st["kernel_ms"]
        film = it.film().copy()
        if ref is None:
            ref = film
        print(f"{name} coh={coh}: {tot / 4:7.2f} ms; by class {[round(x / 4, 2) for x in ms]}; film bit-identical to coh=0: "
              f"{np.array_equal(film.view(np.uint64), ref.view(np.uint64))}", flush=True)
    del it
