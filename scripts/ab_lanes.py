"""A/B (GPU): one lane against two lanes inside pm_render (PBRT_MI355_LANES, _STAGGER, _LANE_GRID_DIV; all read at call time).
SCENE=c2|c2pl|c5|c4s (c4s: the C4-shaped soup at 1 M triangles, 1920x1080)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")

scene = os.environ.get("SCENE", "c2")
steps = int(os.environ.get("STEPS", "6"))
C = pm._capi
if scene == "c4s":
    sd = pm.scenes.blob_sky_scene(1024, (1920, 1080)); spp, depth, integ, split = int(os.environ.get("SPP", "16")), 5, C.PM_INTEGRATOR_SIMPLE_PATH, C.PM_SPLIT_HLBVH
elif scene == "c4":
    sd = pm.scenes.blob_sky_scene(20480, (3840, 2160)); spp, depth, integ, split = int(os.environ.get("SPP", "8")), 5, C.PM_INTEGRATOR_SIMPLE_PATH, C.PM_SPLIT_HLBVH
elif scene == "c5":
    sd = pm.scenes.c5_scene((1024, 1024)); spp, depth, integ, split = int(os.environ.get("SPP", "64")), 16, C.PM_INTEGRATOR_VOL_PATH, C.PM_SPLIT_SAH
elif scene == "c2pl":
    sd = pm.scenes.cornell_scene((1024, 1024)); spp, depth, integ, split = int(os.environ.get("SPP", "64")), 5, C.PM_INTEGRATOR_SIMPLE_PATH, C.PM_SPLIT_SAH
else:
    sd = pm.scenes.c2_scene((1024, 1024)); spp, depth, integ, split = int(os.environ.get("SPP", "64")), 5, C.PM_INTEGRATOR_SIMPLE_PATH, C.PM_SPLIT_SAH
it = pm.WavefrontIntegrator(depth, True, True, sd, spp, seed=0, integrator=integ, split_method=split)

configs = [("1", None, None)]
for stg in os.environ.get("STAGGERS", "0,2,3").split(","):
    for div in os.environ.get("DIVS", "1,2").split(","):
        configs.append(("2", stg, div))
ref_film = None
for lanes, stg, div in configs:
    os.environ["PBRT_MI355_LANES"] = lanes
    for k, v in (("PBRT_MI355_STAGGER", stg), ("PBRT_MI355_LANE_GRID_DIV", div)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    for _ in range(2):
        it.clear(); it.render()
    t0 = time.perf_counter()
    for _ in range(steps):
        it.clear(); st = it.render()
    dt = (time.perf_counter() - t0) / steps
    film = it.film()
    film = film.cpu().numpy() if hasattr(film, "cpu") else np.asarray(film)
    if ref_film is None:
        ref_film = film.copy()
    same = bool(np.array_equal(film.view(np.uint64), ref_film.view(np.uint64)))
    print(f"{scene} lanes={lanes} stagger={stg} grid_div={div}: {dt * 1e3:8.2f} ms wall, kernel_ms {st['kernel_ms']:8.2f}, passes {st['passes']}, "
          f"by class {[round(x, 2) for x in st['ms_by_kernel']]}, film bit-identical to one lane: {same}", flush=True)
