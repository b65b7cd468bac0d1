"""Dev probe (GPU): one build of the library (PBRT_MI355_LIB) on the C2 frame - per-class times on one lane - and the film of a
small C2 / C5 / glass-sphere render saved for comparison with another build (FILM_OUT / FILM_REF)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")
os.environ["PBRT_MI355_LANES"] = "1"
tag = os.environ.get("TAG", "lib")
films = {}
for name, sd, depth, integ in (("c2", pm.scenes.c2_scene((96, 96)), 5, pm._capi.PM_INTEGRATOR_SIMPLE_PATH),
                               ("c2pl", pm.scenes.cornell_scene((96, 96)), 5, pm._capi.PM_INTEGRATOR_SIMPLE_PATH),
                               ("c5", pm.scenes.c5_scene((64, 64)), 16, pm._capi.PM_INTEGRATOR_VOL_PATH)):
    it = pm.WavefrontIntegrator(depth, True, True, sd, 4, seed=3, integrator=integ)
    it.clear(); it.render()
    a = it.film().copy()
    it.clear(); it.render()
    b = it.film().copy()
    films[name] = a
    print(f"[{tag}] {name}: run-to-run identical {np.array_equal(a.view(np.uint64), b.view(np.uint64))}", flush=True)
    del it
if os.environ.get("FILM_OUT"):
    np.savez(os.environ["FILM_OUT"], **films)
if os.environ.get("FILM_REF"):
    ref = np.load(os.environ["FILM_REF"])
    for k, v in films.items():
        d = (v != ref[k]).any(axis=-1)
        print(f"[{tag}] {k}: {int(d.sum())} of {d.size} pixels differ from the reference build", flush=True)
if os.environ.get("TIME", "1") == "1":
    it = pm.WavefrontIntegrator(5, True, True, pm.scenes.c2_scene((1024, 1024)), 64, seed=0)
    for _ in range(2):
        it.clear(); it.render()
    ms = np.zeros(5); tot = 0.0
    for _ in range(5):
        it.clear(); st = it.render(); ms += np.asarray(st["ms_by_kernel"]); tot += st["kernel_ms"]
    print(f"[{tag}] C2 1024x1024@64 one lane: {tot / 5:.2f} ms; by class {[round(x / 5, 2) for x in ms]}", flush=True)
