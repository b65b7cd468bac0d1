import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
pm = importlib.import_module("pbrt-rs_amd")
import test_coh_gpu as t
bad = 0
for seed in range(100, 500):
    sd = t._cloud_scene(pm, seed, (64, 48), 0.0)
    integ = pm.WavefrontIntegrator(2, True, True, sd, 2, seed=seed)
    films = []
    for coh in ("0", "1"):
        os.environ["PBRT_MI355_COH"] = coh
        integ.clear(); integ.render(); films.append(integ.film().copy())
    if not np.array_equal(films[0].view(np.uint64), films[1].view(np.uint64)):
        bad += 1; print("DIFF seed", seed, int((films[0] != films[1]).any(axis=-1).sum()), flush=True)
    del integ
print("seeds 100..499: differing films", bad)
