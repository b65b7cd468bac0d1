#!/usr/bin/env python3
"""GPU box: the film with PBRT_MI355_REORDER on must be bit-identical to the film without (hits do not depend on walk order)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pm = importlib.import_module("pbrt-rs_amd")
films = []
for env in (None, "6"):
    if env: os.environ["PBRT_MI355_REORDER"] = env
    else: os.environ.pop("PBRT_MI355_REORDER", None)
    sc = pm.scenes.blob_sky_scene(64, (512, 512))
    integ = pm.WavefrontIntegrator(5, True, True, sc, 4)
    integ.render()
    films.append(integ.film().copy())
same = np.array_equal(films[0].view(np.uint64), films[1].view(np.uint64))
print("reorder parity: film bit-identical =", same)
sys.exit(0 if same else 1)
