import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
pm = importlib.import_module("pbrt-rs_amd")
from test_tie_gpu import A, B, D
rng = np.random.default_rng(200)
fill = (rng.uniform(-1.0, 1.0, (200, 1, 3)) * [1.5, 1.5, 0.3] + [0.0, 0.0, 6.0] + rng.uniform(-0.02, 0.02, (200, 3, 3))).astype(np.float32)
tris = np.concatenate([fill, np.stack([B, A])])
pos = tris.reshape(-1, 3); idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)])
r = np.zeros(1, pm.RAY_DTYPE); r["d"] = D; r["tmax"] = np.inf
print(agg.intersect(r))
