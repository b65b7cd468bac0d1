import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")
os.environ["PBRT_MI355_VERIFY"] = "1"
sd = pm.scenes.blob_sky_scene(20480, (3840, 2160))
integ = pm.WavefrontIntegrator(5, True, True, sd, 2, split_method=pm._capi.PM_SPLIT_HLBVH)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    integ.clear(); integ.render(); sys.stderr.flush()
