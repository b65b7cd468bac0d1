"""probe: K timed renders of the C4 scene against the counting kernels' film (deterministic, = the CPU mirror's walk): pixels
that differ.  usage: determinism_probe.py [K] [blobs]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
blobs = int(sys.argv[2]) if len(sys.argv) > 2 else 20480
sd = pm.scenes.blob_sky_scene(blobs, (3840, 2160))
integ = pm.WavefrontIntegrator(5, True, True, sd, 2, split_method=pm._capi.PM_SPLIT_HLBVH)
integ.aggregate.set_counting(True)
integ.clear(); integ.render(); ref = integ.film().copy()
integ.clear(); integ.render(); assert np.array_equal(ref, integ.film()), "counting kernels are not deterministic"
integ.aggregate.set_counting(False)
bad = 0
for k in range(K):
    integ.clear(); st = integ.render()
    d = (integ.film().view(np.uint64) != ref.view(np.uint64)).any(axis=-1)
    bad += int(d.sum())
    ys, xs = np.nonzero(d)
    print(f"run {k}: {int(d.sum())} pixels differ {list(zip(xs.tolist(), ys.tolist()))[:4]} closest {st['closest_rays']}")
print("TOTAL differing pixels over", K, "runs:", bad, "lib", os.environ.get("PBRT_MI355_LIB", "current"))
