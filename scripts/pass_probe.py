"""probe: is the film of a tree-walk render independent of (a) the run, (b) the cut into passes?  prints differing pixels"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")
blobs = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
res = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 5
sd = pm.scenes.blob_sky_scene(blobs, res)
integ = pm.WavefrontIntegrator(depth, True, True, sd, 2, split_method=pm._capi.PM_SPLIT_HLBVH)
if os.environ.get("PROBE_TRACE_FIRST"):
    agg = integ.aggregate
    prim = pm.scenes.primary_rays(1024, 1024, eye=(0.0, 0.0, 0.0))
    step = os.environ["PROBE_TRACE_FIRST"]
    hp = agg.intersect(prim)
    if "c" in step:
        agg.set_counting(True); agg.reset_counters(); agg.intersect(prim); print(agg.counters()); agg.set_counting(False)
    if "a" in step:
        agg.intersect_p(prim)
    if "b" in step:
        nb, pb = agg.node_buffer(), agg.prim_buffer(); print(len(nb), len(pb))
def run(env=None, counting=False):
    if env: os.environ["PBRT_MI355_PASS_PATHS"] = env
    integ.aggregate.set_counting(counting)
    integ.clear(); st = integ.render()
    os.environ.pop("PBRT_MI355_PASS_PATHS", None)
    return integ.film().copy(), st
def diff(a, b, name):
    d = (a.view(np.uint64) != b.view(np.uint64)).any(axis=-1)
    n = int(d.sum())
    print(f"{name}: {n} of {d.size} pixels differ", end="")
    if n:
        ys, xs = np.nonzero(d)
        rel = np.abs(a[d] - b[d]) / np.maximum(np.abs(b[d]), 1e-30)
        print(f"; max rel diff {rel.max():.3e}; first at (x={xs[0]}, y={ys[0]}): {a[ys[0], xs[0]]} vs {b[ys[0], xs[0]]}")
    else:
        print()
a, st = run(); print("passes", st["passes"], "closest", st["closest_rays"], "shadow", st["shadow_rays"])
b, _ = run(); diff(a, b, "same run twice")
c, st = run(str(1 << 23)); print("passes", st["passes"], "closest", st["closest_rays"], "shadow", st["shadow_rays"]); diff(c, a, "two passes vs one")
d, _ = run(str(1 << 23)); diff(d, c, "two passes twice")
e, st = run(counting=True); print("counting: fallback", st["fallback_walks"]); diff(e, a, "counting kernels vs timed")
f, _ = run(str(1 << 23), counting=True); diff(f, e, "counting: two passes vs one")
