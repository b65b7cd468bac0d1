"""GPU box, debug build (-DPM_DBG_UTIL): lane utilisation of the walk's node steps and flush trips on the C3 ray sets"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")
pos, idx = pm.scenes.blob_mesh(1024, 1024, seed=1)
agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)], split_method=pm._capi.PM_SPLIT_HLBVH)
for name, rays in (("primary", pm.scenes.primary_rays(2048, 2048)), ("incoherent", pm.scenes.incoherent_rays(1 << 22, seed=2))):
    agg.reset_counters()
    before = agg.counters()
    agg.intersect(rays)
    after = agg.counters()
    d = {k: after[k] - before[k] for k in after}
    print(name, d)
    n, trips = d["nodes_visited"], d["prims_tested"]
    print("  node steps per ray %.2f ; node-step lane utilisation %.3f" % (n / len(rays), n / (64.0 * max(1, trips))))
    ex, ft = d.get("exact_tests", 0), d.get("fallback_walks", 0)
    print("  flush trips per 64 rays %.2f ; flush lane utilisation %.3f" % (ft / (len(rays) / 64.0), ex / (64.0 * max(1, ft))))
