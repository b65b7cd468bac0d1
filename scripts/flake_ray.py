"""debug: the shadow ray of pixel (290,630) sample 0 of the C4 frame, any-hit, in several wave compositions"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
pm = importlib.import_module("pbrt-rs_amd")
import oracle as ora
sd = pm.scenes.blob_sky_scene(20480, (3840, 2160))
agg = pm.GpuAggregate(sd.prims, split_method=pm._capi.PM_SPLIT_HLBVH)
r = np.zeros(1, pm.RAY_DTYPE)
r["o"] = np.array([-0.9118588, 0.44718105, 2.951237], np.float32)
r["d"] = np.array([4.157226, -1.1536916, -0.86167717], np.float32)
r["tmax"] = np.float32(0.9999)
nb, pb = agg.node_buffer(), agg.prim_buffer()
print("mirror any:", ora.wide_trace(nb, pb, r, any_hit=True)[0], "mirror closest:", ora.wide_trace(nb, pb, r)[0])
for counting in (True, False):
    agg.set_counting(counting)
    print("counting" if counting else "timed", "alone any:", agg.intersect_p(r), "closest:", agg.intersect(r))
    for n in (2, 17, 64, 128, 1000):
        b = np.repeat(r, n)
        o = agg.intersect_p(b)
        print("   x%d copies: occluded %d of %d" % (n, int(o.sum()), n))
    other = pm.scenes.incoherent_rays(4096, seed=3)
    other["o"] += np.array([0, 0, 4], np.float32); other["tmax"] = 1.5
    for pos in (0, 1, 63, 64, 100, 2000):
        b = other.copy(); b[pos] = r[0]
        print("   among 4096 incoherent rays at", pos, ":", int(agg.intersect_p(b)[pos]))
# all triangles near: brute force with the mirror's exact test through closest with shrinking tmax
rr = r.copy()
hits = []
for k in range(12):
    h = ora.wide_trace(nb, pb, rr)[0][0]
    if h["prim_id"] == 0xFFFFFFFF: break
    hits.append((int(h["prim_id"]), float(h["t"])))
    rr["tmax"] = h["t"]
print("successive closest hits below 0.9999 (mirror):", hits)
rr = r.copy(); rr["tmax"] = np.inf
h = agg.intersect(rr)[0]; print("GPU closest with tmax=inf:", h)
