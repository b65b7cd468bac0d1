#!/usr/bin/env python3
"""CPU check: axis-parallel rays (one or two exactly-zero direction components) through the wide tree give the reference's
hits and do not degrade into visiting whole columns of boxes."""
import importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as ora
pm = importlib.import_module("pbrt-rs_amd")
pos, idx = pm.scenes.blob_mesh(64, 1024, seed=3)
agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)], device_id=-1)
nb, pb = agg.node_buffer(), agg.prim_buffer()
sc = ora.Scene(); sc.add_triangle_mesh(pos, idx); sc.build(ora.SPLIT_MIDDLE, 255)
rng = np.random.default_rng(1); n = 30000
rays = np.zeros(n, ora.RAY_DTYPE); rays["o"] = rng.uniform(-1.2, 1.2, (n, 3)); rays["tmax"] = np.inf
axes = rng.integers(0, 3, n); sign = rng.choice([-1.0, 1.0], n)
rays["d"][np.arange(n), axes] = sign * rng.uniform(0.5, 5, n)
rays["d"][n // 2:, (axes[n // 2:] + 1) % 3] = rng.uniform(-1, 1, n - n // 2)   # half with one zero component only
ref, _ = sc.intersect(rays); w, wc = ora.wide_trace(nb, pb, rays)
print('mismatch t', int((w['t'].view(np.uint32) != ref['t'].view(np.uint32)).sum()), 'id', int((w['prim_id'] != ref['prim_id']).sum()),
      'hit frac', float((ref['prim_id'] != 0xFFFFFFFF).mean()), 'nodes/ray', wc['nodes_visited'] / n, 'of', agg.info.n_nodes)
occ, _ = ora.wide_trace(nb, pb, rays, any_hit=True); oref, _ = sc.intersect_p(rays); print('any mismatch', int((occ != oref).sum()))
