#!/usr/bin/env python3
"""CPU-only check of the wide-tree walk (kernel logic mirrored in oracle/wide_walk.h) against the reference traversal
at C3 scale: hits must be bit-identical; prints per-ray counters (nodes, pre-filter tests, exact tests, fallbacks)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as ora
pm = importlib.import_module("pbrt-rs_amd")
blobs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
pos, idx = pm.scenes.blob_mesh(blobs, 1024, seed=1)
agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)], device_id=-1)
nb, pb = agg.node_buffer(), agg.prim_buffer()
sc = ora.Scene(); sc.add_triangle_mesh(pos, idx); sc.build(ora.SPLIT_SAH, 4)
prim = pm.scenes.primary_rays(res, res)
ref, rc = sc.intersect(prim)
sets = {"primary": prim, "shadow": pm.scenes.shadow_rays(prim, ref), "incoherent": pm.scenes.incoherent_rays(res * res, seed=2)}
for name, rays in sets.items():
    ref, rc = sc.intersect(rays)
    t0 = time.time(); wide, wc = ora.wide_trace(nb, pb, rays); dt = time.time() - t0
    n = len(rays)
    bad_t = int((wide["t"].view(np.uint32) != ref["t"].view(np.uint32)).sum())
    bad_uv = int(((wide["u"].view(np.uint32) != ref["u"].view(np.uint32)) | (wide["v"].view(np.uint32) != ref["v"].view(np.uint32))).sum())
    bad_id = int((wide["prim_id"] != ref["prim_id"]).sum())
    occ, oc = ora.wide_trace(nb, pb, rays, any_hit=True)
    occ_ref, _ = sc.intersect_p(rays)
    bad_occ = int((occ != occ_ref).sum())
    print(f"{name:10s} n={n} hit={float((ref['prim_id']!=0xFFFFFFFF).mean()):.3f} mismatches t={bad_t} uv={bad_uv} id={bad_id} any={bad_occ} | "
          f"closest nodes/ray={wc['nodes_visited']/n:.2f} pre/ray={wc['prims_tested']/n:.2f} exact/ray={wc['exact_tests']/n:.3f} "
          f"fallback={wc['fallback_walks']} | any nodes/ray={oc['nodes_visited']/n:.2f} pre/ray={oc['prims_tested']/n:.2f} "
          f"exact/ray={oc['exact_tests']/n:.3f} fallback={oc['fallback_walks']} | ref nodes/ray={rc['nodes_visited']/n:.1f} ({dt:.1f}s)")
