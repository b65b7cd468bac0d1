#!/usr/bin/env python3
"""development aid: device-side (PM_SPLIT_HLBVH) vs host (binned SAH) build time and traversal cost on C3/C4 geometry"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pm = importlib.import_module("pbrt-rs_amd")
for blobs in [int(a) for a in sys.argv[1:]] or [1024]:
    pos, idx = pm.scenes.blob_mesh(blobs, 1024, seed=1)
    rays = pm.scenes.primary_rays(2048, 2048)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_hits = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
    res = {}
    for name, m, leaf in (("hlbvh_gpu", pm._capi.PM_SPLIT_HLBVH, 4), ("hlbvh_gpu2", pm._capi.PM_SPLIT_HLBVH, 2),
                          ("hlbvh_gpu1", pm._capi.PM_SPLIT_HLBVH, 1), ("sah_host", pm._capi.PM_SPLIT_SAH, 4),
                          ("sah_host2", pm._capi.PM_SPLIT_SAH, 2), ("sah_host1", pm._capi.PM_SPLIT_SAH, 1)):
        t0 = time.time()
        agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)], split_method=m, max_prims_in_node=leaf)
        wall = time.time() - t0
        agg.set_counting(True); agg.reset_counters(); agg.intersect(d_rays, out=d_hits); c = agg.counters(); agg.set_counting(False)
        agg.intersect(d_rays, out=d_hits)
        ms = min(agg.intersect(d_rays, out=d_hits) is None or agg.last_kernel_ms() for _ in range(3))
        res[name] = d_hits.cpu().numpy().copy()
        print(f"{blobs*1024:9d} tris {name:10s} build {agg.info.build_seconds*1e3:8.1f} ms (ctor wall {wall:.2f} s) nodes {agg.info.n_nodes:8d} "
              f"nodes/ray {c['nodes_visited']/len(rays):6.2f} prims/ray {c['prims_tested']/len(rays):5.2f} trace {ms:7.2f} ms "
              f"= {len(rays)/ms/1e3:7.1f} Mrays/s", flush=True)
        del agg
    a, b = res["hlbvh_gpu"].view(np.uint32), res["sah_host"].view(np.uint32)
    print("   t/u/v identical:", bool(np.array_equal(a[:, 1:], b[:, 1:])), " prim ids differing:", int((a[:, 0] != b[:, 0]).sum()), flush=True)
