#!/usr/bin/env python3
"""development aid: device LBVH build time (PmAccelInfo.build_seconds) at C3 / C4 sizes, and that hits equal the host tree's"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pm = importlib.import_module("pbrt-rs_amd")
for blobs in [int(a) for a in sys.argv[1:]] or [1024, 20480]:
    pos, idx = pm.scenes.blob_mesh(blobs, 1024, seed=1)
    mesh = pm.TriangleMesh(pos, idx)
    for rep in range(3):
        agg = pm.GpuAggregate([mesh], split_method=pm._capi.PM_SPLIT_HLBVH)
        print(f"{blobs * 1024:9d} tris device build {agg.info.build_seconds * 1e3:8.1f} ms, {agg.info.n_nodes} nodes", flush=True)
        if rep < 2:
            agg.close()
    rays = pm.scenes.primary_rays(1024, 1024)
    h = agg.intersect(rays)
    print("   hit fraction", float((h["prim_id"] != 0xFFFFFFFF).mean()), flush=True)
    agg.close()
