"""Development aid: the MIXED walk at C3 scale - 1 M triangles or 1 M patches, with and without 512 small spheres in the scene
(a scene that holds spheres, or several primitive types, runs the MIXED kernels; pure scenes run the TRI / PATCH kernels)."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")


def spheres(n, rng):
    out = []
    for _ in range(n):
        c = rng.uniform(-1.0, 1.0, 3).astype(np.float32)
        r = np.float32(rng.uniform(0.01, 0.05))
        m = np.eye(4, dtype=np.float32); m[:3, 3] = c
        mi = np.eye(4, dtype=np.float32); mi[:3, 3] = -c
        out.append(pm.Sphere(r, -r, r, 360.0, m, mi))
    return out


def rate(prims, label):
    agg = pm.GpuAggregate(prims)
    res = []
    for rname, rays in (("primary", pm.scenes.primary_rays(4096, 4096)), ("incoherent", pm.scenes.incoherent_rays(1 << 24, seed=2))):
        d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
        d_hits = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
        d_occ = torch.empty((len(rays),), dtype=torch.uint8, device="cuda")
        for name, fn in (("closest", lambda: agg.intersect(d_rays, out=d_hits)), ("any", lambda: agg.intersect_p(d_rays, out=d_occ))):
            fn()
            ms = []
            for _ in range(3):
                fn(); ms.append(agg.last_kernel_ms())
            res.append(f"{rname} {name} {len(rays) / np.mean(ms) / 1e3:.0f}")
    print(f"{label}: {agg.info.n_prims} prims, {agg.info.n_nodes} nodes, build {agg.info.build_seconds:.2f} s | Mrays/s: " + ", ".join(res), flush=True)


rng = np.random.default_rng(7)
pos, idx = pm.scenes.blob_mesh(1024, 1024, seed=1)
ppos, pidx = pm.scenes.blob_patch_mesh(1024, 1024, seed=1)
rate([pm.TriangleMesh(pos, idx)], "triangles only")
rate([pm.TriangleMesh(pos, idx)] + spheres(512, rng), "triangles + 512 spheres")
rate([pm.BilinearPatchMesh(ppos, pidx)], "patches only")
rate([pm.BilinearPatchMesh(ppos, pidx)] + spheres(512, rng), "patches + 512 spheres")
rate([pm.TriangleMesh(pos[: len(pos) // 2], idx[: len(idx) // 2]), pm.BilinearPatchMesh(ppos, pidx[: len(pidx) // 2])], "0.5 M triangles + 0.5 M patches")
