"""Development aid (for rocprofv3 --stats): 1 M triangles + 512 spheres, closest and any hit on primary rays."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")
rng = np.random.default_rng(7)
pos, idx = pm.scenes.blob_mesh(1024, 1024, seed=1)
prims = [pm.TriangleMesh(pos, idx)]
for _ in range(512):
    c = rng.uniform(-1.0, 1.0, 3).astype(np.float32); r = np.float32(rng.uniform(0.01, 0.05))
    m = np.eye(4, dtype=np.float32); m[:3, 3] = c
    mi = np.eye(4, dtype=np.float32); mi[:3, 3] = -c
    prims.append(pm.Sphere(r, -r, r, 360.0, m, mi))
agg = pm.GpuAggregate(prims)
rays = pm.scenes.primary_rays(4096, 4096)
d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_hits = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
d_occ = torch.empty((len(rays),), dtype=torch.uint8, device="cuda")
for _ in range(4):
    agg.intersect(d_rays, out=d_hits); agg.intersect_p(d_rays, out=d_occ)
print("done", agg.last_kernel_ms())
