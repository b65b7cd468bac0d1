"""pm_trace_closest_noalloc (include/pbrt_mi355.h): the allocation-free variant of pm_trace_closest, through the C ABI.

It launches the same kernels with their scratch (work counters, deferred-primitive and slow-ray lists) in the library's
file-scope device array instead of hipMalloc'ed buffers, so its output must be BIT-IDENTICAL to pm_trace_closest's
(integer / float bits of prim_id, t, u, v) — on two calls in a row, the second with a smaller input (the scratch keeps the
first call's contents), for each kernel family: wide-BVH walk of triangles, the MIXED walk (spheres deferred to the
second pass), and the flat sweep of a tiny scene (deferred + slow lists).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scenes(pm):
    rng = np.random.default_rng(11)
    pos, idx = pm.scenes.blob_mesh(32, 512, seed=4, blob_sigma=0.25, tri_size=0.05)
    tri_scene = [pm.TriangleMesh(pos, idx)]

    def sphere(r, c):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = c
        mi = np.eye(4, dtype=np.float32)
        mi[:3, 3] = -np.asarray(c, np.float32)
        return pm.Sphere(r, -r, r, 360.0, m, mi)

    mixed = [pm.TriangleMesh(pos, idx)] + [sphere(float(rng.uniform(0.05, 0.3)), rng.uniform(-1, 1, 3).astype(np.float32))
                                           for _ in range(80)]
    quad = np.array([[-2, -2, 1.5], [2, -2, 1.5], [-2, 2, 1.5], [2, 2, 1.5]], np.float32)
    tiny_pos, tiny_idx = pm.scenes.blob_mesh(1, 20, seed=9, blob_sigma=0.5, tri_size=0.4)
    tiny = [pm.TriangleMesh(tiny_pos, tiny_idx), sphere(0.5, (0.2, 0.1, 0.0)), sphere(0.4, (-0.5, 0.3, 0.4)),
            pm.BilinearPatchMesh(quad, [[0, 1, 2, 3]])]
    return {"triangles (tree walk)": tri_scene, "triangles + spheres (MIXED walk, deferred list)": mixed,
            "tiny scene (flat sweep, deferred + slow lists)": tiny}


def test_noalloc_matches_trace_closest_twice(pm):
    import torch
    rays_big = pm.scenes.primary_rays(512, 512)                 # 262144 = PM_NOALLOC_MAX_RAYS
    assert len(rays_big) == pm._capi.PM_NOALLOC_MAX_RAYS
    rays_small = pm.scenes.incoherent_rays(70_001, seed=5)      # ragged, smaller second call
    for name, prims in _scenes(pm).items():
        agg = pm.GpuAggregate(prims)
        if "flat" in name:
            assert agg.info.flat_prims > 0
        else:
            assert agg.info.flat_prims == 0
        for rays in (rays_big, rays_small):
            d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
            ref = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
            out = torch.full((len(rays), 4), -7.0, dtype=torch.float32, device="cuda")
            agg.intersect(d_rays, out=ref)
            agg.intersect_noalloc(d_rays, out)
            a, b = ref.cpu().numpy().view(np.uint32), out.cpu().numpy().view(np.uint32)
            assert (a[:, 0] != 0xFFFFFFFF).sum() > 100, name
            assert np.array_equal(a, b), f"{name}: {(a != b).any(axis=1).sum()} of {len(rays)} hits differ"
        agg.close()


def test_noalloc_limits(pm):
    import torch
    pos, idx = pm.scenes.blob_mesh(4, 256, seed=4, blob_sigma=0.25, tri_size=0.05)
    agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)])
    n = pm._capi.PM_NOALLOC_MAX_RAYS + 1
    d_rays = torch.zeros((n, 8), dtype=torch.float32, device="cuda")
    d_out = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(pm.PmError, match="PM_NOALLOC_MAX_RAYS"):
        agg.intersect_noalloc(d_rays, d_out)
    host = pm.scenes.primary_rays(64, 64)
    with pytest.raises(pm.PmError, match="device memory"):
        agg.intersect_noalloc(host, np.empty(len(host), pm.HIT_DTYPE))
    # n == 0 is a no-op
    agg.intersect_noalloc(d_rays[:0], d_out[:0])
