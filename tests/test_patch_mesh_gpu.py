"""Bilinear-patch meshes at scale - the reference's only mesh primitive is the BilinearPatchMesh (shapes/bilinear_patch.rs:506-612,
615-707); triangles are this repository's extension.  A 256 x 256 heightfield of non-planar quads with shared vertices (65,536
patches):
  * PM_SPLIT_HLBVH builds its tree ON THE DEVICE (round 3; round 2 sent every patch scene to the host SAH build);
  * the patch-only walk (PM_WALK_PATCH: leaves parked, the reference-arithmetic test in shared flush trips) gives the CPU mirror's
    hits bit for bit - timed and counting kernels, closest and any hit - and the counting kernels its counters;
  * against the reference traversal (the oracle's binary BVH over the same patches): t bit-equal everywhere; id / (u, v) differ only
    where two patches that share an edge are hit at bit-equal t (the tie goes to the lower id here, to the first met there);
  * the host-built tree and round 2's routing (PBRT_MI355_PATCH_WALK=0: the MIXED walk) give the same hits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


def _rays(pm, agg_hits=None):
    prim = pm.scenes.primary_rays(512, 512)
    inc = pm.scenes.incoherent_rays(1 << 18, seed=5)
    return prim, inc


@pytest.mark.parametrize("shape", ["heightfield", "blobs"])
def test_patch_mesh_device_build_and_patch_walk(pm, ora, monkeypatch, shape):
    """heightfield: 256 x 256 quads with shared vertices; blobs: 64 blobs x 1024 small warped patches (C3's shape, deep walks)"""
    pos, idx = pm.scenes.terrain_patch_mesh(256) if shape == "heightfield" else pm.scenes.blob_patch_mesh(64, 1024, seed=4, blob_sigma=0.12, size=0.03)
    assert len(idx) == 65536
    mesh = pm.BilinearPatchMesh(pos, idx)
    agg = pm.GpuAggregate([mesh], split_method=pm._capi.PM_SPLIT_HLBVH)
    assert agg.info.flat_prims == 0 and agg.info.n_prims == 65536
    assert agg.info.build_seconds < 0.5
    host = pm.GpuAggregate([mesh], split_method=pm._capi.PM_SPLIT_SAH)
    prim, inc = _rays(pm)
    hp = agg.intersect(prim)
    assert (hp["prim_id"] != MISS).mean() > 0.2
    sh = pm.scenes.shadow_rays(prim, hp, light=(0.3, 2.5, -2.0))
    rays = np.concatenate([prim, inc, sh])
    nb, pb = agg.node_buffer(), agg.prim_buffer()
    assert len(pb) == 65536 * 64                    # 64-byte patch records, one per leaf
    cpu, cw = ora.wide_trace(nb, pb, rays, nthreads=16)
    occ_cpu, cwa = ora.wide_trace(nb, pb, rays, any_hit=True, nthreads=16)
    for counting in (False, True):
        agg.set_counting(counting)
        agg.reset_counters()
        g = agg.intersect(rays)
        assert np.array_equal(g.view(np.uint32), cpu.view(np.uint32)), f"closest hits differ from the CPU mirror (counting={counting})"
        if counting:
            c = agg.counters()
            for k in ("nodes_visited", "prims_tested", "exact_tests"):
                assert c[k] == cw[k], (k, c[k], cw[k])
            assert c["fallback_walks"] == 0 and c["stack_overflows"] == 0
        assert np.array_equal(agg.intersect_p(rays), occ_cpu)
    agg.set_counting(False)
    gpu = agg.intersect(rays)
    assert np.array_equal(agg.intersect_p(rays).astype(bool), gpu["prim_id"] != MISS)
    # the host-built (SAH) tree and round 2's routing of patch scenes through the MIXED walk: same hits
    assert np.array_equal(host.intersect(rays).view(np.uint32), gpu.view(np.uint32))
    monkeypatch.setenv("PBRT_MI355_PATCH_WALK", "0")
    assert np.array_equal(agg.intersect(rays).view(np.uint32), gpu.view(np.uint32))
    assert np.array_equal(agg.intersect_p(rays).astype(bool), gpu["prim_id"] != MISS)
    monkeypatch.delenv("PBRT_MI355_PATCH_WALK")
    # the reference traversal
    sc = ora.Scene()
    sc.add_patch_mesh(pos, idx, False, False)
    sc.build(ora.SPLIT_MIDDLE, 255)
    ref, _ = sc.intersect(rays, nthreads=16)
    assert np.array_equal(gpu["t"].view(np.uint32), ref["t"].view(np.uint32))
    diff = gpu["prim_id"] != ref["prim_id"]
    assert diff.mean() < 1e-3, diff.mean()          # ties on shared edges only ...
    assert (gpu["prim_id"][diff] < ref["prim_id"][diff]).all()          # ... and they go to the lower id
    same = ~diff
    assert np.array_equal(gpu["u"][same].view(np.uint32), ref["u"][same].view(np.uint32))
    assert np.array_equal(gpu["v"][same].view(np.uint32), ref["v"][same].view(np.uint32))
    occ_ref, _ = sc.intersect_p(rays, nthreads=16)
    assert np.array_equal(agg.intersect_p(rays).astype(np.uint8), occ_ref)
    print(f"[patch-mesh] {shape}, 65,536 patches: device build {agg.info.build_seconds * 1e3:.1f} ms, {cw['nodes_visited'] / len(rays):.1f} nodes and "
          f"{cw['prims_tested'] / len(rays):.2f} patch tests per ray; {int(diff.sum())} shared-edge ties of {len(rays)} rays")


def test_render_over_a_patch_mesh_bitwise(pm, ora):
    """the path tracer over a patch mesh with the patch-only walk (closest + shadow kernels): film contributions bit-equal to the
    oracle's (correctly-rounded mode)"""
    from test_branches_gpu import _bit_equal, _report
    sd = pm.scenes.main_rs_scene((64, 64))
    pos, idx = pm.scenes.terrain_patch_mesh(64, tilt_deg=0.0)
    rfw = sd.render_from_world
    pos = pm.scenes._xf_points(rfw.m, pos * np.float32(6.0) + np.array([0.0, 0.0, -1.5], np.float32))
    sd.prims = [pm.BilinearPatchMesh(pos, idx)]
    sd.prim_materials = [1 + (k % 3) for k in range(len(idx))]
    frac, tot = _bit_equal(pm, ora, sd, 4)
    _report("4096-patch heightfield under the main.rs lights (patch-only walk)", frac, tot)
    assert frac >= 0.9999
