"""Direction length must not matter (CPU side: the mirror of the walk and of the flat sweep in oracle/wide_walk.h against the
reference traversal; the GPU test is tests/test_trace_gpu.py::test_direction_scale_invariance).  Round 2 treated a direction
COMPONENT below 1e-18 as zero whatever the direction's length, so a direction of length ~1e-18 - legal: the reference finds its
hits at t ~ 1e18 - missed everything; the threshold is now relative to the largest component (pm_device.hpp safe_rcp_dir)."""
import numpy as np
import pytest

MISS = 0xFFFFFFFF


@pytest.mark.parametrize("flat", [False, True])
def test_mirror_of_the_walk_is_direction_scale_invariant(pm, ora, monkeypatch, flat):
    import helpers
    monkeypatch.setenv("PBRT_MI355_FLAT_MAX", "64" if flat else "0")
    if flat:
        pos, idx = pm.scenes.blob_mesh(1, 40, seed=9, blob_sigma=0.3, tri_size=0.3)
        ppos, pidx = pm.scenes.blob_patch_mesh(1, 10, seed=9, blob_sigma=0.3, size=0.3)
    else:
        pos, idx = pm.scenes.blob_mesh(8, 256, seed=9, blob_sigma=0.15, tri_size=0.06)
        ppos, pidx = pm.scenes.blob_patch_mesh(8, 128, seed=9, blob_sigma=0.15, size=0.06)
    prims = [pm.TriangleMesh(pos, idx), pm.BilinearPatchMesh(ppos, pidx)]
    agg = pm.GpuAggregate(prims, device_id=-1)                  # host-only context: build + export, no device
    assert (agg.info.flat_prims != 0) == flat

    class SD:
        pass
    sd = SD()
    sd.prims = prims
    sc = helpers.oracle_scene_from(sd, ora)
    rays = np.concatenate([pm.scenes.primary_rays(64, 64), pm.scenes.incoherent_rays(20_000, seed=3)])
    nb, pb = agg.node_buffer(), agg.prim_buffer()
    base, bc = ora.wide_trace(nb, pb, rays, flat=flat)
    hit = base["prim_id"] != MISS
    assert hit.sum() > 500
    for k in (-60, -30, 30, 60):
        s = np.float32(2.0 ** k)
        scaled = rays.copy()
        scaled["d"] *= s
        ref, _ = sc.intersect(scaled)
        got, c = ora.wide_trace(nb, pb, scaled, flat=flat)
        assert np.array_equal(got["prim_id"], ref["prim_id"])
        assert np.array_equal(got["t"][hit].view(np.uint32), ref["t"][hit].view(np.uint32))
        # against the unit-length rays: the same primitives at t / s (not bit for bit: the reference's patch test normalises
        # by lengths, so its roundings are not homogeneous in d - hence "vs the oracle on the scaled rays" above is the bar)
        if k > 0:                   # LONG directions lose hits in the reference itself (its t <= epsilon rejection grows with |d|,
            continue                # shapes/bilinear_patch.rs:522-560): equal to the oracle is all there is to check
        assert (got["prim_id"] != base["prim_id"]).mean() < 1e-3
        same = hit & (got["prim_id"] == base["prim_id"])
        rel = np.abs(got["t"][same] * s - base["t"][same]) / base["t"][same]
        assert (rel < 1e-4).mean() > 0.998 and rel.max() < 0.05         # (grazing hits move by more)
        # the walk itself is the same walk: node and primitive counters do not move (but for the ulps of t above)
        assert abs(c["nodes_visited"] - bc["nodes_visited"]) <= 1e-3 * bc["nodes_visited"]
        assert abs(c["prims_tested"] - bc["prims_tested"]) <= 1e-3 * bc["prims_tested"] and c["fallback_walks"] == 0
