"""Scenes that mix primitive types in the SAME space (round 3: one tree over all types, wide nodes whose leaves share a type, the
minority types of a node in a node of their own - pm_build.cpp; the MIXED walk parks triangle and patch leaves and keeps the
reference tests out of line - pm_device.hpp).  Hits must equal the oracle's bit for bit whichever way the tree is built
(PBRT_MI355_UNIFIED_TREE=0: round 2's tree per type under a join node), counters must equal the CPU walk of the same tree, and
every primitive must be reachable (brute-force comparison through the oracle)."""
import numpy as np
import pytest

from test_trace_gpu import _build_both, _compare_closest, MISS

pytestmark = pytest.mark.gpu


def _spec(pm, n_spheres=96):
    pos, idx = pm.scenes.blob_mesh(16, 512, seed=4)                   # 8192 triangles
    ppos, pidx = pm.scenes.blob_patch_mesh(16, 512, seed=4)           # 8192 patches, same blob centres: fully interleaved
    rng = np.random.default_rng(8)
    spec = [("tri", pos, idx), ("patch", ppos, pidx)]
    first_sphere = len(idx) + len(pidx)
    for _ in range(n_spheres):
        c = rng.uniform(-1.0, 1.0, 3).astype(np.float32)
        r = np.float32(rng.uniform(0.02, 0.08))
        m = np.eye(4, dtype=np.float32); m[:3, 3] = c
        mi = np.eye(4, dtype=np.float32); mi[:3, 3] = -c
        spec.append(("sphere", r, m, mi))
    return spec, range(first_sphere, first_sphere + n_spheres)


@pytest.mark.parametrize("unified", ["1", "0"])
def test_interleaved_triangles_patches_spheres(pm, ora, monkeypatch, unified):
    monkeypatch.setenv("PBRT_MI355_UNIFIED_TREE", unified)
    spec, sphere_ids = _spec(pm)
    agg, sc = _build_both(pm, ora, spec)
    assert agg.info.flat_prims == 0
    rays = np.concatenate([pm.scenes.primary_rays(256, 256), pm.scenes.incoherent_rays(120_000, seed=5)])
    ref, _ = sc.intersect(rays)
    hit_types = {("tri" if i < 8192 else "patch" if i < 16384 else "sphere") for i in np.unique(ref["prim_id"]) if i != MISS}
    assert hit_types == {"tri", "patch", "sphere"}
    gpu = agg.intersect(rays)
    _compare_closest(gpu, ref, sphere_ids=tuple(sphere_ids))
    sh = pm.scenes.shadow_rays(rays, ref)
    ref_a, _ = sc.intersect_p(sh)
    assert np.array_equal(agg.intersect_p(sh), ref_a)
    # the counting kernels against the CPU walk of the exported tree
    agg.set_counting(True)
    agg.reset_counters()
    gc = agg.intersect(rays[:60_000])
    c = agg.counters()
    cpu, cw = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays[:60_000])
    assert np.array_equal(gc["prim_id"], cpu["prim_id"]) and np.array_equal(gc["t"].view(np.uint32), cpu["t"].view(np.uint32))
    assert (c["nodes_visited"], c["prims_tested"], c["fallback_walks"]) == (cw["nodes_visited"], cw["prims_tested"], 0)
    agg.set_counting(False)


def test_unified_tree_visits_fewer_nodes_than_a_tree_per_type(pm, ora, monkeypatch):
    spec, _ = _spec(pm, n_spheres=32)
    rays = pm.scenes.incoherent_rays(60_000, seed=6)
    visited = {}
    for unified in ("1", "0"):
        monkeypatch.setenv("PBRT_MI355_UNIFIED_TREE", unified)
        agg, _ = _build_both(pm, ora, spec)
        agg.set_counting(True)
        agg.reset_counters()
        agg.intersect(rays)
        visited[unified] = agg.counters()["nodes_visited"]
    assert visited["1"] < 0.8 * visited["0"], visited
