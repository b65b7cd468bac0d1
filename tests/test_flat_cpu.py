"""Scenes of at most 64 primitives are swept flat by the kernels (no tree walk; pm_device.hpp trace_flat).  CPU check of
that logic through its mirror in oracle/wide_walk.h: the flat sweep must return the reference traversal's hits bit for
bit, also for rays aimed exactly at edges and vertices, where the float pre-filter has to stay undecided."""
import numpy as np
import pytest


def _edge_rays(pm, pos, idx, rng, n):
    """rays through points ON triangle edges / vertices / just inside / just outside, from random origins"""
    tri = pos[np.asarray(idx).reshape(-1, 3)]
    k = rng.integers(0, len(tri), n)
    a, b, c = tri[k, 0], tri[k, 1], tri[k, 2]
    kind = rng.integers(0, 4, n)
    s = rng.random((n, 1)).astype(np.float32)
    w = rng.dirichlet([1, 1, 1], n).astype(np.float32)
    target = np.where((kind == 0)[:, None], a + (b - a) * s, 0)                       # on edge ab
    target = np.where((kind == 1)[:, None], a, target)                              # vertex
    target = np.where((kind == 2)[:, None], a * w[:, :1] + b * w[:, 1:2] + c * w[:, 2:], target)     # interior
    out = a + (b - a) * s + (a - c) * np.float32(1e-6)
    target = np.where((kind == 3)[:, None], out, target)                            # a hair outside edge ab
    o = (target + rng.normal(size=(n, 3)) * rng.uniform(0.3, 30.0, (n, 1))).astype(np.float32)
    rays = np.zeros(n, dtype=pm.RAY_DTYPE)
    rays["o"] = o
    rays["d"] = ((target - o) * rng.uniform(0.2, 3.0, (n, 1))).astype(np.float32)
    rays["tmax"] = np.where(rng.random(n) < 0.3, rng.uniform(0.2, 1.5, n), np.inf).astype(np.float32)
    return rays


def _scene(pm, seed):
    rng = np.random.default_rng(seed)
    nt = int(rng.integers(8, 40))
    pos = (rng.normal(size=(nt * 3, 3)) * rng.uniform(0.2, 3.0, (nt * 3, 1))).astype(np.float32)
    idx = np.arange(nt * 3, dtype=np.uint32).reshape(-1, 3)
    prims = [pm.TriangleMesh(pos, idx)]
    quad = np.array([[-3, -3, -2.5], [3, -3, -2.5], [-3, 3, -2.0], [3, 3, -3.0]], np.float32)
    prims.append(pm.BilinearPatchMesh(quad, [[0, 1, 2, 3]]))
    for _ in range(int(rng.integers(1, 5))):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = rng.uniform(-2, 2, 3)
        mi = np.eye(4, dtype=np.float32)
        mi[:3, 3] = -m[:3, 3]
        r = float(rng.uniform(0.3, 1.0))
        prims.append(pm.Sphere(r, -r, r, 360.0, m, mi))
    return prims, pos, idx, rng


def _patch_scene(pm, seed, n_patches=28):
    """mostly bilinear patches (twisted, flat, skinny), plus two spheres: the closest-hit sweep pre-filters every patch by its
    padded box and defers what passes"""
    rng = np.random.default_rng(100 + seed)
    quads = []
    for k in range(n_patches):
        c = rng.uniform(-2.5, 2.5, 3)
        e1 = rng.normal(size=3) * rng.uniform(0.05, 1.5)
        e2 = rng.normal(size=3) * rng.uniform(0.05, 1.5)
        twist = rng.normal(size=3) * (0.0 if k % 3 == 0 else rng.uniform(0.0, 0.5))       # every third patch is flat
        quads.append([c, c + e1, c + e2, c + e1 + e2 + twist])
    pos = np.asarray(quads, np.float32).reshape(-1, 3)
    prims = [pm.BilinearPatchMesh(pos, np.arange(4 * n_patches, dtype=np.uint32).reshape(-1, 4))]
    for _ in range(2):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = rng.uniform(-2, 2, 3)
        mi = np.eye(4, dtype=np.float32)
        mi[:3, 3] = -m[:3, 3]
        prims.append(pm.Sphere(0.5, -0.5, 0.5, 360.0, m, mi))
    return prims, pos.reshape(-1, 4, 3), rng


def _patch_rays(pm, quads, rng, n):
    """rays through points p(u, v) ON the patches - interior, on the boundary (u or v exactly 0 / 1), a hair outside it - from
    random origins, some starting on another patch (t ~ 0 candidates)"""
    k = rng.integers(0, len(quads), n)
    q = quads[k]
    u = rng.random(n).astype(np.float32)
    v = rng.random(n).astype(np.float32)
    kind = rng.integers(0, 4, n)
    u = np.where(kind == 1, np.round(u), u)                                  # on the edge u = 0 / 1
    v = np.where(kind == 2, np.round(v), v)
    u = np.where(kind == 3, np.where(u < 0.5, np.float32(-1e-6), np.float32(1.0 + 1e-6)), u)      # a hair outside
    u, v = u[:, None], v[:, None]
    target = (1 - u) * (1 - v) * q[:, 0] + u * (1 - v) * q[:, 1] + (1 - u) * v * q[:, 2] + u * v * q[:, 3]
    o = (target + rng.normal(size=(n, 3)) * rng.uniform(0.3, 20.0, (n, 1))).astype(np.float32)
    on = rng.random(n) < 0.2                                                 # origins on a patch
    k2 = rng.integers(0, len(quads), n)
    q2 = quads[k2]
    o = np.where(on[:, None], (q2[:, 0] + q2[:, 3]) * np.float32(0.5), o).astype(np.float32)
    rays = np.zeros(n, dtype=pm.RAY_DTYPE)
    rays["o"] = o
    rays["d"] = ((target - o) * rng.uniform(0.2, 3.0, (n, 1))).astype(np.float32)
    rays["tmax"] = np.where(rng.random(n) < 0.3, rng.uniform(0.2, 1.5, n), np.inf).astype(np.float32)
    return rays


@pytest.mark.parametrize("seed", [1, 2])
def test_flat_sweep_of_patches_equals_reference(pm, ora, seed):
    import helpers
    prims, quads, rng = _patch_scene(pm, seed)
    agg = pm.GpuAggregate(prims, device_id=-1)
    assert agg.info.flat_prims == agg.info.n_prims <= 64

    class SD:
        pass
    sd = SD()
    sd.prims = prims
    sc = helpers.oracle_scene_from(sd, ora)
    rays = np.concatenate([_patch_rays(pm, quads, rng, 120_000), pm.scenes.incoherent_rays(40_000, seed=seed, radius=6.0)])
    ref, _ = sc.intersect(rays)
    nb, pb = agg.node_buffer(), agg.prim_buffer()
    flat, fc = ora.wide_trace(nb, pb, rays, flat=True)
    assert 0.3 < (ref["prim_id"] != 0xFFFFFFFF).mean() < 0.99
    same = flat["prim_id"] == ref["prim_id"]
    assert (~same).sum() <= 4                                   # exact-t ties
    assert np.array_equal(flat["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert np.array_equal(flat["u"][same].view(np.uint32), ref["u"][same].view(np.uint32))
    # the confirm path of the box pre-filter (every rejected patch goes through the reference arithmetic as well in a counting
    # walk): no rejection hid a hit
    assert fc["fallback_walks"] == 0
    occ, _ = ora.wide_trace(nb, pb, rays, any_hit=True, flat=True)
    occ_ref, _ = sc.intersect_p(rays)
    assert np.array_equal(occ.astype(bool), occ_ref.astype(bool))


@pytest.mark.parametrize("offset,scale", [(0.0, 1.0), (900.0, 1.0), (0.0, 1e-3), (5e4, 40.0)])
def test_patch_prefilter_has_no_false_rejects_on_grazing_and_far_patches(pm, ora, offset, scale):
    """the pad of patch_prefilter was argued, then fuzzed on one 28-patch scene near the origin.  Here: patches far
    from the origin (coordinates ~ 900, ~ 5e4: the pad and the exact test's epsilon scale with them) and tiny ones, and rays that
    GRAZE them - aimed at points of the surface along directions within a fraction of a degree of the tangent plane, and along
    the patch edges.  The counting walk's confirm path puts every rejected patch through the reference arithmetic; a hit it hid
    would be counted in fallback_walks, and the hits must equal the reference traversal's."""
    import helpers
    rng = np.random.default_rng(int(offset) + int(scale * 1000) + 5)
    base = rng.uniform(-3, 3, size=(24, 1, 3))
    e1 = rng.normal(size=(24, 3)); e2 = rng.normal(size=(24, 3))
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True); e2 -= (e2 * e1).sum(1, keepdims=True) * e1
    e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    n = np.cross(e1, e2)
    corners = np.stack([base[:, 0], base[:, 0] + e1, base[:, 0] + e2, base[:, 0] + e1 + e2 + 0.3 * n * rng.uniform(-1, 1, (24, 1))], axis=1)
    quads = ((corners * scale) + offset).astype(np.float32)
    pos = quads.reshape(-1, 3)
    idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 4)
    prims = [pm.BilinearPatchMesh(pos, idx)]
    agg = pm.GpuAggregate(prims, device_id=-1)
    assert agg.info.flat_prims == 24
    m = 60_000
    k = rng.integers(0, 24, m)
    q = quads[k].astype(np.float64)
    u = rng.random(m); v = rng.random(m)
    edge = rng.random(m) < 0.3
    u = np.where(edge, np.round(u), u)
    target = ((1 - u) * (1 - v))[:, None] * q[:, 0] + (u * (1 - v))[:, None] * q[:, 1] + ((1 - u) * v)[:, None] * q[:, 2] + (u * v)[:, None] * q[:, 3]
    tang = (q[:, 1] - q[:, 0]) * rng.normal(size=(m, 1)) + (q[:, 2] - q[:, 0]) * rng.normal(size=(m, 1))
    tang /= np.linalg.norm(tang, axis=1, keepdims=True)
    nrm = np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    slope = 10.0 ** rng.uniform(-6, -1, (m, 1)) * rng.choice([-1.0, 1.0], (m, 1))          # grazing angle: 1e-6 .. 0.1 rad
    d = tang + slope * nrm
    dist = scale * rng.uniform(0.5, 30.0, (m, 1))
    rays = np.zeros(m, dtype=pm.RAY_DTYPE)
    rays["o"] = (target - d * dist).astype(np.float32)
    rays["d"] = (d * rng.uniform(0.3, 3.0, (m, 1))).astype(np.float32)
    rays["tmax"] = np.inf

    class SD:
        pass
    sd = SD()
    sd.prims = prims
    sc = helpers.oracle_scene_from(sd, ora)
    ref, _ = sc.intersect(rays)
    flat, fc = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, flat=True)
    assert (ref["prim_id"] != 0xFFFFFFFF).mean() > 0.2
    assert fc["fallback_walks"] == 0, f"{fc['fallback_walks']} rays had a patch rejected by the pre-filter that the reference arithmetic hits"
    assert np.array_equal(flat["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert (flat["prim_id"] != ref["prim_id"]).sum() <= 4


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_flat_sweep_equals_reference(pm, ora, seed):
    import helpers
    prims, pos, idx, rng = _scene(pm, seed)
    agg = pm.GpuAggregate(prims, device_id=-1)
    assert agg.info.flat_prims == agg.info.n_prims <= 64

    class SD:
        pass
    sd = SD()
    sd.prims = prims
    sc = helpers.oracle_scene_from(sd, ora)
    rays = np.concatenate([_edge_rays(pm, pos, idx, rng, 150_000), pm.scenes.incoherent_rays(50_000, seed=seed, radius=6.0)])
    ref, _ = sc.intersect(rays)
    nb, pb = agg.node_buffer(), agg.prim_buffer()
    flat, fc = ora.wide_trace(nb, pb, rays, flat=True)
    tree, tc = ora.wide_trace(nb, pb, rays, flat=False)
    assert 0.3 < (ref["prim_id"] != 0xFFFFFFFF).mean() < 0.99
    for name, got in (("flat", flat), ("tree", tree)):
        same = got["prim_id"] == ref["prim_id"]
        assert (~same).sum() <= 4, name                         # exact-t ties (shared edges / vertices)
        assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32)), name
        assert np.array_equal(got["u"][same].view(np.uint32), ref["u"][same].view(np.uint32)), name
    assert fc["nodes_visited"] == 0 and fc["fallback_walks"] == 0 and tc["fallback_walks"] == 0
    # every ray is offered every primitive; the slow rays (third deferred primitive, fifth triangle candidate: these edge-aimed rays have many) stop
    # early and are swept again by the second pass
    full = (len(rays) - _degenerate(rays)) * agg.info.n_prims
    assert abs(fc["prims_tested"] - full) < 0.1 * full
    occ, oc = ora.wide_trace(nb, pb, rays, any_hit=True, flat=True)
    occ_ref, _ = sc.intersect_p(rays)
    assert np.array_equal(occ.astype(bool), occ_ref.astype(bool))
    assert oc["prims_tested"] < fc["prims_tested"]


def _degenerate(rays):
    d = np.abs(rays["d"]).sum(axis=1)
    return int((~((d > 0) & np.isfinite(d) & np.isfinite(np.abs(rays["o"]).sum(axis=1)))).sum())


def test_flat_threshold_and_switch(pm, monkeypatch):
    pos, idx = pm.scenes.blob_mesh(1, 64, seed=3)
    assert pm.GpuAggregate([pm.TriangleMesh(pos, idx)], device_id=-1).info.flat_prims == 64
    pos, idx = pm.scenes.blob_mesh(1, 65, seed=3)
    assert pm.GpuAggregate([pm.TriangleMesh(pos, idx)], device_id=-1).info.flat_prims == 0
    monkeypatch.setenv("PBRT_MI355_FLAT_MAX", "0")
    pos, idx = pm.scenes.blob_mesh(1, 16, seed=3)
    assert pm.GpuAggregate([pm.TriangleMesh(pos, idx)], device_id=-1).info.flat_prims == 0
