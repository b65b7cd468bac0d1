"""The float pre-filter that decides which spheres get the (deferred) interval test may never reject a sphere the
reference's test accepts.  CPU check: the kernel-order walk of the wide tree (oracle/wide_walk.h, which repeats the
pre-filter operation for operation) must give the reference traversal's hits bit for bit on scenes of random spheres -
rotated, non-uniformly scaled, translated far from the origin, partial (z / phi clipped) - and on rays chosen to be
hard: grazing, starting inside, starting on the surface, far away, nearly parallel to an axis."""
import numpy as np
import pytest


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _spheres(pm, rng, n, spread, partial):
    out, centres, radii = [], [], []
    for _ in range(n):
        scale = np.diag(rng.uniform(0.5, 2.0, 3)) if rng.random() < 0.5 else np.eye(3)
        A = _rot(rng) @ scale
        t = rng.uniform(-spread, spread, 3)
        m = np.eye(4)
        m[:3, :3] = A
        m[:3, 3] = t
        r = float(rng.uniform(0.2, 1.5))
        if partial and rng.random() < 0.5:
            zmin, zmax, phi = -r * rng.uniform(0.2, 1.0), r * rng.uniform(0.2, 1.0), float(rng.uniform(90, 360))
        else:
            zmin, zmax, phi = -r, r, 360.0
        out.append(pm.Sphere(r, zmin, zmax, phi, m.astype(np.float32), np.linalg.inv(m).astype(np.float32)))
        centres.append(t)
        radii.append(r * np.linalg.norm(A, 2))
    return out, np.array(centres), np.array(radii)


def _rays(pm, rng, centres, radii, n):
    """rays aimed at random spheres: through the body, grazing the silhouette, from inside, from the surface, from afar"""
    k = rng.integers(0, len(centres), n)
    c, r = centres[k], radii[k][:, None]
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    w = rng.normal(size=(n, 3))
    w -= (w * u).sum(1, keepdims=True) * u
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    kind = rng.integers(0, 5, n)[:, None]
    dist = np.where(kind == 3, rng.uniform(50, 5000, (n, 1)), rng.uniform(1.5, 6.0, (n, 1))) * r
    off = np.where(kind == 1, 1.0 + rng.normal(scale=1e-5, size=(n, 1)), rng.uniform(0, 1.2, (n, 1))) * r   # 1: grazing
    o = c - u * dist + w * off
    o = np.where(kind == 2, c + u * r * rng.uniform(0, 0.9, (n, 1)), o)                                     # 2: inside
    o = np.where(kind == 4, c - u * r, o)                                                                  # 4: on the surface
    d = np.where(kind == 2, rng.normal(size=(n, 3)), u * rng.uniform(0.1, 10.0, (n, 1)))
    d = np.where(kind == 4, rng.normal(size=(n, 3)), d)
    axis = rng.random(n) < 0.05
    d[axis, 1:] *= 1e-7
    rays = np.zeros(n, dtype=pm.RAY_DTYPE)
    rays["o"], rays["d"] = o.astype(np.float32), d.astype(np.float32)
    rays["tmax"] = np.where(rng.random(n) < 0.2, rng.uniform(0.5, 3.0, n), np.inf).astype(np.float32)
    return rays


@pytest.mark.parametrize("flat_max", ["64", "0"])          # flat sweep / tree walk (both defer spheres)
@pytest.mark.parametrize("seed,spread,partial", [(1, 3.0, False), (2, 3.0, True), (3, 300.0, True), (4, 0.5, True)])
def test_sphere_prefilter_never_drops_a_hit(pm, ora, monkeypatch, seed, spread, partial, flat_max):
    import helpers
    monkeypatch.setenv("PBRT_MI355_FLAT_MAX", flat_max)
    rng = np.random.default_rng(seed)
    prims, centres, radii = _spheres(pm, rng, 24, spread, partial)
    pos = np.array([[-4, -4, -2], [4, -4, -2], [4, 4, -2], [-4, 4, -2]], np.float32) * np.float32(max(1.0, spread / 3))
    prims.append(pm.TriangleMesh(pos, [[0, 1, 2], [0, 2, 3]]))       # something that lowers tcull
    agg = pm.GpuAggregate(prims, device_id=-1)
    assert (agg.info.flat_prims != 0) == (flat_max != "0")
    monkeypatch.setenv("PBRT_MI355_FLAT_MAX", "0")                   # single-sphere checkers below walk their tree

    class SD:
        pass
    sd = SD()
    sd.prims = prims
    sc = helpers.oracle_scene_from(sd, ora)
    rays = _rays(pm, rng, centres, radii, 200_000)
    wide, wc = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, flat=agg.info.flat_prims != 0)
    ref, _ = sc.intersect(rays)
    hit = ref["prim_id"] != 0xFFFFFFFF
    assert 0.2 < hit.mean() < 0.98
    # Bit-identical, except where two overlapping spheres are hit within each other's interval width: the reference's
    # test rejects a sphere whose t interval reaches above the current t_max (sphere.rs: t0.upper_bound() > t_max), so
    # which of the two it returns depends on the order its own BVH visits them in.  Every differing ray must be of that
    # kind: the later hit, tested alone with t_max = the earlier hit's t, is rejected by the reference's own test.
    diff = np.nonzero(wide["t"].view(np.uint32) != ref["t"].view(np.uint32))[0]
    assert len(diff) < 5e-3 * len(rays)
    assert hit[diff].all() and (wide["prim_id"][diff] != 0xFFFFFFFF).all()
    assert (wide["prim_id"][diff] != ref["prim_id"][diff]).all()
    single = {}
    for i in diff:
        near, far = (wide[i], ref[i]) if wide["t"][i] < ref["t"][i] else (ref[i], wide[i])
        k = int(near["prim_id"])
        assert k < 24                                                # a sphere
        if k not in single:
            single[k] = pm.GpuAggregate([prims[k]], device_id=-1)
        one = rays[i:i + 1].copy()
        one["tmax"] = far["t"]
        alone, _ = ora.wide_trace(single[k].node_buffer(), single[k].prim_buffer(), one, flat=False)
        assert alone["prim_id"][0] == 0xFFFFFFFF, (i, near, far)
    diff = wide["t"].view(np.uint32) != ref["t"].view(np.uint32)
    same_id = wide["prim_id"] == ref["prim_id"]
    assert np.array_equal(wide["u"][same_id & ~diff].view(np.uint32), ref["u"][same_id & ~diff].view(np.uint32))
    assert wc["fallback_walks"] == 0
    # the pre-filter earns its keep: far fewer interval tests than sphere visits
    assert wc["exact_tests"] < 0.8 * wc["prims_tested"]
    occ, _ = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, any_hit=True, flat=agg.info.flat_prims != 0)
    occ_ref, _ = sc.intersect_p(rays)
    assert np.array_equal(occ.astype(bool), occ_ref.astype(bool))


def test_disjoint_spheres_bit_exact(pm, ora):
    """No two spheres overlap => no order dependence: every hit is the reference's, bit for bit."""
    import helpers
    rng = np.random.default_rng(11)
    prims, centres, radii = [], [], []
    while len(prims) < 20:
        p, c, r = _spheres(pm, rng, 1, 6.0, True)
        if all(np.linalg.norm(c[0] - c2) > r[0] + r2 + 0.05 for c2, r2 in zip(centres, radii)):
            prims += p
            centres.append(c[0])
            radii.append(r[0])
    agg = pm.GpuAggregate(prims, device_id=-1)

    class SD:
        pass
    sd = SD()
    sd.prims = prims
    sc = helpers.oracle_scene_from(sd, ora)
    rays = _rays(pm, rng, np.array(centres), np.array(radii), 300_000)
    wide, wc = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, flat=agg.info.flat_prims != 0)
    ref, _ = sc.intersect(rays)
    for f in ("prim_id", "t", "u", "v"):
        assert np.array_equal(wide[f].view(np.uint32), ref[f].view(np.uint32)), f
    assert wc["exact_tests"] < 0.8 * wc["prims_tested"]


@pytest.mark.gpu
@pytest.mark.parametrize("flat_max", ["64", "0"])
@pytest.mark.parametrize("seed,spread,partial", [(2, 3.0, True), (3, 300.0, True)])
def test_sphere_prefilter_confirm_path_on_the_gpu(pm, ora, monkeypatch, seed, spread, partial, flat_max):
    """Round 3: the counting kernels put every sphere their pre-filter rejects through the interval test all the same (the confirm
    path, as for patches); a rejected sphere that would have been the hit counts as a fallback and the ray is redone exactly.
    On the hard rays of the CPU test: no fallback, counters equal to the mirror's, hits equal to the timed kernels'."""
    monkeypatch.setenv("PBRT_MI355_FLAT_MAX", flat_max)
    rng = np.random.default_rng(seed)
    prims, centres, radii = _spheres(pm, rng, 24, spread, partial)
    pos = np.array([[-4, -4, -2], [4, -4, -2], [4, 4, -2], [-4, 4, -2]], np.float32) * np.float32(max(1.0, spread / 3))
    prims.append(pm.TriangleMesh(pos, [[0, 1, 2], [0, 2, 3]]))
    agg = pm.GpuAggregate(prims)
    is_flat = agg.info.flat_prims != 0
    assert is_flat == (flat_max != "0")
    rays = _rays(pm, rng, centres, radii, 200_000)
    timed, timed_occ = agg.intersect(rays), agg.intersect_p(rays)
    agg.set_counting(True)
    for any_hit in (False, True):
        agg.reset_counters()
        got = agg.intersect_p(rays) if any_hit else agg.intersect(rays)
        c = agg.counters()
        cpu, cw = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, any_hit=any_hit, flat=is_flat)
        for k in ("nodes_visited", "prims_tested", "exact_tests", "fallback_walks"):
            assert c[k] == cw[k], (any_hit, k, c, cw)
        assert c["fallback_walks"] == 0
        if any_hit:
            assert np.array_equal(got.astype(bool), timed_occ.astype(bool)) and np.array_equal(got.astype(bool), cpu.astype(bool))
        else:
            for f in ("prim_id", "t", "u", "v"):
                assert np.array_equal(got[f].view(np.uint32), timed[f].view(np.uint32)), f
            assert np.array_equal(got["t"].view(np.uint32), cpu["t"].view(np.uint32))
    agg.set_counting(False)
