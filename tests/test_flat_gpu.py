"""GPU parity of the flat sweep (scenes of at most 64 primitives, pm_device.hpp trace_flat) through the C ABI:
hits against the reference traversal, counters against the CPU mirror of the sweep, and the same scene walked
through its tree (PBRT_MI355_FLAT_MAX=0) for comparison."""
import numpy as np
import pytest

from test_flat_cpu import _edge_rays, _scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2])
def test_flat_and_tree_kernels_equal_reference(pm, ora, monkeypatch, seed):
    import helpers
    prims, pos, idx, rng = _scene(pm, seed)

    class SD:
        pass
    sd = SD()
    sd.prims = prims
    sc = helpers.oracle_scene_from(sd, ora)
    rays = np.concatenate([_edge_rays(pm, pos, idx, rng, 400_000), pm.scenes.incoherent_rays(200_000, seed=seed, radius=6.0)])
    rays["d"][:3] = [[0, 0, 0], [np.nan, 1, 0], [np.inf, 0, 0]]      # degenerate rays: misses
    ref, _ = sc.intersect(rays)
    ref_a, _ = sc.intersect_p(rays)
    for flat_max in ("64", "0"):
        monkeypatch.setenv("PBRT_MI355_FLAT_MAX", flat_max)
        agg = pm.GpuAggregate(prims)
        is_flat = agg.info.flat_prims != 0
        assert is_flat == (flat_max == "64")
        agg.set_counting(True)
        agg.reset_counters()
        gpu = agg.intersect(rays)
        c = agg.counters()
        cpu, cw = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, flat=is_flat)
        # the kernel and its CPU mirror: identical, ties included (same order of exact tests)
        assert np.array_equal(gpu["prim_id"], cpu["prim_id"])
        assert np.array_equal(gpu["t"].view(np.uint32), cpu["t"].view(np.uint32))
        for k in ("nodes_visited", "prims_tested", "exact_tests", "fallback_walks"):
            assert c[k] == cw[k], (flat_max, k, c, cw)
        assert c["fallback_walks"] == 0 and (c["nodes_visited"] == 0) == is_flat
        # the reference traversal: bit-identical t, ids up to exact-t ties
        assert np.array_equal(gpu["t"].view(np.uint32), ref["t"].view(np.uint32))
        assert (gpu["prim_id"] != ref["prim_id"]).sum() <= 4
        agg.reset_counters()
        occ = agg.intersect_p(rays)
        ca = agg.counters()
        occ_cpu, cwa = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, any_hit=True, flat=is_flat)
        assert np.array_equal(occ.astype(bool), ref_a.astype(bool))
        for k in ("nodes_visited", "prims_tested", "exact_tests", "fallback_walks"):
            assert ca[k] == cwa[k], (flat_max, "any", k, ca, cwa)
        agg.set_counting(False)
        del agg


@pytest.mark.parametrize("seed", [1, 2])
def test_flat_kernels_on_a_scene_of_patches(pm, ora, seed):
    """28 bilinear patches + 2 spheres: closest hit through the box pre-filter and the second pass, any hit at once; hits
    against the reference traversal, counters against the CPU mirror."""
    import helpers
    from test_flat_cpu import _patch_rays, _patch_scene
    prims, quads, rng = _patch_scene(pm, seed)

    class SD:
        pass
    sd = SD()
    sd.prims = prims
    sc = helpers.oracle_scene_from(sd, ora)
    rays = np.concatenate([_patch_rays(pm, quads, rng, 300_000), pm.scenes.incoherent_rays(100_000, seed=seed, radius=6.0)])
    ref, _ = sc.intersect(rays)
    ref_a, _ = sc.intersect_p(rays)
    agg = pm.GpuAggregate(prims)
    assert agg.info.flat_prims == agg.info.n_prims
    agg.set_counting(True)
    agg.reset_counters()
    gpu = agg.intersect(rays)
    c = agg.counters()
    cpu, cw = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, flat=True)
    assert np.array_equal(gpu["prim_id"], cpu["prim_id"])
    assert np.array_equal(gpu["t"].view(np.uint32), cpu["t"].view(np.uint32))
    for k in ("prims_tested", "exact_tests", "fallback_walks"):
        assert c[k] == cw[k], (k, c, cw)
    assert np.array_equal(gpu["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert (gpu["prim_id"] != ref["prim_id"]).sum() <= 4
    same = gpu["prim_id"] == ref["prim_id"]
    patch = same & (gpu["prim_id"] < len(quads))                # (u, v) of a patch hit: bit for bit
    assert patch.sum() > 100_000
    assert np.array_equal(gpu["u"][patch].view(np.uint32), ref["u"][patch].view(np.uint32))
    assert np.array_equal(gpu["v"][patch].view(np.uint32), ref["v"][patch].view(np.uint32))
    sph = same & (gpu["prim_id"] >= len(quads)) & (gpu["prim_id"] != 0xFFFFFFFF)      # of a sphere hit: acosf / atan2f, 1e-5
    assert np.allclose(gpu["u"][sph], ref["u"][sph], atol=1e-5) and np.allclose(gpu["v"][sph], ref["v"][sph], atol=1e-5)
    agg.reset_counters()
    occ = agg.intersect_p(rays)
    ca = agg.counters()
    _, cwa = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, any_hit=True, flat=True)
    assert np.array_equal(occ.astype(bool), ref_a.astype(bool))
    for k in ("prims_tested", "exact_tests", "fallback_walks"):
        assert ca[k] == cwa[k], ("any", k, ca, cwa)
    agg.set_counting(False)
    # and without the counters (the kernels the renderer runs)
    assert np.array_equal(agg.intersect(rays)["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert np.array_equal(agg.intersect_p(rays).astype(bool), ref_a.astype(bool))


def test_cornell_film_flat_equals_tree(pm, monkeypatch):
    """The wavefront render of the C2 scene is the same film whether rays sweep the 35 primitives or walk their tree."""
    sd = pm.scenes.cornell_scene((96, 96))
    films = []
    for flat_max in ("64", "0"):
        monkeypatch.setenv("PBRT_MI355_FLAT_MAX", flat_max)
        integ = pm.WavefrontIntegrator(5, True, True, sd, 4, seed=3)
        assert (integ.aggregate.info.flat_prims != 0) == (flat_max == "64")
        integ.clear()
        st = integ.render()
        films.append((integ.film().copy(), st))
        del integ
    (fa, sa), (fb, sb) = films
    assert sa["closest_rays"] == sb["closest_rays"] and sa["shadow_rays"] == sb["shadow_rays"]
    differing = (fa != fb).any(axis=-1).mean()
    assert differing < 1e-3, differing          # exact-t ties along shared edges may pick the other triangle


@pytest.mark.gpu
def test_film_independent_of_pass_size(pm, monkeypatch):
    """pm_render cuts the sample range into passes of at most PBRT_MI355_PASS_PATHS paths; the film (f64 sums in
    sample-index order, film.rs add_sample) and the ray counts must not depend on where the cuts fall."""
    sd = pm.scenes.cornell_scene((64, 64))
    films = []
    monkeypatch.setenv("PBRT_MI355_LANES", "1")      # (two lanes halve the passes: tests/test_lanes_gpu.py)
    for paths in (None, "4096", "12288"):       # one pass; one sample per pass; three samples per pass (8 = 3 + 3 + 2)
        if paths is None:
            monkeypatch.delenv("PBRT_MI355_PASS_PATHS", raising=False)
        else:
            monkeypatch.setenv("PBRT_MI355_PASS_PATHS", paths)
        integ = pm.WavefrontIntegrator(5, True, True, sd, 8, seed=1)
        integ.clear()
        st = integ.render()
        films.append((integ.film().copy(), st))
        del integ
    assert [f[1]["passes"] for f in films] == [1, 8, 3]
    for f, st in films[1:]:
        assert st["closest_rays"] == films[0][1]["closest_rays"] and st["shadow_rays"] == films[0][1]["shadow_rays"]
        assert np.array_equal(f, films[0][0])
