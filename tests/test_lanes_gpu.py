"""Two lanes inside pm_render (round 3): the passes of a frame alternate between two streams with their own path state.  The film
must not depend on it bit for bit - samples are accumulated in sample-index order either way - and neither must the counters
of the counting kernels, the ray counts or the queue lengths.  PBRT_MI355_LANES is read at call time (1: one stream; 2: two
lanes whenever the call has two sample indices; unset: two lanes from 2^22 paths on)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(integ, monkeypatch, lanes, pass_paths=None, counting=False, **kw):
    monkeypatch.setenv("PBRT_MI355_LANES", str(lanes))
    if pass_paths is None:
        monkeypatch.delenv("PBRT_MI355_PASS_PATHS", raising=False)
    else:
        monkeypatch.setenv("PBRT_MI355_PASS_PATHS", str(pass_paths))
    integ.aggregate.set_counting(counting)
    integ.clear()
    st = integ.render(**kw)
    integ.aggregate.set_counting(False)
    return integ.film().copy(), st, integ.queue_counts()


def _scenes(pm):
    C = pm._capi
    yield "main_rs (spheres: MIXED walk + deferred list)", pm.scenes.main_rs_scene((40, 24)), 5, C.PM_INTEGRATOR_SIMPLE_PATH, C.PM_SPLIT_SAH
    yield "C2 (flat sweep, area lights)", pm.scenes.c2_scene((48, 40)), 5, C.PM_INTEGRATOR_SIMPLE_PATH, C.PM_SPLIT_SAH
    yield "Cornell + glass sphere (flat sweep + fix kernels)", pm.scenes.cornell_scene((40, 40)), 5, C.PM_INTEGRATOR_SIMPLE_PATH, C.PM_SPLIT_SAH
    yield "C5 (medium, VolPath)", pm.scenes.c5_scene((32, 32)), 8, C.PM_INTEGRATOR_VOL_PATH, C.PM_SPLIT_SAH
    yield "triangle soup (tree walk, device build)", pm.scenes.blob_sky_scene(8, (48, 27)), 5, C.PM_INTEGRATOR_SIMPLE_PATH, C.PM_SPLIT_HLBVH
    yield "random walk", pm.scenes.main_rs_scene((24, 24)), 4, C.PM_INTEGRATOR_RANDOM_WALK, C.PM_SPLIT_SAH


def test_two_lanes_film_and_counts_bit_identical(pm, monkeypatch):
    for name, sd, depth, integrator, split in _scenes(pm):
        spp = 7         # odd: the last pass of a lane is shorter than the others
        integ = pm.WavefrontIntegrator(depth, True, True, sd, spp, seed=3, integrator=integrator, split_method=split)
        one, st1, q1 = _render(integ, monkeypatch, 1)
        assert st1["passes"] == 1, name
        # two passes (4 + 3 sample indices), then one sample index per pass (7 passes alternating between the lanes)
        for pass_paths in (None, 64):
            two, st2, q2 = _render(integ, monkeypatch, 2, pass_paths)
            assert st2["passes"] == (2 if pass_paths is None else spp), (name, st2["passes"])
            assert np.array_equal(two.view(np.uint64), one.view(np.uint64)), name
            for k in ("samples", "closest_rays", "shadow_rays"):
                assert st2[k] == st1[k], (name, k)
            assert [list(map(int, x)) for x in q2] == [list(map(int, x)) for x in q1], name
        # counting kernels: the same counters whatever the lanes
        _, c1, _ = _render(integ, monkeypatch, 1, counting=True)
        _, c2, _ = _render(integ, monkeypatch, 2, counting=True)
        for k in ("closest_nodes", "closest_prims", "shadow_nodes", "shadow_prims", "exact_tests", "fallback_walks"):
            assert c1[k] == c2[k], (name, k)
        # a ragged pixel window and a sample window, two lanes against one
        win, smp = (3, 2, integ.width - 3, integ.height - 3), (2, 7)
        a, _, _ = _render(integ, monkeypatch, 1, window=win, samples=smp)
        b, _, _ = _render(integ, monkeypatch, 2, window=win, samples=smp)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), name


def test_lane_choice_and_later_calls(pm, monkeypatch):
    """unset: small calls stay on one lane; a call right after a two-lane render (film read, film add, another render) sees all
    of its work"""
    sd = pm.scenes.c2_scene((64, 64))
    integ = pm.WavefrontIntegrator(5, True, True, sd, 8, seed=1)
    monkeypatch.delenv("PBRT_MI355_LANES", raising=False)
    integ.clear()
    assert integ.render()["passes"] == 1          # 64 x 64 x 8 paths: below the threshold
    whole = integ.film().copy()
    monkeypatch.setenv("PBRT_MI355_LANES", "2")
    for _ in range(3):      # the film is read straight after the call, and cleared straight after that
        integ.clear()
        assert integ.render()["passes"] == 2
        assert np.array_equal(integ.film().view(np.uint64), whole.view(np.uint64))
    # one sample index cannot be split
    integ.clear()
    assert integ.render(samples=(0, 1))["passes"] == 1
    # the wave schedule (1, 1, 2, 4 sample indices) through two lanes
    integ.clear()
    integ.render_waves()
    assert np.array_equal(integ.film().view(np.uint64), whole.view(np.uint64))
