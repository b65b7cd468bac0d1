"""GPU parity of Integrate::render(): the wavefront path tracer through the C ABI vs the CPU oracle, same scene,
same spp, same seed (identical PCG32 sample streams by construction).

Tolerances (stated here as BASELINE.json asks):
  * per-sample: wavelengths / wavelength pdfs / filter weight of a sample agree to 2e-6 relative (they involve only
    atanh/cosh, evaluated in f64 on the GPU and by glibc's f32 routines on the CPU);
  * per-sample radiance: >= 99% of the sampled (pixel, sample) pairs agree to 1e-3 relative.  The rest are paths on
    which a last-ulp difference in sin/cos flips a discrete outcome (a hit becomes a miss at a silhouette, a Fresnel
    coin flips): same expectation, different realisation;
  * image: per-pixel relative RMSE between the GPU and the oracle images <= 2e-2 at 16 spp (the rare flipped paths
    above dominate it; the mean image level agrees to 2e-3);
  * splitting a render into pixel windows or sample windows changes NOTHING: films are bit-identical (f64).
"""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _pair(pm, ora, sd, spp, **kw):
    integ = pm.WavefrontIntegrator(kw.get("max_depth", 5), kw.get("sample_lights", True), kw.get("sample_bsdf", True), sd, spp,
                                   seed=kw.get("seed", 0))
    oren = helpers.oracle_renderer_from(sd, ora)
    params = helpers.render_params(pm, sd, spp, kw.get("max_depth", 5), kw.get("sample_lights", True), kw.get("sample_bsdf", True),
                                   kw.get("seed", 0))
    return integ, oren, params


def _check_samples(integ, oren, params, res, spp, n=400, seed=1):
    rng = np.random.default_rng(seed)
    ok = 0
    for _ in range(n):
        px, py, s = int(rng.integers(res[0])), int(rng.integers(res[1])), int(rng.integers(spp))
        g = integ.debug_sample(px, py, s)
        o = oren.sample(params, px, py, s)
        np.testing.assert_allclose(g[4:8], o[4:8], rtol=2e-6)                   # lambda
        if np.all(o[8:12] == g[8:12]) or np.allclose(g[8:12], o[8:12], rtol=2e-5):
            pass
        else:
            raise AssertionError(f"wavelength pdf mismatch {g[8:12]} vs {o[8:12]}")
        np.testing.assert_allclose(g[12], o[12], rtol=1e-5)                      # filter weight
        if np.allclose(g[:4], o[:4], rtol=1e-3, atol=1e-7):
            ok += 1
    return ok / n


def test_main_rs_scene_image_and_samples(pm, ora):
    res, spp = (96, 96), 16
    sd = pm.scenes.main_rs_scene(res)
    integ, oren, params = _pair(pm, ora, sd, spp)
    frac = _check_samples(integ, oren, params, res, spp)
    assert frac >= 0.99, f"only {frac:.3f} of the samples agree"
    st = integ.render()
    oc = oren.render(params, 8)
    assert st["samples"] == oc["samples"] == res[0] * res[1] * spp
    # ray counts agree up to the rare flipped paths
    assert abs(st["closest_rays"] - oc["useful_closest_rays"]) <= 2e-3 * oc["closest_rays"]
    assert abs(st["shadow_rays"] - oc["shadow_rays"]) <= 2e-3 * oc["shadow_rays"]
    gi, oi = integ.image(), oren.image()
    assert np.isfinite(gi).all()
    assert abs(gi.mean() - oi.mean()) <= 2e-3 * abs(oi.mean())
    r = helpers.rel_rmse(gi, oi)
    assert r <= 2e-2, f"relative RMSE {r}"
    # the weight sums depend only on the filter samples: must agree to f32 rounding of each weight
    gf, of = integ.film(), oren.film()
    np.testing.assert_allclose(gf[..., 3], of[..., 3], rtol=1e-5)


def test_cornell_scene_image(pm, ora):
    res, spp = (64, 64), 16
    sd = pm.scenes.cornell_scene(res)
    integ, oren, params = _pair(pm, ora, sd, spp)
    frac = _check_samples(integ, oren, params, res, spp, n=300, seed=5)
    assert frac >= 0.99, f"only {frac:.3f} of the samples agree"
    integ.render()
    oren.render(params, 8)
    gi, oi = integ.image(), oren.image()
    assert abs(gi.mean() - oi.mean()) <= 3e-3 * abs(oi.mean())
    assert helpers.rel_rmse(gi, oi) <= 3e-2


def test_no_light_sampling_and_uniform_bsdf_sampling(pm, ora):
    """the other two branches of simple_path.rs: sample_lights = false, sample_bsdf = false"""
    res, spp = (48, 48), 8
    sd = pm.scenes.main_rs_scene(res)
    for sl, sb in ((False, True), (True, False)):
        integ, oren, params = _pair(pm, ora, sd, spp, sample_lights=sl, sample_bsdf=sb, max_depth=3)
        frac = _check_samples(integ, oren, params, res, spp, n=200, seed=7)
        assert frac >= 0.985, (sl, sb, frac)


def test_patch_mesh_with_vertex_normals(pm, ora):
    """shading normals: lerped vertex normals, Transform::rotate_from_to of dpdu, geometric normal flipped to the shading
    side (bilinear_patch.rs:150-164, interaction.rs:134-158)"""
    res, spp = (64, 64), 8
    sd = pm.scenes.shading_normal_scene(res)
    integ, oren, params = _pair(pm, ora, sd, spp)
    frac = _check_samples(integ, oren, params, res, spp, n=400, seed=11)
    assert frac >= 0.99, f"only {frac:.3f} of the samples agree"
    integ.render()
    oren.render(params, 8)
    gi, oi = integ.image(), oren.image()
    assert abs(gi.mean() - oi.mean()) <= 3e-3 * abs(oi.mean())
    assert helpers.rel_rmse(gi, oi) <= 3e-2
    # and the normals matter: the same mesh without them renders a different image
    flat = pm.scenes.shading_normal_scene(res)
    flat.prims[1] = pm.BilinearPatchMesh(flat.prims[1].positions, flat.prims[1].indices)
    i2 = pm.WavefrontIntegrator(5, True, True, flat, spp)
    i2.render()
    assert helpers.rel_rmse(i2.image(), gi) > 5e-2


def test_c4_shaped_scene_small(pm, ora):
    """BASELINE.json configs[3] (blobs of triangles under a uniform sky) at a size the oracle finishes in seconds:
    16 blobs x 1024 triangles, 96x54 @ 4 spp.  Deep tree, long-tailed walks, infinite-light NEE."""
    res, spp = (96, 54), 4
    sd = pm.scenes.blob_sky_scene(16, res, blob_sigma=0.25, tri_size=0.06)
    integ, oren, params = _pair(pm, ora, sd, spp)
    frac = _check_samples(integ, oren, params, res, spp, n=300, seed=2)
    assert frac >= 0.99, f"only {frac:.3f} of the samples agree"
    st = integ.render()
    oc = oren.render(params, 8)
    assert abs(st["closest_rays"] - oc["useful_closest_rays"]) <= 3e-3 * oc["closest_rays"]
    gi, oi = integ.image(), oren.image()
    assert abs(gi.mean() - oi.mean()) <= 3e-3 * abs(oi.mean())
    assert helpers.rel_rmse(gi, oi) <= 3e-2


def test_random_walk_integrator(pm, ora):
    """RandomWalkIntegrator (integrators/random_walk.rs:106-173): uniform-sphere bounces, infinite lights on a miss, no
    next-event estimation.  The GPU unrolls the recursion but folds the result in the recursion's own order, so a sample
    differs from the oracle only through sin/cos in sample_uniform_sphere (same tolerances as the path integrator)."""
    res, spp, depth = (48, 48), 16, 4
    sd = pm.scenes.main_rs_scene(res)
    RW = pm._capi.PM_INTEGRATOR_RANDOM_WALK
    integ = pm.WavefrontIntegrator(depth, True, True, sd, spp, integrator=RW)
    oren = helpers.oracle_renderer_from(sd, ora)
    params = helpers.render_params(pm, sd, spp, depth, integrator=RW)
    frac = _check_samples(integ, oren, params, res, spp, n=300, seed=3)
    assert frac >= 0.99, f"only {frac:.3f} of the samples agree"
    st = integ.render()
    oc = oren.render(params, 8)
    assert st["shadow_rays"] == oc["shadow_rays"] == 0
    assert abs(st["closest_rays"] - oc["useful_closest_rays"]) <= 2e-3 * oc["closest_rays"]
    gi, oi = integ.image(), oren.image()
    assert np.isfinite(gi).all() and gi.mean() > 0
    assert abs(gi.mean() - oi.mean()) <= 5e-3 * abs(oi.mean())
    assert helpers.rel_rmse(gi, oi) <= 3e-2
    # depth 0: only directly visible infinite lights
    integ0 = pm.WavefrontIntegrator(0, True, True, sd, 4, integrator=RW)
    integ0.render()
    oren.clear()
    oren.render(helpers.render_params(pm, sd, 4, 0, integrator=RW), 4)
    np.testing.assert_allclose(integ0.image(), oren.image(), rtol=2e-5, atol=1e-7)


def test_window_and_sample_split_are_bit_identical(pm, ora):
    res, spp = (40, 24), 8       # deliberately not a multiple of the 8x8 tile in y? (24 is; 40 is) -> also use odd window
    sd = pm.scenes.main_rs_scene(res)
    integ = pm.WavefrontIntegrator(5, True, True, sd, spp)
    integ.render()
    whole = integ.film().copy()
    # sample split: [0,3) then [3,8) into the same film
    integ.clear()
    integ.render(samples=(0, 3))
    integ.render(samples=(3, 8))
    assert np.array_equal(integ.film().view(np.uint64), whole.view(np.uint64))
    # pixel-window split with ragged (non multiple of 8) windows
    integ.clear()
    for win in ((0, 0, 13, 24), (13, 0, 40, 11), (13, 11, 40, 24)):
        integ.render(window=win)
    assert np.array_equal(integ.film().view(np.uint64), whole.view(np.uint64))
    # film_add: two half-sample films merged on the host side equal the whole within f64 rounding of one add
    integ.clear()
    integ.render(samples=(0, 4))
    a = integ.film().copy()
    integ.clear()
    integ.render(samples=(4, 8))
    integ.film_add(a)
    np.testing.assert_allclose(integ.film(), whole, rtol=1e-14, atol=0)


def test_wave_schedule_readback(pm):
    """integrators/base.rs:85-127: rendering in the reference's waves gives the same film bit for bit, and after every wave
    the film is the finished image of the sample indices rendered so far (weight sums grow wave by wave)"""
    res, spp = (40, 24), 21
    sd = pm.scenes.main_rs_scene(res)
    integ = pm.WavefrontIntegrator(5, True, True, sd, spp)
    integ.render()
    whole = integ.film().copy()
    integ.clear()
    seen = []

    def on_wave(w0, w1, it):
        f = it.film()
        assert np.isfinite(it.image()).all()
        seen.append((w0, w1, float(f[..., 3].sum())))
    integ.render_waves(on_wave=on_wave)
    assert [(a, b) for a, b, _ in seen] == [(0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 21)]
    assert all(x[2] < y[2] for x, y in zip(seen, seen[1:]))
    assert np.array_equal(integ.film().view(np.uint64), whole.view(np.uint64))


def test_film_exchange_through_device_pointers(pm):
    """the multi-GPU merge path of bench.py: film read into / added from buffers that live in HBM (torch tensors)"""
    import torch
    res, spp = (40, 24), 8
    sd = pm.scenes.main_rs_scene(res)
    integ = pm.WavefrontIntegrator(5, True, True, sd, spp)
    integ.render()
    whole = integ.film().copy()
    d = torch.empty((res[1], res[0], 4), dtype=torch.float64, device="cuda")
    integ.film(out=d)
    assert np.array_equal(d.cpu().numpy().view(np.uint64), whole.view(np.uint64))
    integ.clear()
    integ.render(samples=(0, 4))
    integ.film(out=d)
    integ.clear()
    integ.render(samples=(4, 8))
    integ.film_add(d)
    np.testing.assert_allclose(integ.film(), whole, rtol=1e-14, atol=0)
    stats = pm.distributed.render_distributed(integ)           # world size 1: plain render through the same entry point
    assert np.array_equal(stats.cpu().numpy().view(np.uint64), whole.view(np.uint64))


def test_render_errors(pm):
    sd = pm.scenes.main_rs_scene((16, 16))
    integ = pm.WavefrontIntegrator(5, True, True, sd, 4)
    with pytest.raises(pm.PmError):
        integ.render(samples=(0, 5))          # SamplerError::SampleCountExceeded (sampler.rs:110-122)
    with pytest.raises(pm.PmError):
        integ.render(window=(0, 0, 17, 16))   # outside pixel_bounds
