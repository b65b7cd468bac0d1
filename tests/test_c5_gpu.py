"""BASELINE.json configs[4] (C5): diffuse area light + homogeneous participating medium, depth 16, on the GPU vs the oracle.

Both are EXTENSIONS (the reference has neither an area light nor a medium-aware integrator), so parity
here means oracle <-> GPU only; the oracle's extension is pinned by closed forms in tests/test_c5_oracle.py.  Bar: in
the oracle's correctly-rounded transcendental mode >= 99.99 % of the (pixel, sample) film contributions are bit-equal
(tests/test_branches_gpu.py explains the mode); in the faithful mode the image agrees statistically.
"""
import numpy as np
import pytest

import helpers
from test_branches_gpu import _bit_equal, _report

pytestmark = pytest.mark.gpu


def test_c5_scene_bitwise(pm, ora):
    VOL = pm._capi.PM_INTEGRATOR_VOL_PATH
    res, spp = (64, 64), 8
    sd = pm.scenes.c5_scene(res)
    frac, tot = _bit_equal(pm, ora, sd, spp, max_depth=16, integrator=VOL)
    _report("C5: area light + medium in the box, VolPath depth 16", frac, tot)
    assert frac >= 0.9999
    # the medium matters, and SimplePath ignores it
    a = pm.WavefrontIntegrator(16, True, True, sd, 4, integrator=VOL)
    a.render()
    b = pm.WavefrontIntegrator(16, True, True, sd, 4)
    st = b.render()
    assert helpers.rel_rmse(a.image(), b.image()) > 5e-2
    clear = pm.scenes.cornell_scene(res, area_light=True)
    c = pm.WavefrontIntegrator(16, True, True, clear, 4)
    c.render()
    assert np.array_equal(b.film().view(np.uint64), c.film().view(np.uint64))
    assert st["shadow_rays"] > 0


def test_c5_faithful_mode_statistics(pm, ora):
    """the same scene against the oracle's FAITHFUL (glibc f32) mode: same expectation, a few realisations flipped"""
    VOL = pm._capi.PM_INTEGRATOR_VOL_PATH
    res, spp = (48, 48), 16
    sd = pm.scenes.c5_scene(res)
    integ = pm.WavefrontIntegrator(16, True, True, sd, spp, integrator=VOL)
    oren = helpers.oracle_renderer_from(sd, ora)
    p = helpers.render_params(pm, sd, spp, 16, integrator=VOL)
    st = integ.render()
    oc = oren.render(p, 8)
    gi, oi = integ.image(), oren.image()
    assert np.isfinite(gi).all() and gi.mean() > 0
    assert abs(st["closest_rays"] - oc["useful_closest_rays"]) <= 5e-3 * oc["closest_rays"]
    assert abs(gi.mean() - oi.mean()) <= 5e-3 * abs(oi.mean())
    assert helpers.rel_rmse(gi, oi) <= 5e-2


@pytest.mark.parametrize("two_sided", [False, True])
def test_area_lights_without_medium_bitwise(pm, ora, two_sided):
    """SimplePath with DiffuseAreaLights: NEE through BilinearPatch::sample_with_context (a rectangle and - B14 path - a
    non-rectangular patch and a triangle emitter), emission seen directly and through the glass sphere (specular bounce)"""
    res, spp = (64, 64), 8
    sd = pm.scenes.cornell_scene(res, area_light=True)
    sd.lights[0]["two_sided"] = two_sided
    sp = pm.host.Spectra()
    rfw = sd.render_from_world
    # a skewed (non-rectangular) patch emitter on the back wall and a triangle emitter on the floor
    skew = pm.scenes._xf_points(rfw.m, np.array([[-0.6, 0.2, -0.97], [-0.2, 0.25, -0.97], [-0.55, 0.6, -0.97], [-0.1, 0.75, -0.9]], np.float32))
    sd.prims.append(pm.BilinearPatchMesh(skew, [[0, 1, 2, 3]]))
    sd.prim_materials.append(0)
    tri = pm.scenes._xf_points(rfw.m, np.array([[0.3, -0.97, 0.5], [0.8, -0.97, 0.6], [0.5, -0.97, 0.9]], np.float32))
    sd.prims.append(pm.TriangleMesh(tri, [[0, 1, 2]]))
    sd.prim_materials.append(0)
    n = len(sd.prim_materials)
    t = sp.blackbody_table(4000.0)
    sd.lights.append(pm.host.area_light(t, 3.0 / sp.photometric(t), n - 2, two_sided=two_sided))
    sd.lights.append(pm.host.area_light(sp.d65.copy(), 5.0 / sp.photometric(sp.d65), n - 1, two_sided=True))
    frac, tot = _bit_equal(pm, ora, sd, spp)
    _report(f"area lights (rectangle, skewed patch, triangle), two_sided={two_sided}", frac, tot)
    assert frac >= 0.9999
    for sl in (False,):                      # emission found by BSDF sampling alone
        frac, tot = _bit_equal(pm, ora, sd, spp, max_depth=3, sample_lights=sl)
        _report(f"area lights, sample_lights={sl}", frac, tot)
        assert frac >= 0.9999


def test_chromatic_global_medium_bitwise(pm, ora):
    """a medium that fills all of space (infinite bounds) with wavelength-dependent coefficients: the spectral weights
    T_i sigma_s,i / (T_0 sigma_s,0) and the pass-through weights differ per wavelength; infinite + directional lights"""
    VOL = pm._capi.PM_INTEGRATOR_VOL_PATH
    res, spp = (64, 64), 8
    sd = pm.scenes.main_rs_scene(res)
    lam = np.arange(360, 831, dtype=np.float32)
    sig_s = (0.02 + 0.1 * (830.0 - lam) / 470.0).astype(np.float32)            # bluer light scatters more
    sig_a = (0.01 + 0.02 * (lam - 360.0) / 470.0).astype(np.float32)
    for g in (0.0, -0.4, 0.75):
        sd.medium = pm.host.homogeneous_medium(sig_a, sig_s, 1.5, g)
        frac, tot = _bit_equal(pm, ora, sd, spp, max_depth=8, integrator=VOL)
        _report(f"chromatic global medium, g={g}", frac, tot)
        assert frac >= 0.9999


def test_medium_errors(pm):
    sd = pm.scenes.c5_scene((16, 16))
    sd.medium["g"] = 1.0
    with pytest.raises(pm.PmError):
        pm.WavefrontIntegrator(5, True, True, sd, 4)
    sd = pm.scenes.c5_scene((16, 16))
    sd.lights[0]["prim_id"] = 10_000
    integ = pm.WavefrontIntegrator(5, True, True, sd, 4)
    with pytest.raises(pm.PmError):
        integ.render()


def test_c2_scene_area_instantiation_bitwise(pm, ora, monkeypatch):
    """the Cornell box as 36 triangles under a ceiling quad EMITTER (two triangle area lights), all diffuse.
    Such a scene (area lights, no medium, no textures) runs the AREA instantiation of the plain shade kernel (round 3); the EXT
    kernel (round 2's routing, PBRT_MI355_SHADE_EXT=1) must give the same film bit for bit."""
    res, spp = (64, 64), 8
    sd = pm.scenes.c2_scene(res)
    assert len(sd.prim_materials) == 36 and len(sd.lights) == 2 and len(sd.prims) == 1
    frac, tot = _bit_equal(pm, ora, sd, spp)
    _report("C2 (SURVEY 8d): 36 triangles, ceiling quad emitter, SimplePath depth 5", frac, tot)
    assert frac >= 0.9999
    a = pm.WavefrontIntegrator(5, True, True, sd, 4)
    st = a.render()
    assert st["shadow_rays"] > 0 and a.aggregate.info.flat_prims == 36
    monkeypatch.setenv("PBRT_MI355_SHADE_EXT", "1")
    b = pm.WavefrontIntegrator(5, True, True, sd, 4)
    b.render()
    assert np.array_equal(a.film().view(np.uint64), b.film().view(np.uint64))
    monkeypatch.delenv("PBRT_MI355_SHADE_EXT")
    # emission is seen directly (camera ray = specular bounce) and found by BSDF sampling when light sampling is off
    frac, tot = _bit_equal(pm, ora, sd, 4, max_depth=3, sample_lights=False)
    _report("C2 (SURVEY 8d), sample_lights off", frac, tot)
    assert frac >= 0.9999
    # the emitter is visible: the brightest pixels of the image lie on it
    img = a.image()
    assert img.max() > 20 * np.median(img)


def _sphere_emitter_scene(pm, res, medium=None):
    """the Cornell box with its glass sphere, lit by SPHERE emitters only (round 3): a small one-sided sphere hanging in the box
    (receivers are outside it: Sphere::sample_with_context's cone branch), a sphere the size of a pea inside the cone's
    Taylor-expansion range, and - reverse-oriented, so that it emits inwards - a sphere of radius 6 around the whole scene
    (every receiver is inside it: the area branch), seen through the open front of the box"""
    host = pm.host
    sp = host.Spectra()
    sd = pm.scenes.cornell_scene(res, glass_sphere=True, area_light=False, medium=medium)
    rfw = sd.render_from_world
    sd.lights = []
    # walls only: the enclosing sphere sends rays up through the floor from BELOW, where the bottom faces of the two boxes coincide
    # with it - exact-t ties between different primitives, which the reference resolves by tree order (tie class)
    mesh = sd.prims[0]
    sd.prims[0] = pm.TriangleMesh(mesh.positions, mesh.indices[:10])
    sd.prim_materials = sd.prim_materials[:10] + sd.prim_materials[len(mesh.indices):]
    for centre, radius, reverse, power, two_sided, table in (((-0.3, 0.55, 0.35), 0.18, False, 30.0, False, sp.d65.copy()),
                                                            ((0.6, 0.7, -0.5), 0.02, False, 900.0, True, sp.blackbody_table(3200.0)),
                                                            ((0.0, 0.0, 0.0), 6.0, True, 0.4, False, sp.d65.copy())):
        t = rfw * host.Transform.translate(centre)
        sd.prims.append(pm.Sphere(radius, -radius, radius, 360.0, t.m, t.m_inv, reverse))
        sd.prim_materials.append(0)
        sd.lights.append(host.area_light(table, power / sp.photometric(table), len(sd.prim_materials) - 1, two_sided=two_sided))
    return sd


def test_sphere_emitters_bitwise(pm, ora):
    """area lights on spheres (Sphere::sample / sample_with_context, sphere.rs:100-213; round 3): NEE from surface points outside
    and inside an emitting sphere, emission seen directly, through the glass sphere and by BSDF sampling"""
    res, spp = (64, 64), 8
    sd = _sphere_emitter_scene(pm, res)
    frac, tot = _bit_equal(pm, ora, sd, spp)
    _report("sphere emitters (cone branch, Taylor branch, enclosing sphere), SimplePath depth 5", frac, tot)
    assert frac >= 0.9999
    frac, tot = _bit_equal(pm, ora, sd, 4, max_depth=3, sample_lights=False)
    _report("sphere emitters, sample_lights off", frac, tot)
    assert frac >= 0.9999
    a = pm.WavefrontIntegrator(5, True, True, sd, 4)
    st = a.render()
    img = a.image()
    assert st["shadow_rays"] > 0 and np.isfinite(img).all() and img.max() > 10 * np.median(img) > 0


def test_sphere_emitters_in_a_medium_bitwise(pm, ora):
    """the same lights sampled from vertices INSIDE the medium (no surface normal: the receiving point itself decides inside /
    outside) by the VolPath integrator"""
    VOL = pm._capi.PM_INTEGRATOR_VOL_PATH
    sd = _sphere_emitter_scene(pm, (48, 48), medium=(0.05, 0.6, 0.3))
    frac, tot = _bit_equal(pm, ora, sd, 8, max_depth=8, integrator=VOL)
    _report("sphere emitters + homogeneous medium, VolPath depth 8", frac, tot)
    assert frac >= 0.9999
