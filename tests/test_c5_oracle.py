"""C5 (area light + homogeneous medium) has NO reference integrator to compare with: the oracle's
extension is pinned here by closed forms instead - HG normalisation and mean cosine, the Beer-Lambert transmittance,
the white-furnace identity, and the area measure of patch sampling.  "parity unpinned by the reference": GPU parity for
C5 means oracle <-> GPU only (tests/test_c5_gpu.py)."""
import ctypes as C

import numpy as np
import pytest

import helpers


def _lib(ora):
    L = ora.lib()
    L.ora_henyey_greenstein.restype = C.c_float
    L.ora_henyey_greenstein.argtypes = [C.c_float, C.c_float]
    L.ora_sample_henyey_greenstein.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    L.ora_patch_sample_ctx.restype = C.c_int
    L.ora_patch_sample_ctx.argtypes = [C.c_void_p] * 4
    L.ora_patch_is_rectangle.restype = C.c_int
    L.ora_patch_is_rectangle.argtypes = [C.c_void_p]
    return L


@pytest.mark.parametrize("g", [-0.7, 0.0, 0.3, 0.6, 0.9])
def test_henyey_greenstein_is_normalised_and_sampled_by_its_own_pdf(ora, g):
    L = _lib(ora)
    # integral over the sphere: 2 pi * int_{-1}^{1} p(cos) dcos = 1 (Gauss-Legendre)
    x, w = np.polynomial.legendre.leggauss(400)
    vals = np.array([L.ora_henyey_greenstein(float(c), g) for c in x])
    assert abs(2 * np.pi * (vals * w).sum() - 1.0) < 2e-4
    rng = np.random.default_rng(1)
    wo = np.array([0.3, -0.5, 0.81], np.float32)
    wo /= np.linalg.norm(wo)
    n = 40_000
    out = np.zeros(4, np.float32)
    cs = np.empty(n)
    for i in range(n):
        u = rng.random(2, dtype=np.float32)
        L.ora_sample_henyey_greenstein(wo.ctypes.data, g, u.ctypes.data, out.ctypes.data)
        wi = out[:3].astype(np.float64)
        assert abs(np.linalg.norm(wi) - 1) < 1e-4
        cs[i] = float(wi @ wo)
        # value == pdf == HG(dot(wo, wi))  (phase_function.rs:43-50)
        assert out[3] == pytest.approx(L.ora_henyey_greenstein(np.float32(cs[i]), g), rel=2e-3, abs=1e-6)
    # pbrt convention (wo and wi both point away): E[dot(wo, wi)] = -g
    assert abs(cs.mean() + g) < 4 / np.sqrt(n)


def test_patch_sampling_has_the_area_measure(ora):
    """rectangle: uniform (u, 1.0) => area pdf 1 / A, converted to solid angle; the Monte-Carlo estimate of the solid angle
    int dw = E[1 / pdf_w] must equal the analytic solid angle of the rectangle seen from the reference point"""
    L = _lib(ora)
    quad = np.array([[-0.5, -0.25, 2], [0.5, -0.25, 2], [-0.5, 0.25, 2], [0.5, 0.25, 2]], np.float32)
    assert L.ora_patch_is_rectangle(quad.ctypes.data) == 1
    ref = np.zeros(3, np.float32)
    rng = np.random.default_rng(2)
    out = np.zeros(7, np.float32)
    acc = []
    for _ in range(20_000):
        u = rng.random(2, dtype=np.float32)
        assert L.ora_patch_sample_ctx(quad.ctypes.data, ref.ctypes.data, u.ctypes.data, out.ctypes.data) == 1
        assert abs(out[2] - 2.0) < 1e-5 and abs(out[0]) <= 0.5 + 1e-6 and abs(out[1]) <= 0.25 + 1e-6
        acc.append(1.0 / out[6])

    def omega(a, b, d):      # solid angle of a centred a x b rectangle at distance d
        return 4 * np.arctan(a * b / (2 * d * np.sqrt(4 * d * d + a * a + b * b)))
    assert np.mean(acc) == pytest.approx(omega(1.0, 0.5, 2.0), rel=5e-3)
    skew = quad.copy()
    skew[3] += np.array([0.3, 0.2, 0.1], np.float32)
    assert L.ora_patch_is_rectangle(skew.ctypes.data) == 0
    tri = quad.copy()
    tri[3] = tri[2]
    assert L.ora_patch_is_rectangle(tri.ctypes.data) == 0           # a triangle (degenerate patch) is sampled bilinearly


def _wall_scene(pm, res, medium=None, depth_light=1.0):
    """a diffuse wall at z = 3 facing the camera at the origin (fov 4 deg: all rays nearly axial), lit by a point light on
    the axis at z = 3 - depth_light"""
    host = pm.host
    sp = host.Spectra()
    sd = host.SceneDescription()
    cfw = host.Transform.look_at((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0))
    sd.film = host.make_film(sp, res, filter_kind=host.PM_FILTER_BOX, filter_radius=(0.5, 0.5))
    sd.camera, rfw = host.make_camera(cfw.inverse(), 4.0, res)
    quad = np.array([[-50, -50, 3], [-50, 50, 3], [50, -50, 3], [50, 50, 3]], np.float32)
    sd.prims = [pm.BilinearPatchMesh(quad, [[0, 1, 2, 3]])]
    sd.materials = [host.diffuse(host.constant_spectrum(0.5))]
    sd.prim_materials = [0]
    lt = host.Transform.translate((0.0, 0.0, 3.0 - depth_light))
    sd.lights = [host.light(host.PM_LIGHT_POINT, sp.d65.copy(), 1.0 / sp.photometric(sp.d65), lt.m)]
    sd.medium = medium
    return sd, sp


def test_beer_lambert_transmittance(pm, ora):
    """purely absorbing medium (sigma_s = 0) everywhere: the camera segment (3 units, stochastic: a collision ends the
    path) and the shadow segment (1 unit, closed form) together attenuate the wall by exp(-sigma_a * 4)"""
    VOL = pm._capi.PM_INTEGRATOR_VOL_PATH
    res, spp, sigma = (4, 4), 4096, 0.4
    sd0, sp = _wall_scene(pm, res)
    r0 = helpers.oracle_renderer_from(sd0, ora)
    r0.render(helpers.render_params(pm, sd0, 16, 1, integrator=VOL, samples=(0, 16)), 8)      # no medium: VOL == SimplePath
    base = r0.image()[..., 1].mean()
    ref = helpers.oracle_renderer_from(sd0, ora)
    ref.render(helpers.render_params(pm, sd0, 16, 1, samples=(0, 16)), 8)
    assert np.array_equal(ref.film(), r0.film())
    sd1, _ = _wall_scene(pm, res, pm.host.homogeneous_medium(sp.constant_table(sigma), sp.constant_table(0.0), 1.0, 0.0))
    r1 = helpers.oracle_renderer_from(sd1, ora)
    r1.render(helpers.render_params(pm, sd1, spp, 1, integrator=VOL), 8)
    got = r1.image()[..., 1].mean() / base
    want = np.exp(-sigma * 4.0)
    n_eff = res[0] * res[1] * spp
    p = np.exp(-sigma * 3.0)                                        # survival probability of the camera segment
    assert got == pytest.approx(want, rel=5 * np.sqrt((1 - p) / (p * n_eff)) + 2e-3)
    # a SimplePath render ignores the medium
    r2 = helpers.oracle_renderer_from(sd1, ora)
    r2.render(helpers.render_params(pm, sd1, 16, 1, samples=(0, 16)), 8)
    assert np.array_equal(r2.film(), ref.film())


def test_white_furnace(pm, ora):
    """a non-absorbing scattering medium (albedo 1) inside a box, under a uniform environment of radiance L: whatever the
    phase function, every pixel converges to L (paths only redistribute the uniform radiance).  Depth 64 makes the
    truncation negligible (optical thickness <= 3.5)."""
    host = pm.host
    VOL = pm._capi.PM_INTEGRATOR_VOL_PATH
    sp = host.Spectra()
    res, spp = (6, 6), 1024
    sd = host.SceneDescription()
    cfw = host.Transform.look_at((0.0, 0.0, -4.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    sd.film = host.make_film(sp, res, filter_kind=host.PM_FILTER_BOX, filter_radius=(0.5, 0.5))
    sd.camera, rfw = host.make_camera(cfw.inverse(), 20.0, res)
    # two specks in opposite far corners: the scene bounds (hence the infinite light's p_light = p + 2 R wi,
    # lights/infinite.rs:52-70) must enclose the medium, as geometry encloses media in any real scene
    far = np.array([[900, 900, 900], [900.1, 900, 900], [900, 900.1, 900],
                    [-900, -900, -900], [-900.1, -900, -900], [-900, -900.1, -900]], np.float32)
    sd.prims = [pm.TriangleMesh(pm.scenes._xf_points(rfw.m, far), [[0, 1, 2], [3, 4, 5]])]
    sd.materials = [host.diffuse(host.constant_spectrum(0.5))]
    sd.prim_materials = [0, 0]
    sky = sp.constant_table(1.0)
    sd.lights = [host.light(host.PM_LIGHT_UNIFORM_INFINITE, sky, 1.0)]
    r0 = helpers.oracle_renderer_from(sd, ora)
    r0.render(helpers.render_params(pm, sd, 4, 64, integrator=VOL, samples=(0, 4)), 8)
    base = r0.image()
    lo = pm.scenes._xf_points(rfw.m, np.array([[-1, -1, -1]], np.float32))[0]
    hi = pm.scenes._xf_points(rfw.m, np.array([[1, 1, 1]], np.float32))[0]
    for g in (0.0, 0.6):
        sd.medium = host.homogeneous_medium(sp.constant_table(0.0), sp.constant_table(1.0), 1.0, g, bounds=tuple(lo) + tuple(hi))
        r1 = helpers.oracle_renderer_from(sd, ora)
        r1.render(helpers.render_params(pm, sd, spp, 64, integrator=VOL), 8)
        ratio = r1.image().mean() / base.mean()
        assert ratio == pytest.approx(1.0, abs=0.02), (g, ratio)
