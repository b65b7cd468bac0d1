"""Area lights on SPHERES (round 3; Sphere::sample / sample_with_context, /root/reference/src/shapes/sphere.rs:100-213).  The
reference has the shape-sampling code but no light that calls it (lights/base.rs:18-22), so - like the rest of C5 - the oracle's
restatement is pinned by closed forms: the cone branch has the constant pdf 1 / (2 pi (1 - cos theta_max)) and puts every sample on
the visible cap of the sphere; the inside branch has the area measure (E[1 / pdf_w] = 4 pi); a diffuse wall under a sphere of
radiance L receives E = pi L (r / D)^2 cos(theta).  "parity unpinned by the reference"; GPU parity is oracle <-> GPU
(tests/test_c5_gpu.py)."""
import ctypes as C

import numpy as np
import pytest

import helpers


def _lib(ora):
    L = ora.lib()
    L.ora_sphere_sample_ctx.restype = C.c_int
    L.ora_sphere_sample_ctx.argtypes = [C.c_void_p] * 7
    return L


def _sample(L, sph, t, ctx_p, ctx_n, u):
    s5 = np.asarray(sph, np.float32)
    m = np.ascontiguousarray(t.m, np.float32)
    mi = np.ascontiguousarray(t.m_inv, np.float32)
    cp = np.asarray(ctx_p, np.float32)
    cn = None if ctx_n is None else np.asarray(ctx_n, np.float32)
    uu = np.asarray(u, np.float32)
    out = np.zeros(7, np.float32)
    ok = L.ora_sphere_sample_ctx(s5.ctypes.data, m.ctypes.data, mi.ctypes.data, cp.ctypes.data,
                                 None if cn is None else cn.ctypes.data, uu.ctypes.data, out.ctypes.data)
    return ok, out.astype(np.float64)


@pytest.mark.parametrize("radius,centre,ctx", [(0.3, (0.4, -0.1, 0.3), (-0.5, 0.6, 1.5)),      # ordinary cone
                                               (0.05, (1.0, 2.0, -3.0), (-4.0, 0.5, 6.0))])    # sin^2 theta_max < sin^2(1.5 deg): Taylor branch
def test_cone_branch_closed_forms(pm, ora, radius, centre, ctx):
    L = _lib(ora)
    host = pm.host
    a = np.radians(33.0)
    rot = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]], np.float32)
    t = host.Transform.translate(centre) * host.Transform.from_matrix(rot)
    c = np.asarray(centre, np.float64)
    x = np.asarray(ctx, np.float64)
    D = np.linalg.norm(c - x)
    sin2 = (radius / D) ** 2
    cos_max = np.sqrt(1 - sin2)
    want_pdf = 1.0 / (2 * np.pi * (1 - cos_max))
    rng = np.random.default_rng(5)
    n = 20_000
    cosines = np.empty(n)
    axis = (c - x) / D
    for i in range(n):
        ok, o = _sample(L, (radius, -radius, radius, 360.0, 0.0), t, ctx, (0.0, 1.0, 0.0), rng.random(2, dtype=np.float32))
        assert ok == 1
        p, nrm, pdf = o[:3], o[3:6], o[6]
        assert pdf == pytest.approx(want_pdf, rel=2e-3 if sin2 < 0.00068523 else 2e-5)
        assert np.linalg.norm(p - c) == pytest.approx(radius, rel=2e-4)             # on the sphere
        assert np.allclose(nrm, (p - c) / radius, atol=2e-3)                         # outward normal
        w = (p - x) / np.linalg.norm(p - x)
        cosines[i] = w @ axis
        assert cosines[i] >= cos_max - 1e-5                                          # inside the cone
        assert nrm @ (x - p) >= -1e-4 * D                                            # on the cap that faces the point
    # NOT asserted: uniformity inside the cone.  Upstream writes the angle alpha at the sphere's centre as
    #   cos_alpha = sin2_theta / sin_theta_max + cos_theta * safe_sqrt(1 - sin2_theta / sin_theta_max * sin2_theta_max)
    # (sphere.rs:180-181; pbrt-v4 divides by sin_theta_max SQUARED under the root), which is >= 1 for most u and clamps to the
    # point of the sphere nearest to the receiver.  The restatement follows upstream operation for operation (the rule for every
    # upstream quirk that changes results), so the samples crowd the axis:
    assert cosines.mean() > (1 + cos_max) / 2


def test_inside_branch_has_the_area_measure(pm, ora):
    """a point inside the sphere: uniform by area, pdf converted to solid angle; every direction sees the sphere once, so the
    Monte-Carlo estimate of the solid angle E[1 / pdf_w] is 4 pi.  A medium context (no normal) takes the same branch."""
    L = _lib(ora)
    t = pm.host.Transform.translate((0.2, -0.3, 0.5))
    radius = 2.0
    rng = np.random.default_rng(6)
    for ctx_n in ((0.0, 0.0, 1.0), None):
        acc = []
        for _ in range(40_000):
            ok, o = _sample(L, (radius, -radius, radius, 360.0, 0.0), t, (0.7, 0.1, -0.2), ctx_n, rng.random(2, dtype=np.float32))
            if ok:
                assert np.linalg.norm(o[:3] - np.array([0.2, -0.3, 0.5])) == pytest.approx(radius, rel=1e-4)
                acc.append(1.0 / o[6])
        est = np.sum(acc) / 40_000
        assert est == pytest.approx(4 * np.pi, rel=5 * np.std(acc) / np.mean(acc) / np.sqrt(len(acc)) + 1e-3)


def test_reverse_orientation_flips_the_sample_normal(pm, ora):
    L = _lib(ora)
    t = pm.host.Transform.translate((0.0, 0.0, 4.0))
    u = np.array([0.3, 0.6], np.float32)
    _, a = _sample(L, (0.5, -0.5, 0.5, 360.0, 0.0), t, (0, 0, 0), (0, 0, 1), u)
    _, b = _sample(L, (0.5, -0.5, 0.5, 360.0, 1.0), t, (0, 0, 0), (0, 0, 1), u)
    assert np.array_equal(a[:3], b[:3]) and np.array_equal(a[3:6], -b[3:6]) and a[6] == b[6]


def _wall_under_light(pm, res, sphere_light, two_sided=False):
    """a diffuse wall at z = 3 seen by a narrow camera at the origin; a light one unit in front of the wall and one unit off the axis
    (outside the camera's view): a point light, or a sphere of radius 0.3 that emits"""
    host = pm.host
    sp = host.Spectra()
    sd = host.SceneDescription()
    cfw = host.Transform.look_at((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0))
    sd.film = host.make_film(sp, res, filter_kind=host.PM_FILTER_BOX, filter_radius=(0.5, 0.5))
    sd.camera, rfw = host.make_camera(cfw.inverse(), 2.0, res)
    quad = pm.scenes._xf_points(rfw.m, np.array([[-50, -50, 3], [-50, 50, 3], [50, -50, 3], [50, 50, 3]], np.float32))
    sd.prims = [pm.BilinearPatchMesh(quad, [[0, 1, 2, 3]])]
    sd.materials = [host.diffuse(host.constant_spectrum(0.5))]
    sd.prim_materials = [0]
    lt = rfw * host.Transform.translate((1.0, 0.0, 2.0))
    if sphere_light:
        sd.prims.append(pm.Sphere(0.3, -0.3, 0.3, 360.0, lt.m, lt.m_inv, False))
        sd.prim_materials.append(0)
        sd.lights = [host.area_light(sp.d65.copy(), 1.0 / sp.photometric(sp.d65), 1, two_sided=two_sided)]
    else:
        sd.lights = [host.light(host.PM_LIGHT_POINT, sp.d65.copy(), 1.0 / sp.photometric(sp.d65), lt.m)]
    return sd


def test_irradiance_under_a_sphere_light(pm, ora):
    """next-event estimation only (max_depth 1): a point light of intensity I gives the wall point on the axis I cos(theta) / D^2.
    The sphere of radiance L = I is sampled with the cone's constant pdf 1 / Omega, Omega = 2 pi (1 - cos theta_max), and - with
    upstream's cos_alpha (see test_cone_branch_closed_forms) - at the point nearest to the receiver, so the estimate is
    L Omega cos(theta): the ratio of the two images is Omega D^2 (= pi r^2 (1 + sin^2 theta_max / 4 + ...); exact cone sampling
    would give pi r^2)."""
    res, spp = (4, 4), 2048
    sd_p = _wall_under_light(pm, res, False)
    sd_s = _wall_under_light(pm, res, True)
    rp = helpers.oracle_renderer_from(sd_p, ora)
    rp.render(helpers.render_params(pm, sd_p, 16, 1, samples=(0, 16)), 8)
    rs = helpers.oracle_renderer_from(sd_s, ora)
    rs.render(helpers.render_params(pm, sd_s, spp, 1), 8)
    ratio = rs.image()[..., 1].mean() / rp.image()[..., 1].mean()
    D2 = 2.0
    omega = 2 * np.pi * (1 - np.sqrt(1 - 0.09 / D2))
    assert ratio == pytest.approx(omega * D2, rel=5e-3)
    assert ratio == pytest.approx(np.pi * 0.09, rel=3e-2)
