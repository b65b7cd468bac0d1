"""The product's host layer (pbrt-rs_amd/csrc/pm_host.cpp + host.py + scenes.py) against the ORACLE's own restatement of
the reference's constructors (oracle/ref_host.h + oracle/host_scene.py), bit for bit: every derived quantity that crosses
the C ABI - camera matrices, filter table, densified spectra, light scales, colour matrices, primitive transforms.  With
this, the parity tests no longer share their scene construction; the GPU half renders a scene built
WITHOUT any product host code and compares it with the GPU rendering the product-built scene.
"""
import ctypes as C

import numpy as np
import pytest

import helpers


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def hs():
    import host_scene
    return host_scene


@pytest.fixture(scope="module")
def color(pm, hs):
    d = np.load(pm.host._DATA)
    z, co = pm.host.Spectra().rgb_table()          # data: the regenerated table file (pinned in test_oracle_golden.py)
    return hs.ColorData(d["cie_x"], d["cie_y"], d["cie_z"], d["d65_interleaved"], z, co)


def test_transform_constructors(pm, hs):
    rng = np.random.default_rng(11)
    T = pm.host.Transform
    for _ in range(200):
        pos, look = rng.normal(size=3) * 5, rng.normal(size=3) * 5
        up = rng.normal(size=3)
        a, b = T.look_at(pos, look, up), hs.look_at(pos, look, up)
        assert np.array_equal(_bits(a.m), _bits(b.m)) and np.array_equal(_bits(a.m_inv), _bits(b.inv))
        d = rng.normal(size=3) * 3
        ta, tb = T.translate(d), hs.translate(d)
        assert np.array_equal(_bits(ta.m), _bits(tb.m)) and np.array_equal(_bits(ta.m_inv), _bits(tb.inv))
        ca, cb = a * ta, b * tb
        assert np.array_equal(_bits(ca.m), _bits(cb.m)) and np.array_equal(_bits(ca.m_inv), _bits(cb.inv))
        pts = rng.normal(size=(7, 3)).astype(np.float32)
        assert np.array_equal(_bits(pm.scenes._xf_points(ca.m, pts)), _bits(cb.points(pts)))


def test_perspective_camera(pm, hs):
    rng = np.random.default_rng(12)
    for _ in range(100):
        pos, look = rng.normal(size=3) * 4, rng.normal(size=3) * 4
        wfc_a = pm.host.Transform.look_at(pos, look, (0, 0, 1)).inverse()
        wfc_b = hs.look_at(pos, look, (0, 0, 1)).inverse()
        fov = float(rng.uniform(10, 120))
        res = (int(rng.integers(8, 4000)), int(rng.integers(8, 3000)))
        asp = res[0] / res[1]
        sw = (-asp, -1.0, asp, 1.0) if rng.random() < 0.5 else (-1.0, -1.0 / asp, 1.0, 1.0 / asp)
        lens, focal = float(rng.uniform(0, 0.2)), float(rng.uniform(1, 20))
        ca, rfw_a = pm.host.make_camera(wfc_a, fov, res, sw, lens, focal)
        cb, rfw_b = hs.perspective_camera(wfc_b, fov, sw, res, lens, focal)
        assert bytes(ca) == bytes(cb)
        assert np.array_equal(_bits(rfw_a.m), _bits(rfw_b.m)) and np.array_equal(_bits(rfw_a.m_inv), _bits(rfw_b.inv))


def test_filter_table(pm, hs):
    sp = pm.host.Spectra()
    for rx, ry, sigma in ((1.5, 1.5, 0.5), (2.0, 1.25, 0.7), (0.75, 3.0, 1.1)):
        film = pm.host.make_film(sp, (8, 8), filter_radius=(rx, ry), sigma=sigma)
        vals, nx, ny = hs.gaussian_filter_table(rx, ry, sigma)
        assert (film["filter_nx"], film["filter_ny"]) == (nx, ny)
        assert np.array_equal(_bits(film["filter_vals"]), _bits(vals))


def test_spectra_and_colour_space(pm, hs, color):
    sp = pm.host.Spectra()
    assert np.array_equal(_bits(sp.d65), _bits(color.d65))
    assert np.array_equal(_bits(sp.xyz_from_rgb), _bits(color.xyz_from_rgb))
    assert np.array_equal(_bits(sp.rgb_from_xyz), _bits(color.rgb_from_xyz))
    for temp in (1500.0, 3000.0, 5500.0, 6504.0, 12000.0):
        a, b = sp.blackbody_table(temp), color.blackbody_table(temp)
        assert np.array_equal(_bits(a), _bits(b))
        assert _bits(np.float32(sp.photometric(a))) == _bits(color.photometric(b))
    rng = np.random.default_rng(13)
    for _ in range(50):
        rgb = rng.random(3).astype(np.float32) * rng.choice([0.2, 1.0, 4.0])
        ta, _, _ = sp.rgb_illuminant_table(rgb)
        assert np.array_equal(_bits(ta), _bits(color.rgb_illuminant_table(rgb)))
        q = np.minimum(rgb, 1.0)
        alb = sp.rgb_albedo(q)
        assert np.array_equal(_bits([alb["a"], alb["b"], alb["c"]]), _bits(color.rgb_coeffs(q)))


def _same_scene(pm, hs, color, res):
    """the reference's scene from both constructions; asserts every struct the C ABI / the oracle takes is bit-identical"""
    sd = pm.scenes.main_rs_scene(res)
    os_ = hs.main_rs_scene(color, res)
    assert bytes(sd.camera) == bytes(os_.camera)
    fa, fb = sd.c_film(), os_.film
    for name in ("full_resolution", "pixel_bounds", "output_rgb_from_sensor_rgb", "filter_radius"):
        assert bytes(getattr(fa, name)) == bytes(getattr(fb, name)), name
    for name in ("max_component_value", "imaging_ratio", "filter_kind", "filter_nx", "filter_ny"):
        assert getattr(fa, name) == getattr(fb, name), name
    n = fa.filter_nx * fa.filter_ny
    assert np.array_equal(_bits(np.ctypeslib.as_array(fa.filter_vals, (n,))), _bits(np.ctypeslib.as_array(fb.filter_vals, (n,))))
    assert np.array_equal(_bits(np.ctypeslib.as_array(fa.sensor_rgb_bar, (3 * 471,))), _bits(np.ctypeslib.as_array(fb.sensor_rgb_bar, (3 * 471,))))
    la = sd.c_lights()
    assert len(sd.lights) == len(os_.lights) == 2
    for a, b in zip(la, os_.lights):
        assert a.type == b.type and _bits(np.float32(a.scale)) == _bits(np.float32(b.scale))
        assert bytes(a.render_from_light) == bytes(b.render_from_light)
        assert np.array_equal(_bits(np.ctypeslib.as_array(a.spectrum, (471,))), _bits(np.ctypeslib.as_array(b.spectrum, (471,))))
    ma = sd.c_materials()
    assert len(sd.materials) == len(os_.materials)
    for a, b in zip(ma, os_.materials):
        assert a.type == b.type and a.spectrum.kind == b.spec.kind
        assert _bits([a.spectrum.a, a.spectrum.b, a.spectrum.c, a.spectrum.d]).tolist() == _bits([b.spec.a, b.spec.b, b.spec.c, b.spec.d]).tolist()
        assert (a.u_roughness, a.v_roughness, a.remap_roughness, a.texture) == (b.u_rough, b.v_rough, b.remap_roughness, b.texture)
    assert np.array_equal(sd.c_prim_materials(), os_.prim_materials)
    return sd, os_


def test_reference_scene_construction_is_bit_identical(pm, ora, hs, color):
    sd, os_ = _same_scene(pm, hs, color, (400, 400))
    # primitives: same hits for the same rays <=> same transforms and vertices (bitwise t, u, v)
    sc = helpers.oracle_scene_from(sd, ora)
    rays = pm.scenes.incoherent_rays(20000, seed=4, radius=6.0)
    a, _ = sc.intersect(rays)
    b, _ = os_.scene.intersect(rays)
    assert (a["prim_id"] != 0xFFFFFFFF).mean() > 0.05
    assert a.tobytes() == b.tobytes()
    lo_a, hi_a = sc.bounds()
    lo_b, hi_b = os_.scene.bounds()
    assert np.array_equal(_bits(lo_a), _bits(lo_b)) and np.array_equal(_bits(hi_a), _bits(hi_b))


@pytest.mark.gpu
def test_gpu_render_against_oracle_built_without_product_host_code(pm, ora, hs, color):
    """GPU renders the product-built scene; the oracle renders the scene built by oracle/host_scene.py: per-sample film
    contributions bit-equal (correctly-rounded mode, bar as in test_branches_gpu.py)"""
    res, spp = (96, 96), 4
    sd, os_ = _same_scene(pm, hs, color, res)
    integ = pm.WavefrontIntegrator(5, True, True, sd, spp)
    eq = tot = 0
    with ora.correctly_rounded():
        oren = os_.renderer()
        for s in range(spp):
            p = helpers.render_params(pm, sd, spp, 5, True, True, samples=(s, s + 1))
            integ.clear()
            integ.render(samples=(s, s + 1))
            oren.clear()
            oren.render(p, 8)
            g, o = integ.film(), oren.film()
            same = (g.view(np.uint64) == o.view(np.uint64)).all(axis=-1)
            eq += int(same.sum())
            tot += same.size
    print(f"[bit-parity] main.rs scene, oracle-side construction: {eq / tot * 100:.4f} % of {tot} samples bit-equal")
    assert eq / tot >= 0.9999
