"""GPU parity (through the C ABI, oracle as the checker) of the device branches the round-1 tests did not reach:
partial / rotated / non-uniformly scaled spheres through the flat sweep and the tree walk, the MIXED walk on a large scene,
rough and dispersive dielectrics, the thin lens, box and triangle filters, the film's max_component_value clamp, and the
raw sampler stream.

Render comparisons run the oracle in its CORRECTLY-ROUNDED transcendental mode (oracle/ref_core.h trans_mode: sin, cos,
atan2, atanh, cosh, exp evaluated in f64 and rounded once, exactly as the device evaluates them), so the two sides run
the same arithmetic and are compared BIT FOR BIT per (pixel, sample): the film after rendering ONE sample index holds
(double)weight * (double)rgb of that sample, so equal f64 bits == equal sample.  Bar: >= 99.99 % of the samples
bit-equal (the residue: two f64 libms may round a value differently for ~2^-29 of the arguments, which can flip a
discrete outcome); the measured residual is printed.  The faithful (glibc f32) mode keeps its statistical bar in
test_render_gpu.py.
"""
import numpy as np
import pytest

import helpers
import test_sphere_prefilter as tsp

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ spheres
@pytest.mark.parametrize("flat_max", ["64", "0"])
def test_partial_rotated_scaled_spheres_bit_exact(pm, ora, monkeypatch, flat_max):
    """shapes/sphere.rs:320-368: z / phi clipping and the retry with t1, under rotations and non-uniform scales.  Disjoint
    spheres (no order dependence) => ids, t bit-identical; (u,v) to 1e-5 (acos of different libms)."""
    monkeypatch.setenv("PBRT_MI355_FLAT_MAX", flat_max)
    rng = np.random.default_rng(21)
    prims, centres, radii = [], [], []
    while len(prims) < 24:
        p, c, r = tsp._spheres(pm, rng, 1, 8.0, True)
        if all(np.linalg.norm(c[0] - c2) > r[0] + r2 + 0.05 for c2, r2 in zip(centres, radii)):
            prims += p
            centres.append(c[0])
            radii.append(r[0])
    assert sum(1 for p in prims if p.phi_max < 360.0) >= 5
    agg = pm.GpuAggregate(prims)
    assert (agg.info.flat_prims != 0) == (flat_max != "0")

    class SD:
        pass
    sd = SD()
    sd.prims = prims
    with ora.correctly_rounded():
        sc = helpers.oracle_scene_from(sd, ora)
        rays = tsp._rays(pm, rng, np.array(centres), np.array(radii), 400_000)
        ref, _ = sc.intersect(rays)
        occ_ref, _ = sc.intersect_p(rays)
    gpu = agg.intersect(rays)
    hit = ref["prim_id"] != MISS
    assert 0.2 < hit.mean() < 0.98
    assert np.array_equal(gpu["prim_id"], ref["prim_id"]), f"{(gpu['prim_id'] != ref['prim_id']).sum()} id mismatches"
    assert np.array_equal(gpu["t"][hit].view(np.uint32), ref["t"][hit].view(np.uint32))
    np.testing.assert_allclose(gpu["u"][hit], ref["u"][hit], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(gpu["v"][hit], ref["v"][hit], rtol=1e-5, atol=2e-6)
    assert np.array_equal(agg.intersect_p(rays).astype(np.uint8), occ_ref)


def test_large_mixed_scene_closest_and_any(pm, ora):
    """the MIXED tree-walk kernel variant (spheres + patches + triangles in one tree) at scale: 65 536 triangles, 300
    spheres (a third of them partial), 256 bilinear patches; primary + incoherent rays; closest and any hit."""
    rng = np.random.default_rng(5)
    pos, idx = pm.scenes.blob_mesh(64, 1024, seed=6, blob_sigma=0.12, tri_size=0.02)
    prims = [pm.TriangleMesh(pos, idx)]
    sph_ids = []
    next_id = len(idx)
    for k in range(300):
        r = float(rng.uniform(0.01, 0.06))
        c = rng.uniform(-1.0, 1.0, 3)
        m = np.eye(4)
        m[:3, :3] = tsp._rot(rng)
        m[:3, 3] = c
        if k % 3 == 0:
            zmin, zmax, phi = -0.6 * r, 0.8 * r, 250.0
        else:
            zmin, zmax, phi = -r, r, 360.0
        prims.append(pm.Sphere(r, zmin, zmax, phi, m.astype(np.float32), np.linalg.inv(m).astype(np.float32)))
        sph_ids.append(next_id)
        next_id += 1
    g = 17
    xs = np.linspace(-1.0, 1.0, g, dtype=np.float32)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    Z = (0.3 * np.sin(3 * X) * np.cos(2 * Y) + 1.2).astype(np.float32)        # a non-planar sheet behind the blobs
    ppos = np.stack([X.reshape(-1), Y.reshape(-1), Z.reshape(-1)], axis=1)
    pidx = [[j * g + i, j * g + i + 1, (j + 1) * g + i, (j + 1) * g + i + 1] for j in range(g - 1) for i in range(g - 1)]
    prims.append(pm.BilinearPatchMesh(ppos, pidx))
    agg = pm.GpuAggregate(prims)
    assert agg.info.flat_prims == 0 and agg.info.n_prims == 65536 + 300 + 256

    class SD:
        pass
    sd = SD()
    sd.prims = prims
    with ora.correctly_rounded():
        sc = helpers.oracle_scene_from(sd, ora)
        for rays in (pm.scenes.primary_rays(512, 512), pm.scenes.incoherent_rays(300_000, seed=8)):
            ref, _ = sc.intersect(rays)
            gpu = agg.intersect(rays)
            hit = ref["prim_id"] != MISS
            same = gpu["prim_id"] == ref["prim_id"]
            # documented tie classes only (equal t on two primitives; overlapping spheres within an interval width)
            assert (~same).sum() <= 1e-5 * len(rays) + 2, f"{(~same).sum()} id mismatches"
            ok = same & hit
            assert np.array_equal(gpu["t"][ok].view(np.uint32), ref["t"][ok].view(np.uint32))
            np.testing.assert_allclose(gpu["t"][~same & hit], ref["t"][~same & hit], rtol=1e-5)
            nonsph = ok & ~np.isin(ref["prim_id"], np.asarray(sph_ids, np.uint32))
            assert np.array_equal(gpu["u"][nonsph].view(np.uint32), ref["u"][nonsph].view(np.uint32))
            assert np.array_equal(gpu["v"][nonsph].view(np.uint32), ref["v"][nonsph].view(np.uint32))
            assert (np.isin(ref["prim_id"][hit], np.asarray(sph_ids, np.uint32))).sum() > 500          # spheres do get hit
            assert (ref["prim_id"][hit] >= 65536 + 300).sum() > 500                                      # and patches
            sh = pm.scenes.shadow_rays(rays, ref)
            occ_ref, _ = sc.intersect_p(sh)
            occ = agg.intersect_p(sh).astype(np.uint8)
            assert (occ != occ_ref).sum() <= 1e-5 * len(rays) + 2


# ------------------------------------------------------------------------------------------------ sampler
def test_raw_sampler_stream_bit_exact(pm, ora):
    """IndependentSampler: murmur3 pixel hash + PCG32 + the [0, 1 - 2^-23] float map, drawn ON THE DEVICE, against the
    oracle's stream (which the published PCG / MurmurHash3 vectors pin): 64 uniforms for a spread of pixels, sample indices,
    dimensions and seeds, bit for bit."""
    sd = pm.scenes.main_rs_scene((16, 16))
    integ = pm.WavefrontIntegrator(5, True, True, sd, 4)
    L = ora.lib()
    rng = np.random.default_rng(0)
    cases = [(0, 0, 0, 0, 0), (399, 399, 0, 255, 0), (-3, 7, 0, 1, 5), (2147483647, -2147483648, 0xFFFFFFFF, (1 << 40) + 3, 65535)]
    for _ in range(60):
        cases.append((int(rng.integers(-5000, 5000)), int(rng.integers(-5000, 5000)), int(rng.integers(0, 2 ** 32)),
                      int(rng.integers(0, 2 ** 20)), int(rng.integers(0, 100))))
    for px, py, seed, si, dim in cases:
        g = integ.aggregate.debug_uniforms(px, py, seed, si, dim, 64)
        o = np.zeros(64, np.float32)
        L.ora_sampler_stream(px, py, seed, si, dim, 64, o.ctypes.data)
        assert np.array_equal(g.view(np.uint32), o.view(np.uint32)), (px, py, seed, si, dim)
        assert g.max() <= np.float32(1.0) - np.float32(2.0 ** -23) and g.min() >= 0.0


# ------------------------------------------------------------------------------------------------ render branches
def _bit_equal(pm, ora, sd, spp, max_depth=5, sample_lights=True, sample_bsdf=True, integrator=None):
    """fraction of (pixel, sample) pairs whose film contribution is bit-identical between GPU and oracle (cr mode)"""
    kw = {} if integrator is None else {"integrator": integrator}
    integ = pm.WavefrontIntegrator(max_depth, sample_lights, sample_bsdf, sd, spp, **kw)
    eq = tot = 0
    worst = 0.0
    with ora.correctly_rounded():
        oren = helpers.oracle_renderer_from(sd, ora)
        for s in range(spp):
            p = helpers.render_params(pm, sd, spp, max_depth, sample_lights, sample_bsdf, samples=(s, s + 1), integrator=integrator)
            integ.clear()
            integ.render(samples=(s, s + 1))
            oren.clear()
            oren.render(p, 8)
            g, o = integ.film(), oren.film()
            assert np.isfinite(g).all()
            same = (g.view(np.uint64) == o.view(np.uint64)).all(axis=-1)
            eq += int(same.sum())
            tot += same.size
            assert np.array_equal(g[..., 3].view(np.uint64), o[..., 3].view(np.uint64)), "filter weights differ"
    return eq / tot, tot


def _report(name, frac, tot):
    print(f"[bit-parity] {name}: {frac * 100:.4f} % of {tot} samples bit-equal ({round((1 - frac) * tot)} differ)")


def test_reference_scene_bitwise(pm, ora):
    res, spp = (96, 96), 8
    frac, tot = _bit_equal(pm, ora, pm.scenes.main_rs_scene(res), spp)
    _report("main.rs scene, SimplePath depth 5", frac, tot)
    assert frac >= 0.9999


def test_cornell_bitwise(pm, ora):
    res, spp = (64, 64), 8
    frac, tot = _bit_equal(pm, ora, pm.scenes.cornell_scene(res), spp)
    _report("Cornell box (flat sweep)", frac, tot)
    assert frac >= 0.9999


@pytest.mark.parametrize("ur,vr,remap", [(0.3, 0.3, False), (0.05, 0.4, False), (0.25, 0.1, True)])
def test_rough_dielectric_bitwise(pm, ora, ur, vr, remap):
    """reflection/dielectric.rs:109-413: Trowbridge-Reitz VNDF sampling, generalised half vector, eval + pdf (NEE evaluates
    the rough BSDF against both lights); isotropic, anisotropic and remapped (sqrt) roughness (materials/material.rs:168-195)"""
    res, spp = (64, 64), 8
    sd = pm.scenes.main_rs_scene(res)
    sd.materials[0] = pm.host.dielectric(pm.host.constant_spectrum(1.5), ur, vr, remap)
    frac, tot = _bit_equal(pm, ora, sd, spp)
    _report(f"rough dielectric ({ur}, {vr}, remap={remap})", frac, tot)
    assert frac >= 0.9999
    # the rough branch was taken: the image differs from the smooth one
    a = pm.WavefrontIntegrator(5, True, True, sd, 4)
    a.render()
    b = pm.WavefrontIntegrator(5, True, True, pm.scenes.main_rs_scene(res), 4)
    b.render()
    assert helpers.rel_rmse(a.image(), b.image()) > 2e-2


def test_dispersive_eta_terminates_secondary_wavelengths(pm, ora):
    """a wavelength-dependent eta (DenselySampled spectrum) => SampledWavelengths::terminate_secondary
    (spectrum.rs:893-905, material.rs:176-184): pdf[0] /= 4, pdf[1..] = 0"""
    res, spp = (64, 64), 8
    sd = pm.scenes.main_rs_scene(res)
    lam = np.arange(360, 831, dtype=np.float32)
    eta = (1.45 + 0.08 * (830.0 - lam) / 470.0).astype(np.float32)
    for ur in (0.0, 0.2):
        sd.materials[0] = pm.host.dielectric(pm.host.dense_spectrum(eta), ur, ur, False)
        frac, tot = _bit_equal(pm, ora, sd, spp)
        _report(f"dispersive eta, roughness {ur}", frac, tot)
        assert frac >= 0.9999


def test_thin_lens_bitwise(pm, ora):
    """camera/perspective.rs:143-160: lens_radius > 0 (concentric disk sample of the lens, focal plane at 5)"""
    res, spp = (64, 64), 8
    sd = pm.scenes.main_rs_scene(res)
    cfw = pm.host.Transform.look_at((3.0, 4.0, 1.5), (0.5, 0.5, 0.0), (0.0, 0.0, 1.0))
    sd.camera, _ = pm.host.make_camera(cfw.inverse(), 45.0, res, lens_radius=0.15, focal_distance=5.0)
    frac, tot = _bit_equal(pm, ora, sd, spp)
    _report("thin lens", frac, tot)
    assert frac >= 0.9999
    a = pm.WavefrontIntegrator(5, True, True, sd, 4)
    a.render()
    b = pm.WavefrontIntegrator(5, True, True, pm.scenes.main_rs_scene(res), 4)
    b.render()
    assert helpers.rel_rmse(a.image(), b.image()) > 2e-2          # depth of field is visible


@pytest.mark.parametrize("kind,radius", [("box", (0.5, 0.5)), ("triangle", (2.0, 1.5))])
def test_box_and_triangle_filters_bitwise(pm, ora, kind, radius):
    """imaging/filter.rs:43-125: BoxFilter / TriangleFilter sample (lerp / sample_tent), weight 1"""
    res, spp = (48, 48), 8
    sd = pm.scenes.main_rs_scene(res)
    k = pm.host.PM_FILTER_BOX if kind == "box" else pm._capi.PM_FILTER_TRIANGLE
    sd.film = pm.host.make_film(pm.host.Spectra(), res, filter_kind=k, filter_radius=radius)
    frac, tot = _bit_equal(pm, ora, sd, spp)
    _report(f"{kind} filter", frac, tot)
    assert frac >= 0.9999


def test_max_component_value_clamp_bitwise(pm, ora):
    """camera/film.rs:160-192: rgb scaled by max_component_value / max(rgb) when it exceeds it"""
    res, spp = (48, 48), 8
    sd = pm.scenes.main_rs_scene(res)
    sd.film = pm.host.make_film(pm.host.Spectra(), res, max_component_value=0.35)
    frac, tot = _bit_equal(pm, ora, sd, spp)
    _report("max_component_value = 0.35", frac, tot)
    assert frac >= 0.9999
    a = pm.WavefrontIntegrator(5, True, True, sd, 4)
    a.render()
    b = pm.WavefrontIntegrator(5, True, True, pm.scenes.main_rs_scene(res), 4)
    b.render()
    fa, fb = a.film(), b.film()
    clamped = (fa[..., :3] != fb[..., :3]).any(axis=-1).mean()
    assert clamped > 0.05, clamped                                 # the clamp bites on a visible share of the pixels


def test_other_integrator_branches_bitwise(pm, ora):
    res, spp = (48, 48), 8
    sd = pm.scenes.main_rs_scene(res)
    for sl, sb in ((False, True), (True, False)):
        frac, tot = _bit_equal(pm, ora, sd, spp, max_depth=3, sample_lights=sl, sample_bsdf=sb)
        _report(f"sample_lights={sl} sample_bsdf={sb}", frac, tot)
        assert frac >= 0.9999
    frac, tot = _bit_equal(pm, ora, sd, spp, max_depth=4, integrator=pm._capi.PM_INTEGRATOR_RANDOM_WALK)
    _report("RandomWalk depth 4", frac, tot)
    assert frac >= 0.9999
    frac, tot = _bit_equal(pm, ora, pm.scenes.shading_normal_scene(res), spp)
    _report("vertex-normal patch mesh", frac, tot)
    assert frac >= 0.9999


def test_c4_shaped_scene_bitwise(pm, ora):
    """BASELINE.json configs[3] shape (blobs of triangles under a uniform sky; deep tree walk, infinite-light NEE) at a size
    the oracle finishes in seconds, host SAH tree and device-built LBVH: bit-equal samples"""
    res, spp = (96, 54), 4
    sd = pm.scenes.blob_sky_scene(16, res, blob_sigma=0.25, tri_size=0.06)
    frac, tot = _bit_equal(pm, ora, sd, spp)
    _report("C4-shaped, 16 384 triangles", frac, tot)
    assert frac >= 0.9999


def test_checkerboard_texture_bitwise(pm, ora):
    """the checkerboard extension without vertex uv (st = the patch's own (u, v)), offsets and anisotropic scales, on a
    multi-patch mesh with an rgb / constant pair of reflectances"""
    res, spp = (64, 64), 8
    sd = pm.scenes.shading_normal_scene(res)
    sp = pm.host.Spectra()
    sd.materials[1] = pm.host.checkerboard_diffuse(sp.rgb_albedo((0.8, 0.3, 0.2)), pm.host.constant_spectrum(0.15), 3.0, 5.0, 0.25, -0.4)
    frac, tot = _bit_equal(pm, ora, sd, spp)
    _report("checkerboard on a patch mesh without vertex uv", frac, tot)
    assert frac >= 0.9999
