"""PiecewiseLinearSpectrum (the one SpectrumEnum variant that had no general constructor in the host API before round 3;
/root/reference/src/sampling/spectrum.rs:252-357): pmh_piecewise_linear_table against the oracle's independent restatement
(oracle/ref_host.h), bit for bit, and against the definition (linear interpolation, range extension, luminance normalisation)."""
import ctypes as C
import importlib

import numpy as np
import pytest


def _ora_table(ora, il, normalize, cie_y):
    L = ora.lib()
    L.ora_host_piecewise_linear_table.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    out = np.zeros(471, np.float32)
    L.ora_host_piecewise_linear_table(il.ctypes.data, len(il), int(normalize), cie_y.ctypes.data, out.ctypes.data)
    return out


def test_matches_the_oracle_bit_for_bit_and_the_definition(pm, ora):
    sp = pm.host.Spectra()
    rng = np.random.default_rng(3)
    for trial in range(200):
        n = int(rng.integers(1, 40))
        lo, hi = (300.0, 900.0) if trial % 3 == 0 else ((400.0, 700.0) if trial % 3 == 1 else (360.0, 830.0))
        lam = np.sort(rng.uniform(lo, hi, n)).astype(np.float32)
        if trial % 3 == 2:
            lam[0], lam[-1] = 360.0, 830.0
        if len(np.unique(lam)) != n:
            continue
        val = rng.uniform(0.0, 2.0, n).astype(np.float32)
        il = np.stack([lam, val], axis=1).reshape(-1).astype(np.float32)
        for normalize in (False, True):
            got = sp.piecewise_linear_table(il, normalize)
            want = _ora_table(ora, il, normalize, sp.cie_y)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (trial, normalize)
        # the definition: inside [lam0, lamN] linear interpolation; a spectrum that starts above 360 nm is extended flat down to
        # 359 nm, one that ends below 830 nm flat up to 831 nm (spectrum.rs:281-300); outside the samples it is zero
        t = sp.piecewise_linear_table(il, False).astype(np.float64)
        grid = np.arange(360, 831, dtype=np.float64)
        l64, v64 = lam.astype(np.float64), val.astype(np.float64)
        ext_l = np.concatenate([[359.0] if l64[0] > 360 else [], l64, [831.0] if l64[-1] < 830 else []])
        ext_v = np.concatenate([[v64[0]] if l64[0] > 360 else [], v64, [v64[-1]] if l64[-1] < 830 else []])
        ref = np.where((grid < ext_l[0]) | (grid > ext_l[-1]), 0.0, np.interp(grid, ext_l, ext_v))
        assert np.allclose(t, ref, rtol=2e-5, atol=2e-6)
        tn = sp.piecewise_linear_table(il, True)
        if ref @ sp.cie_y.astype(np.float64) > 1e-3:
            assert sp.photometric(tn) == pytest.approx(106.856895, rel=1e-4)


def test_d65_is_the_normalised_piecewise_linear_spectrum(pm, ora):
    sp = pm.host.Spectra()
    assert sp.photometric(sp.d65) == pytest.approx(106.856895, rel=1e-5)
    # two samples reproduce a straight line
    t = sp.piecewise_linear_table([360.0, 0.0, 830.0, 470.0])
    assert np.allclose(t, np.arange(471), rtol=1e-6, atol=1e-4)


def test_inputs_upstream_asserts_on_are_rejected(pm):
    sp = pm.host.Spectra()
    for bad in ([400.0, 1.0, 400.0, 2.0], [500.0, 1.0, 400.0, 2.0], [400.0, 1.0, 500.0], []):
        with pytest.raises(ValueError):
            sp.piecewise_linear_table(bad)


def test_scene_files_take_sampled_spectra(pm):
    ps = pm.pbrt_scene
    text = '''Camera "perspective" "float fov" 40
    WorldBegin
    LightSource "point" "spectrum I" [400 1 500 2 600 0.5 700 1]
    Material "diffuse" "spectrum reflectance" [360 0.1 830 0.9]
    Shape "sphere" "float radius" 0.5
    Material "dielectric" "spectrum eta" [360 1.55 830 1.45]
    Shape "sphere" "float radius" 0.25'''
    sd, _ = ps.load_pbrt(text)
    assert sd.materials[0]["spectrum"]["kind"] == pm.host.PM_SPEC_DENSE
    assert np.allclose(sd.materials[0]["spectrum"]["table"][[0, 470]], [0.1, 0.9], atol=1e-6)
    assert sd.materials[1]["spectrum"]["kind"] == pm.host.PM_SPEC_DENSE and sd.materials[1]["spectrum"]["table"][0] == np.float32(1.55)
    assert sd.lights[0]["table"][140] == pytest.approx(2.0)           # 500 nm
    with pytest.raises(ps.PbrtSceneError):
        ps.load_pbrt(text.replace("[360 0.1 830 0.9]", "[500 0.1 400 0.9]"))
