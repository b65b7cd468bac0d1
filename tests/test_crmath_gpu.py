"""The device's f64-then-round transcendentals (csrc/pm_crmath.hpp, pm_shade.hpp cr_*) against glibc's f64 functions rounded to
f32, through the pm_debug_math hook of the C ABI.  tests/test_crmath_cpu.py checks the same header compiled for the host over
every argument; this shows that the DEVICE evaluates it to the same bits (fma, ldexp, correctly rounded f64 division)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ref(fn, x):
    with np.errstate(over="ignore"):        # (exp beyond the f32 range rounds to inf, as on the device)
        return np.array([fn(float(v)) for v in x], np.float64).astype(np.float32)


def test_device_transcendentals_round_like_glibc(pm):
    agg = pm.GpuAggregate([pm.TriangleMesh(np.eye(3, dtype=np.float32), np.array([[0, 1, 2]], np.uint32))])
    rng = np.random.default_rng(5)
    n = 200_000
    f32 = np.float32
    u = (rng.integers(0, 1 << 23, n).astype(np.float32) * f32(2.0 ** -23)).astype(np.float32)     # the sampler's uniforms
    # wavelength sampling (sampling/spectrum.rs:1004-1018): atanh argument and cosh argument as the kernels form them
    xa = (f32(0.85691062) - f32(1.827502) * u).astype(np.float32)
    got = agg.debug_math(0, xa)
    assert np.array_equal(got.view(np.uint32), _ref(math.atanh, xa).view(np.uint32))
    lam = (f32(538.0) - f32(138.88889) * got).astype(np.float32)
    lam = lam[(lam >= 360.0) & (lam <= 830.0)]
    xc = (f32(0.0072) * (lam - f32(538.0))).astype(np.float32)
    assert np.array_equal(agg.debug_math(1, xc).view(np.uint32), _ref(math.cosh, xc).view(np.uint32))
    # azimuths 2 pi u
    phi = (f32(2.0) * f32(math.pi) * u).astype(np.float32)
    assert np.array_equal(agg.debug_math(4, phi).view(np.uint32), _ref(math.sin, phi).view(np.uint32))
    assert np.array_equal(agg.debug_math(5, phi).view(np.uint32), _ref(math.cos, phi).view(np.uint32))
    # the concentric disk's angle in [-pi/4, 3 pi/4]
    th = rng.uniform(-math.pi / 4, 3 * math.pi / 4, n).astype(np.float32)
    assert np.array_equal(agg.debug_math(6, th).view(np.uint32), _ref(math.sin, th).view(np.uint32))
    assert np.array_equal(agg.debug_math(7, th).view(np.uint32), _ref(math.cos, th).view(np.uint32))
    # exp: transmittance arguments -sigma d over many magnitudes, the f32 underflow range, small positive ones; specials
    xe = np.concatenate([-np.exp(rng.uniform(-30, 4.6, n)), rng.uniform(-104, -80, n // 4), rng.uniform(0, 88.7, n // 4),
                         [0.0, -0.0, -103.9, -104.5, -1e30, 88.9, 1e30]]).astype(np.float32)

    def exp_(v):
        try:
            return math.exp(v)
        except OverflowError:
            return math.inf
    assert np.array_equal(agg.debug_math(2, xe).view(np.uint32), _ref(exp_, xe).view(np.uint32))
    # log: 1 - u (free-flight sampling) and positive floats of every magnitude incl. denormals
    bits = rng.integers(1, 0x7F7FFFFF, n).astype(np.uint32)
    xl = np.concatenate([(f32(1.0) - u).astype(np.float32), bits.view(np.float32)])
    xl = xl[xl > 0]
    assert np.array_equal(agg.debug_math(3, xl).view(np.uint32), _ref(math.log, xl).view(np.uint32))
