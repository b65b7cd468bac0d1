"""pbrt-rs_amd/csrc/pm_crmath.hpp (the wavelength sampler's f64 atanh / cosh, written out for the device) against glibc:
every argument the sampler can produce - u = j * 2^-23 for all j < 2^23, sampling/spectrum.rs:1004-1018 - must round to the
same f32 as the library function.  Host build of the same header (tests/native/crmath_check.cpp); the GPU side of the same
claim is tests/test_render_gpu.py's raw wavelength comparison."""
import math
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_sampler_argument_rounds_like_glibc(tmp_path):
    exe = tmp_path / "crmath_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "crmath_check.cpp")], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=300).stdout.split()
    calls, mism_atanh, mism_cosh = int(out[0]), int(out[1]), int(out[2])
    worst_atanh, worst_cosh = float(out[3]), float(out[4])
    assert calls == 1 << 23
    assert mism_atanh == 0 and mism_cosh == 0
    assert worst_atanh < 4.5e-16 and worst_cosh < 4.5e-16       # 2 ulp of f64


def test_taylor_coefficients_are_reciprocal_factorials():
    src = open(os.path.join(ROOT, "pbrt-rs_amd", "csrc", "pm_crmath.hpp")).read()
    body = src[src.index("PM_CR_FN double pm_cosh_small"):src.index("PM_CR_FN void pm_sincos_2pi")]
    found = {int(n): float(v) for v, n in re.findall(r"([0-9.]+e-[0-9]+)\)?;?\s*// 1/([0-9]+)!", body)}
    assert set(found) == {26, 24, 22, 20, 18, 16, 14, 12, 10, 8, 6, 4}
    for n, v in found.items():
        assert abs(v * math.factorial(n) - 1.0) < 1e-15, n


def _check2(tmp_path, *args):
    exe = tmp_path / "crmath_check2"
    if not exe.exists():
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-pthread", "-o", str(exe),
                        os.path.join(ROOT, "tests", "native", "crmath_check2.cpp")], check=True)
    out = subprocess.run([str(exe), *args], check=True, capture_output=True, text=True, timeout=600).stdout.split()
    return int(out[0]), int(out[1]), float(out[2])


def test_sincos_exp_log_round_like_glibc(tmp_path):
    """sincos: every azimuth 2 pi u the samplers draw.  exp / log: every 251st f32 of [-104, 89] / of the positive floats
    (the exhaustive runs - `crmath_check2 exp 1`, `crmath_check2 log 1`, 2.24e9 and 2.14e9 arguments - had no mismatch either)."""
    calls, mism, worst = _check2(tmp_path, "sincos")
    assert calls == 1 << 23 and mism == 0 and worst < 4.5e-16
    for what in ("exp", "log"):
        calls, mism, worst = _check2(tmp_path, what, "251")
        assert calls > 8_000_000 and mism == 0 and worst < 4.5e-16, what
