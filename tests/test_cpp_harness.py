"""The C++ host mirror (include/pbrt_mi355.hpp) and its example (examples/main_rs.cpp: the reference binary's scene):
it builds with plain g++ against the C ABI, fails loudly without a GPU, and on a GPU renders the same film as the
Python-driven path (same C ABI underneath; the two hosts build the scene independently)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "main_rs")


def test_example_builds_and_needs_a_gpu(tmp_path):
    import torch
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    assert os.path.exists(EXE)
    if not torch.cuda.is_available():
        r = subprocess.run([EXE, "8", "8", "1", str(tmp_path / "x.exr")], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_and_python_hosts_render_the_same_film(pm, tmp_path):
    w, h, spp = 64, 48, 8
    film_path, exr_path = tmp_path / "film.f64", tmp_path / "img.exr"
    r = subprocess.run([EXE, str(w), str(h), str(spp), str(exr_path), str(film_path)], cwd=ROOT, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    cpp = np.fromfile(film_path, np.float64).reshape(h, w, 4)
    integ = pm.WavefrontIntegrator(5, True, True, pm.scenes.main_rs_scene((w, h)), spp)
    integ.render()
    py = integ.film()
    np.testing.assert_allclose(cpp, py, rtol=1e-6, atol=1e-9)
    from test_host_logic import _read_exr
    np.testing.assert_allclose(_read_exr(exr_path), integ.image(), rtol=1e-5, atol=1e-7)
    # the reference's wave schedule through the C++ mirror: progress per wave, the same film bit for bit
    film2 = tmp_path / "film_waves.f64"
    r = subprocess.run([EXE, str(w), str(h), str(spp), str(tmp_path / "img2.exr"), str(film2), "pbrt-rs_amd/data/cie_tables.bin", "waves"],
                       cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert [l for l in r.stderr.splitlines() if l.startswith("wave [")] == [f"wave [{a}, {b}) of {spp} spp done" for a, b in ((0, 1), (1, 2), (2, 4), (4, 8))]
    assert np.array_equal(np.fromfile(film2, np.float64).view(np.uint64), np.fromfile(film_path, np.float64).view(np.uint64))
