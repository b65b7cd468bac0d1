"""Per-wave table culling for camera rays (pm_device.hpp trace_flat, COH; round 3): the first closest-hit launch of a pass of a
flat-swept scene sweeps, per 8 x 8 pixel tile, only the table entries some ray of the tile can reach.  Culling may never change a
hit: the film with PBRT_MI355_COH=1 (default) must equal the film with PBRT_MI355_COH=0 bit for bit - on cameras INSIDE a cloud of
triangles (vertices behind the origin, beside it, triangles through the camera plane, triangles a hair in front of the eye), on
ragged windows (dead lanes in a tile), with a thin lens (origins differ: the wave keeps the whole table), and on the C2 / C5 / glass
sphere boxes.  (That the un-culled sweep is right is what the oracle parity tests establish.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud_scene(pm, seed, res, lens):
    rng = np.random.default_rng(seed)
    host = pm.host
    sp = host.Spectra()
    sd = host.SceneDescription()
    sd.film = host.make_film(sp, res, filter_kind=pm._capi.PM_FILTER_GAUSSIAN, filter_radius=(1.5, 1.5), sigma=0.5)
    eye = rng.normal(size=3) * 0.3
    c2w = host.Transform.look_at(eye, eye + rng.normal(size=3), (0.0, 0.0, 1.0)).inverse()
    sd.camera, rfw = host.make_camera(c2w, float(rng.uniform(25, 110)), res, lens_radius=lens, focal_distance=2.0)
    sd.render_from_world = rfw
    sd.materials = [host.diffuse(sp.rgb_albedo(tuple(rng.uniform(0.2, 0.9, 3)))) for _ in range(3)]
    n = int(rng.integers(20, 61))
    centre = rng.normal(size=(n, 3)) * rng.choice([0.05, 0.5, 2.0, 8.0], size=(n, 1))      # some a hair from the eye, some far
    size = rng.choice([0.02, 0.3, 2.0, 20.0], size=(n, 1, 1))
    verts = (eye + centre[:, None, :] + rng.normal(size=(n, 3, 3)) * size).astype(np.float32).reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    sd.prims = [pm.TriangleMesh(pm.scenes._xf_points(rfw.m, verts), idx)]
    sd.prim_materials = [int(rng.integers(0, 3)) for _ in range(n)]
    sd.lights = [host.light(host.PM_LIGHT_UNIFORM_INFINITE, sp.d65.copy(), np.float32(1.0) / np.float32(sp.photometric(sp.d65)))]
    return sd


def _film(pm, monkeypatch, integ, coh, **kw):
    monkeypatch.setenv("PBRT_MI355_COH", coh)
    integ.clear()
    st = integ.render(**kw)
    return integ.film().copy(), st


@pytest.mark.parametrize("seed", range(12))
def test_culled_sweep_equals_full_sweep_inside_a_triangle_cloud(pm, monkeypatch, seed):
    lens = 0.0 if seed % 4 else 0.05
    sd = _cloud_scene(pm, seed, (72, 52), lens)          # 52 rows: ragged tiles at the bottom
    integ = pm.WavefrontIntegrator(3, True, True, sd, 4, seed=seed)
    assert integ.aggregate.info.flat_prims > 0
    off, st0 = _film(pm, monkeypatch, integ, "0")
    on, st1 = _film(pm, monkeypatch, integ, "1")
    assert np.array_equal(on.view(np.uint64), off.view(np.uint64))
    assert st0["closest_rays"] == st1["closest_rays"] and st0["shadow_rays"] == st1["shadow_rays"]
    assert (off[..., :3] > 0).any()
    win = (5, 3, 61, 47)                                  # a window that starts and ends inside tiles
    a, _ = _film(pm, monkeypatch, integ, "0", window=win)
    b, _ = _film(pm, monkeypatch, integ, "1", window=win)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_culled_sweep_on_the_boxes(pm, monkeypatch):
    C = pm._capi
    for sd, depth, integrator in ((pm.scenes.c2_scene((200, 136)), 5, C.PM_INTEGRATOR_SIMPLE_PATH),
                                  (pm.scenes.cornell_scene((136, 200)), 5, C.PM_INTEGRATOR_SIMPLE_PATH),
                                  (pm.scenes.c5_scene((96, 96)), 8, C.PM_INTEGRATOR_VOL_PATH),
                                  (pm.scenes.main_rs_scene((120, 80)), 5, C.PM_INTEGRATOR_SIMPLE_PATH)):
        integ = pm.WavefrontIntegrator(depth, True, True, sd, 3, seed=5, integrator=integrator)
        assert integ.aggregate.info.flat_prims > 0
        off, _ = _film(pm, monkeypatch, integ, "0")
        on, _ = _film(pm, monkeypatch, integ, "1")
        assert np.array_equal(on.view(np.uint64), off.view(np.uint64))
