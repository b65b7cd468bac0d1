"""Randomised scenes, GPU vs oracle bit for bit (correctly-rounded mode, per-sample film contributions as in
test_branches_gpu.py).  Each seed draws: a mix of spheres (some clipped / rotated / scaled), bilinear patches (some with
vertex normals) and triangles - few enough for the flat sweep on even seeds, a tree walk on odd ones -, random materials
(diffuse, smooth / rough / dispersive dielectric), random lights (point, directional, infinite, area lights on a patch and on
a triangle), a random camera (sometimes a thin lens), filter, sampling flags, depth and - for every third seed - a medium with
VolPath."""
import numpy as np
import pytest

import test_branches_gpu as tb
import test_sphere_prefilter as tsp

pytestmark = pytest.mark.gpu


def _scene(pm, seed):
    rng = np.random.default_rng(1000 + seed)
    host = pm.host
    sp = host.Spectra()
    res = (48, 40)
    sd = host.SceneDescription()
    filt = [pm._capi.PM_FILTER_GAUSSIAN, pm._capi.PM_FILTER_BOX, pm._capi.PM_FILTER_TRIANGLE][seed % 3]
    sd.film = host.make_film(sp, res, filter_kind=filt, filter_radius=(1.5, 1.5) if filt == pm._capi.PM_FILTER_GAUSSIAN else (0.5 + rng.random(), 0.5 + rng.random()),
                             max_component_value=float("inf") if seed % 4 else 2.0)
    cam_pos = rng.normal(size=3)
    cam_pos[2] = abs(cam_pos[2]) + 0.3            # above the floor
    cam_pos = cam_pos / np.linalg.norm(cam_pos) * rng.uniform(4.0, 6.0)
    c2w = host.Transform.look_at(cam_pos, rng.normal(size=3) * 0.3, (0.0, 0.0, 1.0)).inverse()
    lens = 0.0 if seed % 2 else float(rng.uniform(0.02, 0.1))
    sd.camera, rfw = host.make_camera(c2w, float(rng.uniform(30, 60)), res, screen_window=(-1.2, -1.0, 1.2, 1.0), lens_radius=lens,
                                      focal_distance=float(np.linalg.norm(cam_pos)))
    sd.render_from_world = rfw
    # materials
    lam = np.arange(360, 831, dtype=np.float32)
    eta_tab = (1.45 + 0.1 * (830.0 - lam) / 470.0).astype(np.float32)
    sd.materials = [host.diffuse(sp.rgb_albedo(tuple(rng.uniform(0.1, 0.9, 3)))) for _ in range(3)]
    sd.materials.append(host.dielectric(host.constant_spectrum(1.5)))
    sd.materials.append(host.dielectric(host.constant_spectrum(1.33), float(rng.uniform(0.05, 0.4)), float(rng.uniform(0.05, 0.4)), bool(seed % 2)))
    sd.materials.append(host.dielectric(host.dense_spectrum(eta_tab), 0.0, 0.0, False))
    n_mat = len(sd.materials)
    prims, mats = [], []
    many = seed % 2 == 1

    def add(p, n_faces, mat=None):
        prims.append(p)
        for _ in range(n_faces):
            mats.append(int(rng.integers(0, n_mat)) if mat is None else mat)
    # spheres: disjoint, some partial / rotated / scaled (placed in render space through rfw)
    centres, radii = [], []
    want = 10 if many else 4
    while len(centres) < want:
        p, c, r = tsp._spheres(pm, rng, 1, 2.0, True)
        if all(np.linalg.norm(c[0] - c2) > r[0] + r2 + 0.05 for c2, r2 in zip(centres, radii)):
            s = p[0]
            m = (rfw.m.astype(np.float64) @ s.render_from_object.astype(np.float64)).astype(np.float32)
            add(pm.Sphere(s.radius, s.z_min, s.z_max, s.phi_max, m, np.linalg.inv(m.astype(np.float64)).astype(np.float32), s.reverse_orientation), 1)
            centres.append(c[0])
            radii.append(r[0])
    # a floor of patches with vertex normals, a wavy patch sheet, a triangle blob
    g = 5 if many else 2
    xs = np.linspace(-3.0, 3.0, g + 1, dtype=np.float32)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    world = np.stack([X.reshape(-1), Y.reshape(-1), np.full(X.size, -2.2, np.float32) + 0.1 * np.sin(X.reshape(-1))], axis=1)
    nrm = np.stack([0.2 * np.sin(1.3 * world[:, 0]), 0.2 * np.cos(1.1 * world[:, 1]), np.ones(len(world))], axis=1)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    idx = [[j * (g + 1) + i, j * (g + 1) + i + 1, (j + 1) * (g + 1) + i, (j + 1) * (g + 1) + i + 1] for j in range(g) for i in range(g)]
    add(pm.BilinearPatchMesh(pm.scenes._xf_points(rfw.m, world), idx, normals=nrm if seed % 3 else None), len(idx), mat=0)
    nt = 120 if many else 10
    pos, tidx = pm.scenes.blob_mesh(1, nt, seed=seed, blob_sigma=0.8, tri_size=0.5)
    add(pm.TriangleMesh(pm.scenes._xf_points(rfw.m, pos + np.float32([0.0, 0.0, 0.3])), tidx), len(tidx))
    # emitters: one quad, one triangle
    quad = np.array([[-0.8, -0.8, 2.6], [0.8, -0.7, 2.6], [-0.7, 0.8, 2.5], [0.9, 0.9, 2.7]], np.float32)
    add(pm.BilinearPatchMesh(pm.scenes._xf_points(rfw.m, quad), [[0, 1, 2, 3]]), 1, mat=1)
    quad_id = len(mats) - 1
    tri = np.array([[2.2, -1.0, -1.0], [2.4, 1.0, -0.8], [2.3, 0.0, 1.2]], np.float32)
    add(pm.TriangleMesh(pm.scenes._xf_points(rfw.m, tri), [[0, 1, 2]]), 1, mat=2)
    tri_id = len(mats) - 1
    sd.prims, sd.prim_materials = prims, mats
    # lights
    bb = sp.blackbody_table(float(rng.uniform(2500, 6500)))
    sky, _, _ = sp.rgb_illuminant_table(tuple(rng.uniform(0.2, 0.6, 3)))
    sd.lights = [host.area_light(bb, 4.0 / sp.photometric(bb), quad_id, two_sided=True),
                 host.area_light(sp.d65.copy(), 2.0 / sp.photometric(sp.d65), tri_id, two_sided=bool(seed % 2))]
    if seed % 2 == 0:
        sd.lights.append(host.light(pm._capi.PM_LIGHT_UNIFORM_INFINITE, sky, np.float32(0.5) / np.float32(sp.photometric(sp.d65))))
    pl = host.Transform.translate(tuple(rng.uniform(-1, 1, 3) + np.array([0, 0, 1.5])))
    sd.lights.append(host.light(pm._capi.PM_LIGHT_POINT, bb, 3.0 / sp.photometric(bb), (rfw * pl).m))
    if seed % 3 == 0:
        sun = host.Transform.look_at((0, 0, 0), tuple(rng.normal(size=3)), (0.0, 1.0, 0.1)).inverse()
        sd.lights.append(host.light(pm._capi.PM_LIGHT_DIRECTIONAL, sp.d65.copy(), 0.8 / sp.photometric(sp.d65), (rfw * sun).m))
    if seed % 3 == 2:
        sig_s = (0.05 + 0.1 * (830.0 - lam) / 470.0).astype(np.float32)
        sig_a = np.full(471, 0.02, np.float32)
        lo = pm.scenes._xf_points(rfw.m, np.array([[-2.5, -2.5, -2.0]], np.float32))[0]
        hi = pm.scenes._xf_points(rfw.m, np.array([[2.5, 2.5, 2.4]], np.float32))[0]
        sd.medium = host.homogeneous_medium(sig_a, sig_s, 1.0, float(rng.uniform(-0.5, 0.8)), bounds=tuple(lo) + tuple(hi))
    return sd, many


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PBRT_MI355_FUZZ_SEEDS", "6"))))
def test_random_scene_bitwise(pm, ora, seed):
    sd, many = _scene(pm, seed)
    integrator = pm._capi.PM_INTEGRATOR_VOL_PATH if getattr(sd, "medium", None) is not None else None
    depth = [5, 8, 3][seed % 3]
    frac, tot = tb._bit_equal(pm, ora, sd, 4, max_depth=depth, sample_lights=seed != 4, sample_bsdf=seed != 3, integrator=integrator)
    tb._report(f"random scene {seed} ({'tree walk' if many else 'flat sweep'}, {len(sd.prim_materials)} primitives)", frac, tot)
    assert frac >= 0.9999
