"""shared test helpers: build the oracle's scene/renderer from a product SceneDescription"""
import ctypes as C
import importlib

import numpy as np


def oracle_scene_from(sd, ora, split=None):
    pmod = importlib.import_module("pbrt-rs_amd")
    sc = ora.Scene()
    for p in sd.prims:
        if isinstance(p, pmod.Sphere):
            sc.add_sphere(p.radius, p.z_min, p.z_max, p.phi_max, p.render_from_object, p.object_from_render, p.reverse_orientation)
        elif isinstance(p, pmod.BilinearPatchMesh):
            sc.add_patch_mesh(p.positions, p.indices, p.reverse_orientation, p.transform_swaps_handedness, normals=p.normals, uv=p.uv)
        else:
            sc.add_triangle_mesh(p.positions, p.indices)
    sc.build(split if split is not None else ora.SPLIT_MIDDLE, 255)
    return sc


def oracle_renderer_from(sd, ora, split=None):
    sc = oracle_scene_from(sd, ora, split)
    mats, lights, film = sd.c_materials(), sd.c_lights(), sd.c_film()
    r = ora.Renderer(sc, mats, len(sd.materials), sd.c_prim_materials(), lights, len(sd.lights), sd.camera, film)
    r._keep = (sc, mats, lights, film, sd)
    if getattr(sd, "medium", None) is not None:
        m = sd.medium
        r.set_medium(m["sigma_a"], m["sigma_s"], m["g"], m["bounds"])
    return r


def render_params(pmod, sd, spp, max_depth=5, sample_lights=True, sample_bsdf=True, seed=0, window=None, samples=None,
                  integrator=None):
    p = pmod._capi.PmRenderParams()
    p.integrator = pmod._capi.PM_INTEGRATOR_SIMPLE_PATH if integrator is None else integrator
    p.max_depth, p.sample_lights, p.sample_bsdf, p.spp, p.seed = max_depth, int(sample_lights), int(sample_bsdf), spp, seed
    p.window[:] = list(window if window is not None else sd.film["pixel_bounds"])
    p.s0, p.s1 = samples if samples is not None else (0, spp)
    return p


def rel_rmse(img, ref):
    """per-pixel relative RMSE: sqrt(mean((a-b)^2 / (b^2 + eps)))"""
    a, b = img.astype(np.float64), ref.astype(np.float64)
    eps = 1e-4 * float(np.mean(b * b)) + 1e-12
    return float(np.sqrt(np.mean((a - b) ** 2 / (b * b + eps))))
