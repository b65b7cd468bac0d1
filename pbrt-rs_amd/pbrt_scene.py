""".pbrt scene ingestion for the subset of the format the hot path can render.

The reference's own parser stops at `todo!()` (src/scene_parsing/construct.rs:57) and its binary hard-codes the scene
(src/main.rs:43-234); this module turns a scene FILE into the same flattened `SceneDescription` the hard-coded builder
in scenes.py produces, following the reference's constructors for each directive (and pbrt-v4's documented defaults
where the reference has none yet):

  LookAt Translate Scale Rotate Identity Transform ConcatTransform ReverseOrientation AttributeBegin/End WorldBegin
  Camera "perspective" (fov, lensradius, focaldistance)      Film "rgb" (xresolution, yresolution)
  Sampler (pixelsamples)   PixelFilter "gaussian" (radius, sigma)   Integrator "simplepath"|"path"|"volpath"|"randomwalk" (maxdepth)
  LightSource "infinite" (L, scale) | "distant" (from, to, L, scale) | "point" (from, I, scale)
  Material "diffuse" (reflectance) | "dielectric" (eta, roughness, uroughness, vroughness, remaproughness)
  Texture <name> "spectrum"|"float" "constant" (value) | "spectrum" "checkerboard" (uscale, vscale, udelta, vdelta, tex1, tex2)
          (other texture classes are rejected; the checkerboard is an extension for scenes/test.pbrt, point-sampled)
  Shape "sphere" (radius, zmin, zmax, phimax) | "bilinearmesh" (P, indices, N, uv) | "trianglemesh" (P, indices)
Substitutions are made LOUDLY: every one is appended to settings["warnings"] (the reference has one sampler and no
MIS / volumetric integrator): Sampler "halton"/"sobol"/... -> independent (same pixelsamples and seed);
Integrator "path" -> simplepath; "volpath" -> VolPath, which without a medium in the file equals simplepath.
Spectrum-valued parameters: "rgb X" [r g b], "blackbody X" T, "float X" v, "spectrum X" v (constant),
"spectrum X" [lambda0 v0 lambda1 v1 ...] (PiecewiseLinearSpectrum::from_interleaved, sampling/spectrum.rs:269-310).
Anything else raises PbrtSceneError naming the directive: nothing is silently dropped.
"""
import re

import numpy as np

from . import host
from .aggregate import BilinearPatchMesh, Sphere, TriangleMesh


class PbrtSceneError(ValueError):
    pass


_TOKEN = re.compile(r'"[^"]*"|\[|\]|[^\s\[\]"#]+|#[^\n]*')


def _tokens(text):
    for m in _TOKEN.finditer(text):
        t = m.group(0)
        if not t.startswith("#"):
            yield t


def _is_number(t):
    try:
        float(t)
        return True
    except ValueError:
        return False


def _literal_kind(t):
    """kind of one atomic literal as the reference's parser classifies it (scene_parsing/common.rs atomic_literal: a number, true /
    false, or a quoted string; a bare word is an error, common.rs:650-654)"""
    if t.startswith('"'):
        return "str"
    if t in ("true", "false"):
        return "bool"
    if _is_number(t):
        return "num"
    raise PbrtSceneError(f"{t} is not a literal (number, true / false or a quoted string)")


def _check_values(typ, pname, vals):
    """an array literal holds literals of ONE kind (common.rs literal_err_mixed_array); "rgb" takes exactly three numbers
    (common.rs rgb_param_wrong_length)"""
    kinds = {_literal_kind(v) for v in vals}
    if len(kinds) > 1:
        raise PbrtSceneError(f'parameter "{typ} {pname}": array literal mixes {" and ".join(sorted(kinds))}')
    if typ == "rgb" and (kinds != {"num"} or len(vals) != 3):
        raise PbrtSceneError(f'parameter "rgb {pname}" needs exactly 3 numbers')


def _parse(text):
    """-> list of (directive, positional args, {param name: (type, values)})"""
    toks = list(_tokens(text))
    out, i = [], 0
    while i < len(toks):
        name = toks[i]
        i += 1
        if name.startswith('"') or name in "[]" or _is_number(name):
            raise PbrtSceneError(f"expected a directive, found {name}")
        args, params = [], {}
        while i < len(toks):
            t = toks[i]
            if t.startswith('"') and " " in t.strip('"') and len(t.strip('"').split()) == 2:
                typ, pname = t.strip('"').split()
                i += 1
                vals = []
                if i < len(toks) and toks[i] == "[":
                    i += 1
                    while toks[i] != "]":
                        vals.append(toks[i])
                        i += 1
                    i += 1
                else:
                    if i >= len(toks):
                        raise PbrtSceneError(f'parameter "{typ} {pname}" has no value')
                    vals.append(toks[i])
                    i += 1
                _check_values(typ, pname, vals)
                params[pname] = (typ, vals)
            elif t.startswith('"') or _is_number(t):
                args.append(t.strip('"') if t.startswith('"') else float(t))
                i += 1
            elif t == "[":
                i += 1
                while toks[i] != "]":
                    args.append(float(toks[i]))
                    i += 1
                i += 1
            else:
                break
        out.append((name, args, params))
    return out


def _floats(p, name, default=None, n=None):
    if name not in p:
        return default
    v = [np.float32(x) for x in p[name][1]]
    if n is not None and len(v) != n:
        raise PbrtSceneError(f'parameter "{name}" needs {n} values')
    return v


def _float(p, name, default):
    v = _floats(p, name)
    return np.float32(default) if v is None else v[0]


def _bool(p, name, default):
    if name not in p:
        return default
    return p[name][1][0].strip('"') == "true"


class _State:
    def __init__(self):
        self.ctm = host.Transform.IDENTITY
        self.material = ("diffuse", {})
        self.reverse = False


def _rotate(angle_deg, axis):
    """Transform::rotate (core/transform.rs:118-150): rotation about an arbitrary axis"""
    a = np.asarray(axis, np.float32)
    a = a / np.sqrt(np.float32(a @ a))
    th = np.float32(np.radians(np.float32(angle_deg)))
    s, c = np.float32(np.sin(th)), np.float32(np.cos(th))
    x, y, z = a
    m = np.eye(4, dtype=np.float32)
    m[0, :3] = [x * x + (1 - x * x) * c, x * y * (1 - c) - z * s, x * z * (1 - c) + y * s]
    m[1, :3] = [x * y * (1 - c) + z * s, y * y + (1 - y * y) * c, y * z * (1 - c) - x * s]
    m[2, :3] = [x * z * (1 - c) - y * s, y * z * (1 - c) + x * s, z * z + (1 - z * z) * c]
    return host.Transform(m, m.T.copy())


def load_pbrt(text, spectra=None):
    """Parse scene text -> (SceneDescription, settings dict with integrator / max_depth / spp / seed)."""
    sp = spectra or host.Spectra()
    sd = host.SceneDescription()
    settings = {"integrator": "simplepath", "max_depth": 5, "spp": 16, "seed": 0, "sample_lights": True, "sample_bsdf": True,
                "warnings": [], "filename": None}
    warn = settings["warnings"].append
    st, stack = _State(), []
    named_textures = {}
    camera = None
    film = {"res": (1280, 720), "filter": ("gaussian", (np.float32(1.5), np.float32(1.5)), np.float32(0.5))}
    pending_lights, pending_shapes = [], []          # need render_from_world, known only after Camera
    in_world = False

    def spectrum_param(p, name):
        if name not in p:
            return None
        typ, vals = p[name]
        if typ == "rgb":
            return ("rgb", [np.float32(v) for v in vals])
        if typ == "blackbody":
            return ("blackbody", np.float32(vals[0]))
        if typ in ("float", "spectrum") and len(vals) == 1 and _is_number(vals[0]):
            return ("const", np.float32(vals[0]))
        if typ == "spectrum" and len(vals) >= 2 and len(vals) % 2 == 0 and all(_is_number(v) for v in vals):
            # "spectrum X" [lambda0 v0 lambda1 v1 ...]: PiecewiseLinearSpectrum::from_interleaved(.., false), densified
            try:
                return ("table", sp.piecewise_linear_table([np.float32(v) for v in vals]))
            except ValueError as e:
                raise PbrtSceneError(f'parameter "spectrum {name}": {e}')
        if typ == "texture":
            tex = named_textures.get(vals[0].strip('"'))
            if tex is None:
                raise PbrtSceneError(f"unknown texture {vals[0]}")
            return tex
        raise PbrtSceneError(f'unsupported spectrum parameter "{typ} {name}"')

    seen_options = set()
    for name, args, p in _parse(text):
        if name in ("Camera", "Sampler", "ColorSpace", "Film", "PixelFilter", "Filter", "Integrator", "Accelerator"):
            # scene_parsing/scene.rs:119-175: each of the options section's directives may appear once (RepeatedDirective;
            # the reference's test_global_options_fail_on_repeat_directive)
            if name in seen_options:
                raise PbrtSceneError(f"repeated directive {name}")
            seen_options.add(name)
        if name == "LookAt":
            if len(args) != 9:
                raise PbrtSceneError("LookAt needs 9 numbers")
            st.ctm = st.ctm * host.Transform.look_at(args[0:3], args[3:6], args[6:9])
        elif name == "Translate":
            st.ctm = st.ctm * host.Transform.translate(args[0:3])
        elif name == "Scale":
            m = np.diag(np.asarray(list(args[0:3]) + [1.0], np.float32)).astype(np.float32)
            st.ctm = st.ctm * host.Transform.from_matrix(m)
        elif name == "Rotate":
            st.ctm = st.ctm * _rotate(args[0], args[1:4])
        elif name == "Identity":
            st.ctm = host.Transform.IDENTITY
        elif name in ("Transform", "ConcatTransform"):
            m = np.asarray(args, np.float32).reshape(4, 4).T        # pbrt files are column-major
            t = host.Transform.from_matrix(np.ascontiguousarray(m))
            st.ctm = t if name == "Transform" else st.ctm * t
        elif name == "ReverseOrientation":
            st.reverse = not st.reverse
        elif name == "AttributeBegin":
            s2 = _State()
            s2.ctm, s2.material, s2.reverse = st.ctm, st.material, st.reverse
            stack.append(s2)
        elif name == "AttributeEnd":
            if not stack:
                raise PbrtSceneError("AttributeEnd without AttributeBegin")
            st = stack.pop()
        elif name == "Camera":
            if args[0] != "perspective":
                raise PbrtSceneError(f'Camera "{args[0]}" is not supported (the reference has only the perspective camera)')
            camera = {"camera_from_world": st.ctm, "fov": _float(p, "fov", 90.0), "lens": _float(p, "lensradius", 0.0),
                      "focal": _float(p, "focaldistance", 1e6)}
        elif name == "Film":
            if args and args[0] != "rgb":
                raise PbrtSceneError(f'Film "{args[0]}" is not supported (the reference has RGBFilm only, camera/film.rs:24-28)')
            film["res"] = (int(_float(p, "xresolution", 1280)), int(_float(p, "yresolution", 720)))
            if "filename" in p:
                settings["filename"] = p["filename"][1][0].strip('"')
                if not settings["filename"].lower().endswith(".exr"):
                    warn(f'Film filename "{settings["filename"]}": the reference writes OpenEXR only (imaging/image.rs:48-57); '
                         f"use write_exr with an .exr name")
        elif name == "Sampler":
            settings["spp"] = int(_float(p, "pixelsamples", 16))
            settings["seed"] = int(_float(p, "seed", 0))
            if args and args[0] != "independent":
                warn(f'Sampler "{args[0]}" -> "independent": the reference has IndependentSampler only (sampling/sampler.rs:10-14); '
                     f'{settings["spp"]} samples per pixel kept')
        elif name == "PixelFilter":
            if args[0] != "gaussian":
                raise PbrtSceneError(f'PixelFilter "{args[0]}" is not supported by the GPU film sampler')
            r = _float(p, "radius", 1.5)
            film["filter"] = ("gaussian", (_float(p, "xradius", r), _float(p, "yradius", r)), _float(p, "sigma", 0.5))
        elif name == "Integrator":
            kind = args[0]
            if kind not in ("simplepath", "path", "volpath", "randomwalk"):
                raise PbrtSceneError(f'Integrator "{kind}" is not supported')
            settings["integrator"] = {"randomwalk": "randomwalk", "volpath": "volpath"}.get(kind, "simplepath")
            if kind == "path":
                warn('Integrator "path" -> "simplepath": the reference has no MIS / Russian-roulette path integrator '
                     "(integrators/base.rs:26-30)")
            if kind == "volpath":
                warn('Integrator "volpath" -> VolPath extension (SimplePath + homogeneous medium); this file declares no medium '
                     "(MakeNamedMedium / MediumInterface are not ingested), so it renders as simplepath")
            settings["max_depth"] = int(_float(p, "maxdepth", 5))
            settings["sample_lights"] = _bool(p, "samplelights", True)
            settings["sample_bsdf"] = _bool(p, "samplebsdf", True)
        elif name == "WorldBegin":
            if camera is None:
                raise PbrtSceneError("WorldBegin before Camera")
            in_world = True
            st.ctm = host.Transform.IDENTITY
        elif name == "Texture":
            tname, ttype, tclass = args[0], args[1], args[2]
            if tclass == "constant":
                named_textures[tname] = spectrum_param(p, "value") or ("const", np.float32(1.0))
            elif tclass == "checkerboard" and ttype == "spectrum":
                if int(_float(p, "dimension", 2)) != 2:
                    raise PbrtSceneError("only the 2-D checkerboard is supported")
                named_textures[tname] = ("checker", (spectrum_param(p, "tex1") or ("const", np.float32(1.0)),
                                                     spectrum_param(p, "tex2") or ("const", np.float32(0.0)),
                                                     _float(p, "uscale", 1.0), _float(p, "vscale", 1.0),
                                                     _float(p, "udelta", 0.0), _float(p, "vdelta", 0.0)))
            else:
                raise PbrtSceneError(f'Texture class "{tclass}" ({ttype}) is not supported (the reference has only constant '
                                     f"textures, materials/textures.rs:63-84; the spectrum checkerboard is this library's extension)")
        elif name == "Material":
            st.material = (args[0], p)
        elif name == "LightSource":
            pending_lights.append((args[0], p, st.ctm))
        elif name == "Shape":
            pending_shapes.append((args[0], p, st.ctm, st.material, st.reverse))
        elif name in ("ColorSpace", "Option", "MakeNamedMedium", "MediumInterface"):
            raise PbrtSceneError(f"directive {name} is not supported")
        else:
            raise PbrtSceneError(f"unknown or unsupported directive {name}")
    if not in_world:
        raise PbrtSceneError("no WorldBegin")

    res = film["res"]
    _, radius, sigma = film["filter"]
    sd.film = host.make_film(sp, res, filter_kind=host.PM_FILTER_GAUSSIAN, filter_radius=(float(radius[0]), float(radius[1])),
                             sigma=float(sigma))
    sd.camera, rfw = host.make_camera(camera["camera_from_world"].inverse(), float(camera["fov"]), res,
                                      lens_radius=float(camera["lens"]), focal_distance=float(camera["focal"]))
    sd.render_from_world = rfw
    f32 = np.float32

    def light_table(spec, default_illum=True):
        if spec is None:
            return sp.d65.copy()
        kind, v = spec
        if kind == "rgb":
            return sp.rgb_illuminant_table(v)[0]
        if kind == "blackbody":
            return sp.blackbody_table(float(v))
        if kind == "table":
            return v.copy()
        return sp.constant_table(float(v))

    for kind, p, ctm in pending_lights:
        scale = _float(p, "scale", 1.0)
        if kind == "infinite":
            table = light_table(spectrum_param(p, "L"))
            # RgbIlluminantSpectrum::to_photometric is its illuminant's (sampling/spectrum.rs:1016-1018)
            y = sp.photometric(sp.d65) if ("L" in p and p["L"][0] == "rgb") or "L" not in p else sp.photometric(table)
            sd.lights.append(host.light(host.PM_LIGHT_UNIFORM_INFINITE, table, f32(scale) / f32(y), (rfw * ctm).m))
        elif kind == "distant":
            frm = np.asarray(_floats(p, "from", [0, 0, 0], 3), f32)
            to = np.asarray(_floats(p, "to", [0, 0, 1], 3), f32)
            w = frm - to
            w = w / np.sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2], dtype=f32)
            if abs(w[0]) > abs(w[1]):
                v1 = np.array([-w[2], 0.0, w[0]], f32) / np.sqrt(w[0] * w[0] + w[2] * w[2], dtype=f32)
            else:
                v1 = np.array([0.0, w[2], -w[1]], f32) / np.sqrt(w[1] * w[1] + w[2] * w[2], dtype=f32)
            v2 = np.cross(w.astype(np.float64), v1.astype(np.float64)).astype(f32)
            m = np.array([[v1[0], v2[0], w[0], 0], [v1[1], v2[1], w[1], 0], [v1[2], v2[2], w[2], 0], [0, 0, 0, 1]], f32)
            table = light_table(spectrum_param(p, "L"))
            y = sp.photometric(sp.d65) if "L" in p and p["L"][0] == "rgb" else sp.photometric(table)
            sd.lights.append(host.light(host.PM_LIGHT_DIRECTIONAL, table, f32(scale) / f32(y),
                                        (rfw * ctm * host.Transform.from_matrix(m)).m))
        elif kind == "point":
            frm = _floats(p, "from", [0, 0, 0], 3)
            table = light_table(spectrum_param(p, "I"))
            y = sp.photometric(sp.d65) if "I" in p and p["I"][0] == "rgb" else sp.photometric(table)
            sd.lights.append(host.light(host.PM_LIGHT_POINT, table, f32(scale) / f32(y),
                                        (rfw * ctm * host.Transform.translate(frm)).m))
        else:
            raise PbrtSceneError(f'LightSource "{kind}" is not supported (the reference has point, distant and uniform infinite lights)')

    material_ids = {}

    def material_id(mat):
        kind, p = mat
        key = (kind, tuple(sorted((k, v[0], tuple(v[1])) for k, v in p.items())))
        if key in material_ids:
            return material_ids[key]
        if kind == "diffuse":
            spec = spectrum_param(p, "reflectance") or ("const", f32(0.5))

            def albedo(s):
                if s[0] == "rgb":
                    return sp.rgb_albedo(s[1])
                if s[0] == "const":
                    return host.constant_spectrum(float(s[1]))
                if s[0] == "table":
                    return host.dense_spectrum(s[1])
                raise PbrtSceneError("diffuse reflectance must be rgb, a constant, a sampled spectrum, or a checkerboard of those")
            if spec[0] == "checker":
                t1, t2, su, sv, du, dv = spec[1]
                m = host.checkerboard_diffuse(albedo(t1), albedo(t2), float(su), float(sv), float(du), float(dv))
            else:
                m = host.diffuse(albedo(spec))
        elif kind == "dielectric":
            eta = spectrum_param(p, "eta") or ("const", f32(1.5))
            if eta[0] not in ("const", "table"):
                raise PbrtSceneError("dielectric eta must be a constant or a sampled spectrum")
            r = _float(p, "roughness", 0.0)
            eta_spec = host.constant_spectrum(float(eta[1])) if eta[0] == "const" else host.dense_spectrum(eta[1])
            m = host.dielectric(eta_spec, float(_float(p, "uroughness", r)),
                                float(_float(p, "vroughness", r)), _bool(p, "remaproughness", True))
        else:
            raise PbrtSceneError(f'Material "{kind}" is not supported (the reference has diffuse and dielectric)')
        sd.materials.append(m)
        material_ids[key] = len(sd.materials) - 1
        return material_ids[key]

    prim_mats = []
    for kind, p, ctm, mat, reverse in pending_shapes:
        mid = material_id(mat)
        t = rfw * ctm
        if kind == "sphere":
            r = float(_float(p, "radius", 1.0))
            sd.prims.append(Sphere(r, float(_float(p, "zmin", -r)), float(_float(p, "zmax", r)), float(_float(p, "phimax", 360.0)),
                                   t.m, t.m_inv, reverse))
            prim_mats.append(mid)
        elif kind in ("bilinearmesh", "trianglemesh"):
            from .scenes import _xf_points
            P = np.asarray(_floats(p, "P"), f32).reshape(-1, 3)
            k = 4 if kind == "bilinearmesh" else 3
            if "indices" in p:
                idx = np.asarray([int(float(x)) for x in p["indices"][1]], np.uint32).reshape(-1, k)
            elif len(P) == k:
                idx = np.arange(k, dtype=np.uint32).reshape(1, k)
            else:
                raise PbrtSceneError(f'Shape "{kind}" needs "integer indices"')
            pos = _xf_points(t.m, P)
            if kind == "bilinearmesh":
                nrm = None
                if "N" in p:
                    n = np.asarray(_floats(p, "N"), f32).reshape(-1, 3)
                    mi = t.m_inv          # Transform * Normal3f uses the inverse transpose (core/transform.rs:300-318)
                    nrm = np.stack([n[:, 0] * mi[0, c] + n[:, 1] * mi[1, c] + n[:, 2] * mi[2, c] for c in range(3)], axis=1).astype(f32)
                    if reverse:
                        nrm = -nrm
                uv = None
                if "uv" in p or "st" in p:
                    uv = np.asarray(_floats(p, "uv" if "uv" in p else "st"), f32).reshape(-1, 2)
                    if len(uv) != len(P):
                        raise PbrtSceneError('"point2 uv" needs one pair per vertex')
                det = np.linalg.det(t.m[:3, :3].astype(np.float64))
                sd.prims.append(BilinearPatchMesh(pos, idx, reverse, bool(det < 0), normals=nrm, uv=uv))
            else:
                sd.prims.append(TriangleMesh(pos, idx))
            prim_mats.extend([mid] * len(idx))
        else:
            raise PbrtSceneError(f'Shape "{kind}" is not supported')
    sd.prim_materials = prim_mats
    return sd, settings


def load_pbrt_file(path, spectra=None):
    with open(path) as f:
        return load_pbrt(f.read(), spectra)


def integrator_from(sd, settings, device_id=0):
    """WavefrontIntegrator for a parsed scene"""
    from ._capi import PM_INTEGRATOR_RANDOM_WALK, PM_INTEGRATOR_SIMPLE_PATH, PM_INTEGRATOR_VOL_PATH
    from .integrator import WavefrontIntegrator
    kind = {"randomwalk": PM_INTEGRATOR_RANDOM_WALK, "volpath": PM_INTEGRATOR_VOL_PATH}.get(settings["integrator"], PM_INTEGRATOR_SIMPLE_PATH)
    return WavefrontIntegrator(settings["max_depth"], settings["sample_lights"], settings["sample_bsdf"], sd, settings["spp"],
                               seed=settings["seed"], device_id=device_id, integrator=kind)
