"""Host-side scene construction over the `pmh_*` helpers (include/pbrt_mi355_host.h).

Mirrors the constructor calls of the reference's `render_cpu()` (src/main.rs:43-234): Transform::look_at,
PerspectiveCamera::builder, GaussianFilter::new, PixelSensor::with_xyz_matching, RGBFilm::new,
RgbIlluminantSpectrum / BlackbodySpectrum / RgbAlbedoSpectrum, UniformInfiniteLight / DirectionalLight /
PointLight, DiffuseMaterial / DielectricMaterial — and flattens the result into the C-ABI structs.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import (PM_FILTER_BOX, PM_FILTER_GAUSSIAN, PM_LIGHT_DIFFUSE_AREA, PM_LIGHT_DIRECTIONAL, PM_LIGHT_POINT,
                    PM_LIGHT_UNIFORM_INFINITE, PM_MAT_DIELECTRIC, PM_MAT_DIFFUSE, PM_SPEC_CONSTANT, PM_SPEC_DENSE,
                    PM_SPEC_RGB_ALBEDO, PmCamera, PmFilm, PmLight, PmMaterial, PmSpectrum)

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cie_tables.npz")


def _p(a):
    return a.ctypes.data


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Transform:
    """core/transform.rs Transform {m, m_inv}, row-major float32 4x4."""

    def __init__(self, m, m_inv):
        self.m = _f32(m).reshape(4, 4)
        self.m_inv = _f32(m_inv).reshape(4, 4)

    IDENTITY = None

    @staticmethod
    def translate(delta):
        m = np.eye(4, dtype=np.float32)
        mi = np.eye(4, dtype=np.float32)
        m[:3, 3] = _f32(delta)
        mi[:3, 3] = -_f32(delta)
        return Transform(m, mi)

    @staticmethod
    def from_matrix(m):
        m = _f32(m).reshape(4, 4)
        out = np.zeros((4, 4), np.float32)
        if _capi.lib().pmh_mat4_inverse(_p(m), _p(out)) != 0:
            raise ValueError("singular matrix")
        return Transform(m, out)

    @staticmethod
    def look_at(pos, look, up):
        m = np.zeros((4, 4), np.float32)
        mi = np.zeros((4, 4), np.float32)
        a, b, c = _f32(pos), _f32(look), _f32(up)
        if _capi.lib().pmh_look_at(_p(a), _p(b), _p(c), _p(m), _p(mi)) != 0:
            raise ValueError("degenerate look_at")
        return Transform(m, mi)

    def inverse(self):
        return Transform(self.m_inv, self.m)

    def __mul__(self, other):
        m = np.zeros((4, 4), np.float32)
        mi = np.zeros((4, 4), np.float32)
        L = _capi.lib()
        L.pmh_mat4_mul(_p(self.m), _p(other.m), _p(m))
        L.pmh_mat4_mul(_p(other.m_inv), _p(self.m_inv), _p(mi))
        return Transform(m, mi)


Transform.IDENTITY = Transform(np.eye(4), np.eye(4))


class Spectra:
    """CIE tables, the sRGB colour space illuminant and the spectrum constructors of sampling/spectrum.rs."""

    def __init__(self, rgb_mode="table"):
        if rgb_mode not in ("table", "direct"):
            raise ValueError(rgb_mode)
        self.rgb_mode = rgb_mode
        self._table = None
        d = np.load(_DATA)
        self.cie_x, self.cie_y, self.cie_z = (_f32(d[k]) for k in ("cie_x", "cie_y", "cie_z"))
        d65i = _f32(d["d65_interleaved"])
        self.d65 = np.zeros(471, np.float32)
        L = _capi.lib()
        L.pmh_d65_table(_p(self.cie_y), _p(d65i), len(d65i), _p(self.d65))
        self.xyz_from_rgb = np.zeros(9, np.float32)
        self.rgb_from_xyz = np.zeros(9, np.float32)
        if L.pmh_srgb_matrices(_p(self.cie_x), _p(self.cie_y), _p(self.cie_z), _p(self.d65), _p(self.xyz_from_rgb),
                               _p(self.rgb_from_xyz)) != 0:
            raise RuntimeError("sRGB matrices")

    def photometric(self, table):
        return float(_capi.lib().pmh_photometric(_p(self.cie_y), _p(_f32(table))))

    # ---- RGBColorSpace::to_rgb_coeffs (color/color_space.rs:110-184) -------------------------------------------
    # The reference interpolates a 3 x 64^3 coefficient table read from src/util/rgbspectrum_srgb.txt, a file missing
    # from the snapshot.  The table is regenerated here (pmh_rgb_table_generate, ~10 s once, cached next to the CIE
    # tables) and looked up exactly as RGBToSpectrumTable::convert does; scripts/gen_rgbspectrum_table.py writes the same
    # table in the text format the reference loads.  rgb_mode="direct" solves each colour without the table instead.
    TABLE_PATH = os.path.join(os.path.dirname(_DATA), "rgbspectrum_srgb.bin")
    _TABLE_MAGIC = b"PMRGBT2\0"

    def rgb_table(self):
        if self._table is None:
            n = 3 * 64 ** 3 * 3
            path = self.TABLE_PATH
            if os.path.exists(path) and os.path.getsize(path) == 8 + 4 * (64 + n) and open(path, "rb").read(8) == self._TABLE_MAGIC:
                raw = np.fromfile(path, np.float32, offset=8)
                z, co = raw[:64].copy(), raw[64:].copy()
            else:
                z, co = np.zeros(64, np.float32), np.zeros(n, np.float32)
                threads = min(16, len(os.sched_getaffinity(0)))
                # matrices = None: the 6-digit sRGB matrices the published table was generated with
                if _capi.lib().pmh_rgb_table_generate(_p(self.cie_x), _p(self.cie_y), _p(self.cie_z), _p(self.d65),
                                                      None, _p(z), _p(co), threads) != 0:
                    raise RuntimeError("rgb table generation failed")
                try:
                    tmp = f"{path}.{os.getpid()}.tmp"
                    with open(tmp, "wb") as f:
                        f.write(self._TABLE_MAGIC)
                        f.write(z.tobytes())
                        f.write(co.tobytes())
                    os.replace(tmp, path)
                except OSError:
                    pass          # read-only tree: keep the table in memory only
            self._table = (z, co)
        return self._table

    def rgb_coeffs(self, rgb):
        out = np.zeros(3, np.float32)
        rgb = _f32(np.maximum(rgb, 0))
        if self.rgb_mode == "table":
            z, co = self.rgb_table()
            _capi.lib().pmh_rgb_table_lookup(_p(z), _p(co), _p(rgb), _p(out))
            return out
        rc = _capi.lib().pmh_rgb_to_coeffs(_p(rgb), _p(self.cie_x), _p(self.cie_y), _p(self.cie_z), _p(self.d65),
                                           _p(self.xyz_from_rgb), _p(out))
        if rc != 0:
            raise RuntimeError("rgb_to_coeffs failed")
        return out

    def rgb_albedo(self, rgb):
        """RgbAlbedoSpectrum::new(&SRGB, rgb) -> ('rgb_albedo', c0, c1, c2)"""
        c = self.rgb_coeffs(rgb)
        return {"kind": PM_SPEC_RGB_ALBEDO, "a": float(c[0]), "b": float(c[1]), "c": float(c[2])}

    def rgb_illuminant_table(self, rgb):
        """RgbIlluminantSpectrum::new(&SRGB, rgb) densified as lights do (lights/base.rs:177-188)."""
        rgb = _f32(rgb)
        m = float(max(rgb))
        scale = np.float32(2.0) * np.float32(m)
        c = self.rgb_coeffs(rgb / scale if scale != 0 else np.zeros(3, np.float32))
        out = np.zeros(471, np.float32)
        _capi.lib().pmh_rgb_spectrum_table(_p(c), C.c_float(scale), _p(self.d65), _p(out))
        return out, c, float(scale)

    def blackbody_table(self, temp):
        out = np.zeros(471, np.float32)
        _capi.lib().pmh_blackbody_table(C.c_float(temp), _p(out))
        return out

    def constant_table(self, c):
        return np.full(471, c, np.float32)

    def piecewise_linear_table(self, interleaved, normalize=False):
        """PiecewiseLinearSpectrum::from_interleaved([lambda0, v0, lambda1, v1, ...], normalize) densified at 360..830 nm
        (sampling/spectrum.rs:269-310); ValueError for what upstream asserts on"""
        il = _f32(interleaved).reshape(-1)
        out = np.zeros(471, np.float32)
        if _capi.lib().pmh_piecewise_linear_table(_p(il), len(il), int(bool(normalize)), _p(self.cie_y), _p(out)) != 0:
            raise ValueError("piecewise-linear spectrum: an even, non-zero number of values with strictly ascending wavelengths is required")
        return out

    def sensor_xyz_tables(self):
        """PixelSensor::with_xyz_matching: r_bar, g_bar, b_bar = X, Y, Z (camera/sensor.rs:85-111)"""
        return np.ascontiguousarray(np.concatenate([self.cie_x, self.cie_y, self.cie_z]).astype(np.float32))


def constant_spectrum(c):
    return {"kind": PM_SPEC_CONSTANT, "a": float(c)}


def dense_spectrum(table):
    return {"kind": PM_SPEC_DENSE, "table": _f32(table)}


def diffuse(reflectance):
    """DiffuseMaterial::new(ConstantSpectrumTexture(reflectance))"""
    return {"type": PM_MAT_DIFFUSE, "spectrum": reflectance, "u_roughness": 0.0, "v_roughness": 0.0, "remap": False}


def checkerboard_diffuse(tex1, tex2, su=1.0, sv=1.0, du=0.0, dv=0.0):
    """DiffuseMaterial whose reflectance is a 2-D checkerboard texture (extension, scenes/test.pbrt:26-30): tex1 where
    floor(su u + du) + floor(sv v + dv) is even, tex2 where it is odd"""
    m = diffuse(tex1)
    m.update(texture=_capi.PM_TEX_CHECKERBOARD, spectrum2=tex2, su=float(su), sv=float(sv), du=float(du), dv=float(dv))
    return m


def dielectric(eta, u_roughness=0.0, v_roughness=0.0, remap_roughness=False):
    """DielectricMaterial::new(u_rough, v_rough, remap, eta)"""
    return {"type": PM_MAT_DIELECTRIC, "spectrum": eta, "u_roughness": float(u_roughness),
            "v_roughness": float(v_roughness), "remap": bool(remap_roughness)}


def light(kind, table, scale, render_from_light=None):
    m = np.eye(4, dtype=np.float32) if render_from_light is None else _f32(render_from_light).reshape(4, 4)
    return {"type": kind, "scale": float(np.float32(scale)), "m": m, "table": _f32(table)}


def area_light(table, scale, prim_id, two_sided=False):
    """DiffuseAreaLight (extension) on the patch / triangle `prim_id`: L = scale * spectrum on the normal's side"""
    l = light(PM_LIGHT_DIFFUSE_AREA, table, scale)
    l.update(prim_id=int(prim_id), two_sided=bool(two_sided))
    return l


def homogeneous_medium(sigma_a, sigma_s, sigma_scale=1.0, g=0.0, bounds=None):
    """HomogeneousMedium::new(sigma_a, sigma_s, sigma_scale, emission = 0, g) (media/medium.rs:49-68): 471-entry tables of the
    two coefficients, scaled; bounds = (min xyz, max xyz) in render space, None = everywhere"""
    b = (-np.inf,) * 3 + (np.inf,) * 3 if bounds is None else tuple(float(x) for x in bounds)
    return {"sigma_a": _f32(sigma_a) * np.float32(sigma_scale), "sigma_s": _f32(sigma_s) * np.float32(sigma_scale),
            "g": float(g), "bounds": b}


class SceneDescription:
    """Everything `SimplePathIntegrator::new(...)` + `render()` consume, flattened."""

    def __init__(self):
        self.prims = []            # Sphere | BilinearPatchMesh | TriangleMesh, insertion order
        self.prim_materials = []   # one material id per primitive (meshes contribute one id per face)
        self.materials = []
        self.lights = []
        self.medium = None         # homogeneous_medium(...) or None
        self.camera = None         # PmCamera
        self.film = None           # dict, see make_film
        self.render_from_world = None
        self._keep = []

    # -- flattening into ctypes (kept alive by self._keep) ------------------------------------------
    def c_materials(self):
        arr = (PmMaterial * max(1, len(self.materials)))()
        for i, m in enumerate(self.materials):
            arr[i].type = m["type"]
            sp = m["spectrum"]
            arr[i].spectrum.kind = sp["kind"]
            for k in "abcd":
                setattr(arr[i].spectrum, k, float(sp.get(k, 0.0)))
            if "table" in sp:
                t = _f32(sp["table"])
                self._keep.append(t)
                arr[i].spectrum.table = t.ctypes.data_as(C.POINTER(C.c_float))
            arr[i].u_roughness, arr[i].v_roughness = m["u_roughness"], m["v_roughness"]
            arr[i].remap_roughness = int(m["remap"])
            arr[i].texture = int(m.get("texture", 0))
            if arr[i].texture:
                arr[i].su, arr[i].sv, arr[i].du, arr[i].dv = m["su"], m["sv"], m["du"], m["dv"]
                s2 = m["spectrum2"]
                arr[i].spectrum2.kind = s2["kind"]
                for k in "abcd":
                    setattr(arr[i].spectrum2, k, float(s2.get(k, 0.0)))
                if "table" in s2:
                    t2 = _f32(s2["table"])
                    self._keep.append(t2)
                    arr[i].spectrum2.table = t2.ctypes.data_as(C.POINTER(C.c_float))
        return arr

    def c_lights(self):
        arr = (PmLight * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            arr[i].type = l["type"]
            arr[i].scale = l["scale"]
            arr[i].render_from_light[:] = l["m"].reshape(-1).tolist()
            self._keep.append(l["table"])
            arr[i].spectrum = l["table"].ctypes.data_as(C.POINTER(C.c_float))
            arr[i].prim_id = l.get("prim_id", 0)
            arr[i].two_sided = int(l.get("two_sided", False))
        return arr

    def c_medium(self):
        if self.medium is None:
            return None
        m = _capi.PmMedium()
        self._keep += [self.medium["sigma_a"], self.medium["sigma_s"]]
        m.sigma_a = self.medium["sigma_a"].ctypes.data_as(C.POINTER(C.c_float))
        m.sigma_s = self.medium["sigma_s"].ctypes.data_as(C.POINTER(C.c_float))
        m.g = self.medium["g"]
        m.bounds[:] = list(self.medium["bounds"])
        return m

    def c_prim_materials(self):
        return np.ascontiguousarray(self.prim_materials, dtype=np.uint32)

    def c_film(self):
        f = self.film
        out = PmFilm()
        out.full_resolution[:] = list(f["full_resolution"])
        out.pixel_bounds[:] = list(f["pixel_bounds"])
        out.max_component_value = f["max_component_value"]
        out.imaging_ratio = f["imaging_ratio"]
        out.output_rgb_from_sensor_rgb[:] = _f32(f["output_rgb_from_sensor_rgb"]).reshape(-1).tolist()
        self._keep.append(f["sensor"])
        out.sensor_rgb_bar = f["sensor"].ctypes.data_as(C.POINTER(C.c_float))
        out.filter_kind = f["filter_kind"]
        out.filter_radius[:] = list(f["filter_radius"])
        if f["filter_kind"] == PM_FILTER_GAUSSIAN:
            out.filter_nx, out.filter_ny = f["filter_nx"], f["filter_ny"]
            self._keep.append(f["filter_vals"])
            out.filter_vals = f["filter_vals"].ctypes.data_as(C.POINTER(C.c_float))
        return out


def make_camera(world_from_camera, fov, resolution, screen_window=(-1.0, -1.0, 1.0, 1.0), lens_radius=0.0,
                focal_distance=10e30):
    """PerspectiveCamera::builder()...build() (camera/perspective.rs:30-86). Returns (PmCamera, render_from_world)."""
    cam = PmCamera()
    wfr = np.zeros((4, 4), np.float32)
    rfw = np.zeros((4, 4), np.float32)
    sw = _f32(screen_window)
    rc = _capi.lib().pmh_perspective_camera(_p(world_from_camera.m), _p(world_from_camera.m_inv), C.c_float(fov), _p(sw),
                                            int(resolution[0]), int(resolution[1]), C.c_float(lens_radius),
                                            C.c_float(min(focal_distance, 3.0e38)), C.byref(cam), _p(wfr), _p(rfw))
    if rc != 0:
        raise RuntimeError("perspective camera")
    return cam, Transform(rfw, wfr)


def make_film(spectra, resolution, pixel_bounds=None, filter_kind=PM_FILTER_GAUSSIAN, filter_radius=(1.5, 1.5), sigma=0.5,
              imaging_ratio=1.0, max_component_value=float("inf")):
    """RGBFilm::new(RGBFilmParams{.., sensor: PixelSensor::with_xyz_matching(&SRGB, None, ratio), color_space: &SRGB})"""
    w, h = resolution
    film = {"full_resolution": (w, h), "pixel_bounds": tuple(pixel_bounds) if pixel_bounds else (0, 0, w, h),
            "max_component_value": float(max_component_value), "imaging_ratio": float(imaging_ratio),
            # output_rgb_from_sensor_rgb = rgb_from_xyz * xyz_from_sensor_rgb (identity for the XYZ sensor) (film.rs:128-129)
            "output_rgb_from_sensor_rgb": spectra.rgb_from_xyz.copy(), "sensor": spectra.sensor_xyz_tables(),
            "filter_kind": filter_kind, "filter_radius": (float(filter_radius[0]), float(filter_radius[1]))}
    if filter_kind == PM_FILTER_GAUSSIAN:
        nx, ny = C.c_int(), C.c_int()
        L = _capi.lib()
        L.pmh_gaussian_filter_table(C.c_float(filter_radius[0]), C.c_float(filter_radius[1]), C.c_float(sigma), None, 0,
                                    C.byref(nx), C.byref(ny))
        vals = np.zeros(nx.value * ny.value, np.float32)
        if L.pmh_gaussian_filter_table(C.c_float(filter_radius[0]), C.c_float(filter_radius[1]), C.c_float(sigma), _p(vals),
                                       vals.size, C.byref(nx), C.byref(ny)) != 0:
            raise RuntimeError("filter table")
        film.update(filter_nx=nx.value, filter_ny=ny.value, filter_vals=vals)
    return film
