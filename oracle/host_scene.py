"""ORACLE - TEST INFRASTRUCTURE ONLY.

The reference's scene construction on the ORACLE's own restatement of its constructors (oracle/ref_host.h), without any
code of the product: ctypes mirrors of the structs ref_integrator.h declares, thin wrappers of the ora_host_* entry points,
and `main_rs_scene`, which follows src/main.rs:43-234 statement by statement.  tests/test_host_oracle.py compares what this
module builds with what pbrt-rs_amd/host.py + scenes.py build, bit for bit, and renders it against the GPU - so that the
product's host layer is no longer common to both sides of the parity tests.

Inputs that are DATA, not code: the CIE tables and the D65 samples (numbers of src/util/data.rs, kept in
pbrt-rs_amd/data/cie_tables.npz) and the sRGB coefficient table (the file missing upstream; the product's regenerated table
is passed in - it is pinned by the reference's vector in tests/test_oracle_golden.py).
"""
import ctypes as C

import numpy as np

import oracle as _o       # oracle/oracle.py (tests put this directory on sys.path)

_f32 = _o._f32


class OSpectrum(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_float), ("b", C.c_float), ("c", C.c_float), ("d", C.c_float),
                ("table", C.POINTER(C.c_float))]


class OMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("spec", OSpectrum), ("u_rough", C.c_float), ("v_rough", C.c_float),
                ("remap_roughness", C.c_int32), ("texture", C.c_int32), ("su", C.c_float), ("sv", C.c_float),
                ("du", C.c_float), ("dv", C.c_float), ("spec2", OSpectrum)]


class OLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("scale", C.c_float), ("render_from_light", C.c_float * 16),
                ("spectrum", C.POINTER(C.c_float)), ("prim_id", C.c_uint32), ("two_sided", C.c_int32)]


class OCamera(C.Structure):
    _fields_ = [("camera_from_raster", C.c_float * 16), ("render_from_camera", C.c_float * 16),
                ("lens_radius", C.c_float), ("focal_distance", C.c_float)]


class OFilm(C.Structure):
    _fields_ = [("full_resolution", C.c_int32 * 2), ("pixel_bounds", C.c_int32 * 4), ("max_component_value", C.c_float),
                ("imaging_ratio", C.c_float), ("output_rgb_from_sensor_rgb", C.c_float * 9),
                ("sensor_rgb_bar", C.POINTER(C.c_float)), ("filter_kind", C.c_int32), ("filter_radius", C.c_float * 2),
                ("filter_nx", C.c_int32), ("filter_ny", C.c_int32), ("filter_vals", C.POINTER(C.c_float))]


SPEC_CONSTANT, SPEC_DENSE, SPEC_RGB_ALBEDO = 0, 1, 2       # ref_integrator.h
MAT_DIFFUSE, MAT_DIELECTRIC = 0, 1
LIGHT_POINT, LIGHT_DIRECTIONAL, LIGHT_UNIFORM_INFINITE = 0, 1, 2
FILTER_GAUSSIAN = 2


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _lib():
    L = _o.lib()
    if not getattr(L, "_host_sigs", False):
        f, i, p = C.c_float, C.c_int, C.c_void_p
        for name, (res, args) in {
            "ora_host_look_at": (i, [p] * 5),
            "ora_host_perspective_camera": (i, [p, p, f, p, i, i, f, f, p, p, p]),
            "ora_host_compose": (None, [p] * 6), "ora_host_translate": (None, [p] * 3),
            "ora_host_frame_towards": (i, [p] * 4),
            "ora_host_gaussian_filter_table": (i, [f, f, f, p, i, p, p]),
            "ora_host_d65_table": (None, [p, p, i, p]), "ora_host_photometric": (f, [p, p]),
            "ora_host_blackbody_table": (f, [f, p]), "ora_host_rgb_spectrum_table": (None, [p, f, p, p]),
            "ora_host_srgb_matrices": (i, [p] * 6), "ora_host_xf_points": (None, [p, p, i, p]),
            "ora_rgb_table_convert": (None, [p] * 4),
        }.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L._host_sigs = True
    return L


class Xf:
    """Transform { m, m_inv }"""

    def __init__(self, m, inv):
        self.m, self.inv = _f32(m).reshape(4, 4), _f32(inv).reshape(4, 4)

    def inverse(self):
        return Xf(self.inv, self.m)

    def __mul__(self, other):
        m, inv = np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32)
        _lib().ora_host_compose(_p(self.m), _p(self.inv), _p(other.m), _p(other.inv), _p(m), _p(inv))
        return Xf(m, inv)

    def points(self, pts):
        pts = _f32(pts).reshape(-1, 3)
        out = np.zeros_like(pts)
        _lib().ora_host_xf_points(_p(self.m), _p(pts), len(pts), _p(out))
        return out


def look_at(pos, look, up):
    m, inv = np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32)
    if _lib().ora_host_look_at(_p(_f32(pos)), _p(_f32(look)), _p(_f32(up)), _p(m), _p(inv)) != 0:
        raise RuntimeError("look_at: singular")
    return Xf(m, inv)


def translate(d):
    m, inv = np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32)
    _lib().ora_host_translate(_p(_f32(d)), _p(m), _p(inv))
    return Xf(m, inv)


IDENTITY = Xf(np.eye(4), np.eye(4))


def frame_towards(src, dst):
    m, inv = np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32)
    if _lib().ora_host_frame_towards(_p(_f32(src)), _p(_f32(dst)), _p(m), _p(inv)) != 0:
        raise RuntimeError("frame: singular")
    return Xf(m, inv)


def perspective_camera(world_from_camera, fov, screen_window, resolution, lens_radius, focal_distance):
    cam = OCamera()
    wfr, rfw = np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32)
    sw = _f32(screen_window)
    if _lib().ora_host_perspective_camera(_p(world_from_camera.m), _p(world_from_camera.inv), fov, _p(sw), resolution[0],
                                          resolution[1], lens_radius, focal_distance, C.addressof(cam), _p(wfr), _p(rfw)) != 0:
        raise RuntimeError("perspective camera")
    return cam, Xf(rfw, wfr)


def gaussian_filter_table(rx, ry, sigma):
    nx, ny = C.c_int(), C.c_int()
    L = _lib()
    L.ora_host_gaussian_filter_table(rx, ry, sigma, None, 0, C.byref(nx), C.byref(ny))
    vals = np.zeros(nx.value * ny.value, np.float32)
    if L.ora_host_gaussian_filter_table(rx, ry, sigma, _p(vals), vals.size, C.byref(nx), C.byref(ny)) != 0:
        raise RuntimeError("filter table")
    return vals, nx.value, ny.value


class ColorData:
    """X, Y, Z, ILLUMD65 and SRGB of the reference (sampling/spectrum.rs:28-65, color/color_space.rs:40-48)"""

    def __init__(self, cie_x, cie_y, cie_z, d65_interleaved, table_z, table_coeffs):
        L = _lib()
        self.cie_x, self.cie_y, self.cie_z = _f32(cie_x), _f32(cie_y), _f32(cie_z)
        il = _f32(d65_interleaved)
        self.d65 = np.zeros(471, np.float32)
        L.ora_host_d65_table(_p(self.cie_y), _p(il), len(il), _p(self.d65))
        self.xyz_from_rgb, self.rgb_from_xyz = np.zeros(9, np.float32), np.zeros(9, np.float32)
        if L.ora_host_srgb_matrices(_p(self.cie_x), _p(self.cie_y), _p(self.cie_z), _p(self.d65), _p(self.xyz_from_rgb),
                                    _p(self.rgb_from_xyz)) != 0:
            raise RuntimeError("sRGB matrices")
        self.table_z, self.table_coeffs = _f32(table_z), _f32(table_coeffs)

    def photometric(self, table):
        return np.float32(_lib().ora_host_photometric(_p(self.cie_y), _p(_f32(table))))

    def rgb_coeffs(self, rgb):
        out = np.zeros(3, np.float32)
        _lib().ora_rgb_table_convert(_p(self.table_z), _p(self.table_coeffs), _p(_f32(rgb)), _p(out))
        return out

    def rgb_illuminant_table(self, rgb):
        """RgbIlluminantSpectrum::new(&SRGB, rgb) (spectrum.rs:988-1005), densified"""
        rgb = _f32(rgb)
        scale = np.float32(2.0) * max(rgb[0], rgb[1], rgb[2])
        c = self.rgb_coeffs(rgb / scale if scale != 0 else np.zeros(3, np.float32))
        out = np.zeros(471, np.float32)
        _lib().ora_host_rgb_spectrum_table(_p(c), scale, _p(self.d65), _p(out))
        return out

    def blackbody_table(self, temp):
        out = np.zeros(471, np.float32)
        _lib().ora_host_blackbody_table(temp, _p(out))
        return out


class HostScene:
    """what SimplePathIntegrator::new + render() consume, as the oracle's structs; keeps every buffer alive"""

    def __init__(self):
        self.scene = _o.Scene()
        self.materials, self.prim_materials, self.lights = [], [], []
        self.camera, self.film = None, None
        self._keep = []

    def _ptr(self, a):
        a = _f32(a)
        self._keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_float))

    def c_materials(self):
        arr = (OMaterial * len(self.materials))()
        for i, m in enumerate(self.materials):
            arr[i] = m
        return arr

    def c_lights(self):
        arr = (OLight * len(self.lights))()
        for i, l in enumerate(self.lights):
            arr[i] = l
        return arr

    def renderer(self):
        mats, lights = self.c_materials(), self.c_lights()
        r = _o.Renderer(self.scene, mats, len(self.materials), self.prim_materials, lights, len(self.lights), self.camera, self.film)
        r._keep = (self, mats, lights)
        return r


def main_rs_scene(color, resolution=(400, 400)):
    """src/main.rs:43-234 (render_cpu), in its order"""
    hs = HostScene()
    camera_from_world = look_at((3.0, 4.0, 1.5), (0.5, 0.5, 0.0), (0.0, 0.0, 1.0))                       # :44-48
    # :52 GaussianFilter::new((1.5, 1.5), 0.5); :54 PixelSensor::with_xyz_matching(&SRGB, None, 1.0); :56-74 RGBFilm
    vals, nx, ny = gaussian_filter_table(1.5, 1.5, 0.5)
    film = OFilm()
    film.full_resolution[:] = list(resolution)
    film.pixel_bounds[:] = [0, 0, resolution[0], resolution[1]]
    film.max_component_value, film.imaging_ratio = float("inf"), 1.0
    # output_rgb_from_sensor_rgb = rgb_from_xyz * xyz_from_sensor_rgb, the latter the identity (sensor.rs:96-101; film.rs:128-129)
    film.output_rgb_from_sensor_rgb[:] = color.rgb_from_xyz.tolist()
    film.sensor_rgb_bar = hs._ptr(np.concatenate([color.cie_x, color.cie_y, color.cie_z]))
    film.filter_kind = FILTER_GAUSSIAN
    film.filter_radius[:] = [1.5, 1.5]
    film.filter_nx, film.filter_ny, film.filter_vals = nx, ny, hs._ptr(vals)
    hs.film = film
    # :76-88 PerspectiveCamera
    hs.camera, render_from_world = perspective_camera(camera_from_world.inverse(), 45.0, (-1.0, -1.0, 1.0, 1.0), resolution, 0.0, 10e30)
    hs.render_from_world = render_from_world
    # :90-92 sky: RgbIlluminantSpectrum::to_photometric is its illuminant's (spectrum.rs:1015-1017)
    sky = OLight()
    sky.type, sky.scale = LIGHT_UNIFORM_INFINITE, np.float32(1.0) / color.photometric(color.d65)
    sky.render_from_light[:] = np.eye(4, dtype=np.float32).reshape(-1).tolist()
    sky.spectrum = hs._ptr(color.rgb_illuminant_table((0.4, 0.45, 0.5)))
    # :94-114 sun
    sun_table = color.blackbody_table(3000.0)
    sun = OLight()
    sun.type, sun.scale = LIGHT_DIRECTIONAL, np.float32(1.5) / color.photometric(sun_table)
    sun.render_from_light[:] = (render_from_world * frame_towards((-30.0, 40.0, 100.0), (0.0, 0.0, 1.0))).m.reshape(-1).tolist()
    sun.spectrum = hs._ptr(sun_table)
    hs.lights = [sky, sun]

    def diffuse(rgb):                                                   # RgbAlbedoSpectrum::new(&SRGB, rgb) in a DiffuseMaterial
        c = color.rgb_coeffs(rgb)
        m = OMaterial()
        m.type = MAT_DIFFUSE
        m.spec.kind, m.spec.a, m.spec.b, m.spec.c = SPEC_RGB_ALBEDO, c[0], c[1], c[2]
        return m
    glass = OMaterial()                                                 # :132-135 DielectricMaterial(0, 0, false, Constant(1.5))
    glass.type = MAT_DIELECTRIC
    glass.spec.kind, glass.spec.a = SPEC_CONSTANT, 1.5

    def sphere(radius, centre_xf):                                      # :116-131, :138-154, :164-180
        t = render_from_world * centre_xf
        hs.scene.add_sphere(radius, -radius, radius, 360.0, t.m, t.inv, False)
    # aggregate order (:217-219): sphere, floor, sphere2, sphere3
    sphere(1.0, IDENTITY)
    floor_xf = render_from_world * translate((0.0, 0.0, -1.0))       # :189-208
    hs.scene.add_patch_mesh(floor_xf.points([[-20, -20, 0], [20, -20, 0], [-20, 20, 0], [20, 20, 0]]), [[0, 1, 2, 3]], False, False)
    sphere(1.5, translate((-3.5, -2.0, 0.5)))
    sphere(0.75, translate((-0.3, -3.0, -0.25)))
    hs.materials = [glass, diffuse((0.8, 0.8, 0.8)), diffuse((0.8, 0.45, 0.15)), diffuse((0.8, 0.25, 0.3))]
    hs.prim_materials = np.array([0, 1, 2, 3], np.uint32)
    hs.scene.build(_o.SPLIT_MIDDLE, 255)                                 # :216-220
    return hs
