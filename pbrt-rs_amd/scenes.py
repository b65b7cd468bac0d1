"""Synthetic geometry and ray generators for the BASELINE.json configurations.

All data is synthetic and seeded (numpy PCG64, `np.random.default_rng(seed)`); nothing is loaded
from disk.  Rays are emitted in 8x8 pixel-tile order so that one wave (64 lanes) covers one tile,
the same order the wavefront renderer schedules camera rays in.
"""
import numpy as np

from ._capi import RAY_DTYPE


def blob_mesh(n_blobs=1024, tris_per_blob=1024, seed=1, blob_sigma=0.06, tri_size=0.012):
    """C3/C4 geometry: `n_blobs` clusters of `tris_per_blob` small triangles.

    Blob centres uniform in [-1,1]^3; triangle centres normal(0, blob_sigma) around the blob centre;
    the three vertices uniform in a cube of half-width `tri_size` around the triangle centre.
    Returns (positions float32 [3T,3], indices uint32 [T,3]).
    """
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-1.0, 1.0, size=(n_blobs, 1, 3)).astype(np.float32)
    tri_c = centres + rng.normal(0.0, blob_sigma, size=(n_blobs, tris_per_blob, 3)).astype(np.float32)
    verts = tri_c[:, :, None, :] + rng.uniform(-tri_size, tri_size, size=(n_blobs, tris_per_blob, 3, 3)).astype(np.float32)
    pos = np.ascontiguousarray(verts.reshape(-1, 3), dtype=np.float32)
    idx = np.arange(pos.shape[0], dtype=np.uint32).reshape(-1, 3)
    return pos, idx


def blob_patch_mesh(n_blobs=1024, per_blob=1024, seed=1, blob_sigma=0.06, size=0.012, warp=0.35):
    """C3's geometry with the reference's own primitive: `n_blobs` clusters of `per_blob` small NON-PLANAR bilinear patches
    (blob centres uniform in [-1,1]^3, patch centres normal(0, blob_sigma) about them, two random edge vectors of at most `size`
    per component, and the fourth corner pushed off the parallelogram's plane by `warp` of the edge lengths).
    Returns (positions float32 [4P, 3], indices uint32 [P, 4] in p00 p10 p01 p11 order)."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-1.0, 1.0, size=(n_blobs, 1, 3)).astype(np.float32)
    c = centres + rng.normal(0.0, blob_sigma, size=(n_blobs, per_blob, 3)).astype(np.float32)
    e1 = rng.uniform(-size, size, size=(n_blobs, per_blob, 3)).astype(np.float32)
    e2 = rng.uniform(-size, size, size=(n_blobs, per_blob, 3)).astype(np.float32)
    w = (rng.uniform(-size, size, size=(n_blobs, per_blob, 3)) * warp).astype(np.float32)
    p00 = c - 0.5 * e1 - 0.5 * e2
    quad = np.stack([p00, p00 + e1, p00 + e2, p00 + e1 + e2 + w], axis=2)          # [B, P, 4, 3]
    pos = np.ascontiguousarray(quad.reshape(-1, 3), dtype=np.float32)
    idx = np.arange(pos.shape[0], dtype=np.uint32).reshape(-1, 4)
    return pos, idx


def terrain_patch_mesh(grid=1024, seed=7, tilt_deg=55.0, roughness=0.002):
    """A BilinearPatchMesh at scale - the reference's only mesh primitive (shapes/bilinear_patch.rs:615-707): a grid x grid
    heightfield of quads over [-1,1]^2 with SHARED vertices ((grid+1)^2 of them), rolling hills plus per-vertex noise so that
    every patch is non-planar, tilted by `tilt_deg` about the x axis so that rays from (0,0,-4) meet it at a slant.
    Returns (positions float32 [(grid+1)^2, 3], indices uint32 [grid^2, 4] in p00 p10 p01 p11 order)."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-1.0, 1.0, grid + 1, dtype=np.float64)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    Z = 0.12 * np.sin(3.1 * X + 0.4) * np.cos(2.6 * Y - 0.3) + 0.04 * np.sin(11.0 * X) * np.sin(13.0 * Y)
    Z += rng.normal(0.0, roughness, size=Z.shape)
    a = np.radians(tilt_deg)
    y2 = Y * np.cos(a) - Z * np.sin(a)
    z2 = Y * np.sin(a) + Z * np.cos(a)
    pos = np.ascontiguousarray(np.stack([X.reshape(-1), y2.reshape(-1), z2.reshape(-1)], axis=1), dtype=np.float32)
    j, i = np.meshgrid(np.arange(grid), np.arange(grid), indexing="ij")
    a00 = (j * (grid + 1) + i).reshape(-1)
    idx = np.ascontiguousarray(np.stack([a00, a00 + 1, a00 + grid + 1, a00 + grid + 2], axis=1), dtype=np.uint32)
    return pos, idx


def _tile_order(width, height, tile=8):
    """pixel (x, y) arrays in row-major order of tile x tile tiles, row-major inside a tile."""
    assert width % tile == 0 and height % tile == 0
    ty, tx, iy, ix = np.meshgrid(np.arange(height // tile), np.arange(width // tile), np.arange(tile), np.arange(tile),
                                 indexing="ij")
    x = (tx * tile + ix).reshape(-1)
    y = (ty * tile + iy).reshape(-1)
    return x.astype(np.int32), y.astype(np.int32)


def primary_rays(width=4096, height=4096, eye=(0.0, 0.0, -4.0), fov_deg=40.0):
    """Pinhole camera at `eye` looking down +z; one ray through every pixel centre, un-jittered."""
    x, y = _tile_order(width, height)
    t = np.float32(np.tan(np.radians(fov_deg) / 2.0))
    aspect = np.float32(width / height)
    px = ((x.astype(np.float32) + np.float32(0.5)) / np.float32(width) * 2 - 1) * t * aspect
    py = (1 - (y.astype(np.float32) + np.float32(0.5)) / np.float32(height) * 2) * t
    d = np.stack([px, py, np.ones_like(px)], axis=1).astype(np.float32)
    d /= np.sqrt((d * d).sum(axis=1, keepdims=True)).astype(np.float32)
    rays = np.zeros(x.shape[0], dtype=RAY_DTYPE)
    rays["o"] = np.asarray(eye, dtype=np.float32)
    rays["d"] = d
    rays["tmax"] = np.inf
    return rays


def shadow_rays(primary, hits, light=(0.0, 3.0, 0.0)):
    """Integrate::unoccluded-style rays (integrators/base.rs:39-44): from each primary hit point towards a
    point light, un-normalised direction, t_max = 1 - 1e-4.  Rays whose primary missed get t_max = 0."""
    hit = hits["prim_id"] != 0xFFFFFFFF
    t = np.where(hit, hits["t"], np.float32(0)).astype(np.float32)
    p = primary["o"] + primary["d"] * (t * np.float32(1 - 1e-3))[:, None]   # pulled slightly off the surface
    out = np.zeros(len(primary), dtype=RAY_DTYPE)
    out["o"] = p.astype(np.float32)
    out["d"] = (np.asarray(light, dtype=np.float32) - p).astype(np.float32)
    out["tmax"] = np.where(hit, np.float32(1.0 - 1e-4), np.float32(0)).astype(np.float32)
    return out


def incoherent_rays(n, seed=2, radius=1.9):
    """Origins uniform on a sphere of `radius`, directions cosine-distributed about the inward normal."""
    rng = np.random.default_rng(seed)
    u = rng.random((n, 4), dtype=np.float32)
    z = 1 - 2 * u[:, 0]
    r = np.sqrt(np.maximum(0, 1 - z * z))
    phi = 2 * np.pi * u[:, 1]
    nrm = np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1).astype(np.float32)
    o = nrm * np.float32(radius)
    # cosine sample about -nrm
    rr = np.sqrt(u[:, 2])
    ph = 2 * np.pi * u[:, 3]
    lx, ly, lz = rr * np.cos(ph), rr * np.sin(ph), np.sqrt(np.maximum(0, 1 - u[:, 2]))
    w = -nrm
    a = np.where(np.abs(w[:, :1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    t1 = np.cross(w, a)
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(w, t1)
    d = (t1 * lx[:, None] + t2 * ly[:, None] + w * lz[:, None]).astype(np.float32)
    rays = np.zeros(n, dtype=RAY_DTYPE)
    rays["o"] = o
    rays["d"] = d
    rays["tmax"] = np.inf
    return rays


# ------------------------------------------------------------------------------------------------
# full scenes for the path tracer (C1: src/main.rs:43-234 verbatim; C2: Cornell box as triangles)
# ------------------------------------------------------------------------------------------------
def _xf_points(m, pts):
    """Transform * Point3f for an affine matrix, f32, in the reference's operation order (transform.rs:320-335)."""
    m = np.asarray(m, np.float32)
    p = np.asarray(pts, np.float32)
    x = p[:, 0] * m[0, 0] + p[:, 1] * m[0, 1] + p[:, 2] * m[0, 2] + m[0, 3]
    y = p[:, 0] * m[1, 0] + p[:, 1] * m[1, 1] + p[:, 2] * m[1, 2] + m[1, 3]
    z = p[:, 0] * m[2, 0] + p[:, 1] * m[2, 1] + p[:, 2] * m[2, 2] + m[2, 3]
    return np.stack([x, y, z], axis=1).astype(np.float32)


def main_rs_scene(resolution=(400, 400), spectra=None):
    """The hardcoded scene of the reference binary (src/main.rs:43-234): 3 spheres + floor quad, sky + sun."""
    from . import host
    from .aggregate import BilinearPatchMesh, Sphere
    f32 = np.float32
    sp = spectra or host.Spectra()
    sd = host.SceneDescription()
    camera_from_world = host.Transform.look_at((3.0, 4.0, 1.5), (0.5, 0.5, 0.0), (0.0, 0.0, 1.0))
    sd.film = host.make_film(sp, resolution, filter_kind=host.PM_FILTER_GAUSSIAN, filter_radius=(1.5, 1.5), sigma=0.5)
    sd.camera, rfw = host.make_camera(camera_from_world.inverse(), 45.0, resolution, lens_radius=0.0, focal_distance=10e30)
    sd.render_from_world = rfw
    # lights (main.rs:89-112)
    sky_table, _, _ = sp.rgb_illuminant_table((0.4, 0.45, 0.5))
    sky_scale = f32(1.0) / f32(sp.photometric(sp.d65))          # RgbIlluminantSpectrum::to_photometric = illuminant's
    sd.lights.append(host.light(host.PM_LIGHT_UNIFORM_INFINITE, sky_table, sky_scale))
    w = np.array([-30.0, 40.0, 100.0], f32) - np.array([0.0, 0.0, 1.0], f32)
    w = w / np.sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2], dtype=f32)
    if abs(w[0]) > abs(w[1]):
        v2 = np.array([-w[2], 0.0, w[0]], f32) / np.sqrt(w[0] * w[0] + w[2] * w[2], dtype=f32)
    else:
        v2 = np.array([0.0, w[2], -w[1]], f32) / np.sqrt(w[1] * w[1] + w[2] * w[2], dtype=f32)
    v3 = np.cross(w.astype(np.float64), v2.astype(np.float64)).astype(f32)
    # (w, v1, v2) = w.coordinate_system()  ->  v1 := v2 above, v2 := v3
    sun_m = np.array([[v2[0], v3[0], w[0], 0], [v2[1], v3[1], w[1], 0], [v2[2], v3[2], w[2], 0], [0, 0, 0, 1]], f32)
    sun_xf = rfw * host.Transform.from_matrix(sun_m)
    sun_table = sp.blackbody_table(3000.0)
    sun_scale = f32(1.5) / f32(sp.photometric(sun_table))
    sd.lights.append(host.light(host.PM_LIGHT_DIRECTIONAL, sun_table, sun_scale, sun_xf.m))
    # materials
    sd.materials = [host.dielectric(host.constant_spectrum(1.5), 0.0, 0.0, False),
                    host.diffuse(sp.rgb_albedo((0.8, 0.8, 0.8))),
                    host.diffuse(sp.rgb_albedo((0.8, 0.45, 0.15))),
                    host.diffuse(sp.rgb_albedo((0.8, 0.25, 0.3)))]

    def sphere(radius, centre):
        t = rfw * host.Transform.translate(centre)
        return Sphere(radius, -radius, radius, 360.0, t.m, t.m_inv, False)

    floor_xf = rfw * host.Transform.translate((0.0, 0.0, -1.0))
    floor_pos = _xf_points(floor_xf.m, [[-20, -20, 0], [20, -20, 0], [-20, 20, 0], [20, 20, 0]])
    sd.prims = [sphere(1.0, (0.0, 0.0, 0.0)), BilinearPatchMesh(floor_pos, [[0, 1, 2, 3]]),
                sphere(1.5, (-3.5, -2.0, 0.5)), sphere(0.75, (-0.3, -3.0, -0.25))]
    sd.prim_materials = [0, 1, 2, 3]          # main.rs:217-219 order: sphere, floor, sphere2, sphere3
    return sd


def shading_normal_scene(resolution=(64, 64), grid=6, spectra=None):
    """main.rs scene whose floor is a grid x grid BilinearPatchMesh WITH vertex normals (a rippled shading normal over a
    flat quad mesh): exercises BilinearPatch::interaction_from_intersection's shading-geometry branch."""
    from . import host
    from .aggregate import BilinearPatchMesh
    sd = main_rs_scene(resolution, spectra)
    xs = np.linspace(-6.0, 6.0, grid + 1, dtype=np.float32)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    world = np.stack([X.reshape(-1), Y.reshape(-1), np.full(X.size, -1.0, np.float32)], axis=1)
    floor_xf = sd.render_from_world
    pos = _xf_points(floor_xf.m, world)
    nrm = np.stack([0.35 * np.sin(1.7 * world[:, 0]), 0.35 * np.cos(1.3 * world[:, 1]), np.ones(len(world))], axis=1)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)     # render_from_world is a translation
    idx = []
    for j in range(grid):
        for i in range(grid):
            a = j * (grid + 1) + i
            idx.append([a, a + 1, a + grid + 1, a + grid + 2])       # p00 p10 p01 p11
    sd.prims[1] = BilinearPatchMesh(pos, idx, normals=nrm)
    mats = sd.prim_materials
    sd.prim_materials = [mats[0]] + [mats[1]] * (grid * grid) + mats[2:]
    return sd


def blob_sky_scene(n_blobs=20480, resolution=(3840, 2160), seed=3, spectra=None, **blob_kw):
    """C4: `n_blobs` x 1024 small diffuse triangles under a uniform sky (UniformInfiniteLight), camera at (0,0,-4)
    looking down +z.  The default is the 20,971,520-triangle configuration of BASELINE.json."""
    from . import host
    from .aggregate import TriangleMesh
    f32 = np.float32
    sp = spectra or host.Spectra()
    sd = host.SceneDescription()
    camera_from_world = host.Transform.look_at((0.0, 0.0, -4.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    sd.film = host.make_film(sp, resolution, filter_kind=host.PM_FILTER_GAUSSIAN, filter_radius=(1.5, 1.5), sigma=0.5)
    sd.camera, rfw = host.make_camera(camera_from_world.inverse(), 40.0, resolution)
    sd.render_from_world = rfw
    pos, idx = blob_mesh(n_blobs, 1024, seed=seed, **blob_kw)
    pos = _xf_points(rfw.m, pos)
    sd.prims = [TriangleMesh(pos, idx)]
    sd.materials = [host.diffuse(sp.rgb_albedo((0.7, 0.6, 0.5)))]
    sd.prim_materials = np.zeros(len(idx), np.uint32)
    sky, _, _ = sp.rgb_illuminant_table((0.5, 0.6, 0.8))
    sd.lights = [host.light(host.PM_LIGHT_UNIFORM_INFINITE, sky, f32(1.0) / f32(sp.photometric(sp.d65)))]
    return sd


def cornell_scene(resolution=(1024, 1024), spectra=None, glass_sphere=True, area_light=False, medium=None):
    """C2: a Cornell box built from triangles (walls, tall box, short box), lit by a PointLight just under the
    ceiling (the reference has no area light, lights/base.rs:18-22), seen through a PerspectiveCamera.
    Optionally one smooth dielectric sphere (divergent specular paths).
    area_light: instead of the point light, a 0.5 x 0.5 quad emitter (DiffuseAreaLight extension) just under the ceiling,
    facing down - True: one bilinear patch; "triangles": two triangles of the scene's triangle mesh, each its own light
    ("Cornell box as triangles ... 1 ceiling quad light").  medium: (sigma_a, sigma_s, g) of a grey homogeneous medium filling the box
    (C5, BASELINE.json configs[4]; render it with PM_INTEGRATOR_VOL_PATH)."""
    from . import host
    from .aggregate import Sphere, TriangleMesh
    f32 = np.float32
    sp = spectra or host.Spectra()
    sd = host.SceneDescription()
    camera_from_world = host.Transform.look_at((0.0, 0.0, 3.9), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    sd.film = host.make_film(sp, resolution, filter_kind=host.PM_FILTER_GAUSSIAN, filter_radius=(1.5, 1.5), sigma=0.5)
    sd.camera, rfw = host.make_camera(camera_from_world.inverse(), 38.0, resolution)
    sd.render_from_world = rfw
    white, red, green, glass = 0, 1, 2, 3
    sd.materials = [host.diffuse(sp.rgb_albedo((0.73, 0.73, 0.73))), host.diffuse(sp.rgb_albedo((0.63, 0.065, 0.05))),
                    host.diffuse(sp.rgb_albedo((0.14, 0.45, 0.091))), host.dielectric(host.constant_spectrum(1.5))]
    verts, tris, mats = [], [], []

    def quad(a, b, c, d, m):
        i = len(verts)
        verts.extend([a, b, c, d])
        tris.extend([[i, i + 1, i + 2], [i, i + 2, i + 3]])
        mats.extend([m, m])

    def box(lo, hi, angle_deg, m):
        cx, cz = (lo[0] + hi[0]) / 2, (lo[2] + hi[2]) / 2
        ca, sa = np.cos(np.radians(angle_deg)), np.sin(np.radians(angle_deg))

        def P(x, y, z):
            dx, dz = x - cx, z - cz
            return [cx + ca * dx + sa * dz, y, cz - sa * dx + ca * dz]
        x0, y0, z0 = lo
        x1, y1, z1 = hi
        quad(P(x0, y0, z1), P(x1, y0, z1), P(x1, y1, z1), P(x0, y1, z1), m)
        quad(P(x1, y0, z0), P(x0, y0, z0), P(x0, y1, z0), P(x1, y1, z0), m)
        quad(P(x0, y0, z0), P(x0, y0, z1), P(x0, y1, z1), P(x0, y1, z0), m)
        quad(P(x1, y0, z1), P(x1, y0, z0), P(x1, y1, z0), P(x1, y1, z1), m)
        quad(P(x0, y1, z1), P(x1, y1, z1), P(x1, y1, z0), P(x0, y1, z0), m)
        quad(P(x0, y0, z0), P(x1, y0, z0), P(x1, y0, z1), P(x0, y0, z1), m)

    quad([-1, -1, -1], [1, -1, -1], [1, -1, 1], [-1, -1, 1], white)     # floor
    quad([-1, 1, -1], [-1, 1, 1], [1, 1, 1], [1, 1, -1], white)        # ceiling
    quad([-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1], white)    # back
    quad([-1, -1, -1], [-1, -1, 1], [-1, 1, 1], [-1, 1, -1], red)      # left
    quad([1, -1, -1], [1, 1, -1], [1, 1, 1], [1, -1, 1], green)        # right
    box((-0.7, -1.0, -0.65), (-0.1, 0.2, -0.05), 17.0, white)           # tall box
    box((0.1, -1.0, 0.0), (0.7, -0.4, 0.6), -18.0, white)               # short box
    emitter_tris = []
    if area_light == "triangles":
        # both normals (v1 - v0) x (v2 - v0) point down (-y): the emitting side faces the room
        i = len(verts)
        verts.extend([[-0.25, 0.98, -0.25], [0.25, 0.98, -0.25], [-0.25, 0.98, 0.25], [0.25, 0.98, 0.25]])
        emitter_tris = [len(tris), len(tris) + 1]
        tris.extend([[i, i + 1, i + 2], [i + 3, i + 2, i + 1]])
        mats.extend([white, white])
    pos = _xf_points(rfw.m, np.asarray(verts, f32))
    sd.prims = [TriangleMesh(pos, np.asarray(tris, np.uint32))]
    sd.prim_materials = list(mats)
    if glass_sphere:
        t = rfw * host.Transform.translate((0.4, -0.1, 0.3))
        sd.prims.append(Sphere(0.3, -0.3, 0.3, 360.0, t.m, t.m_inv, False))
        sd.prim_materials.append(glass)
    table = sp.d65.copy()
    if area_light == "triangles":
        # same radiance as the quad version below: L = 4 I / A with A = 0.25 (the whole quad)
        scale = f32(4.0 * 4.0 / 0.25) / f32(sp.photometric(table))
        sd.lights = [host.area_light(table, scale, t, two_sided=False) for t in emitter_tris]
    elif area_light:
        from .aggregate import BilinearPatchMesh
        # p00 p10 p01 p11 ordered so that dpdu x dpdv points down (-y): the emitting side faces the room
        quad_pos = _xf_points(rfw.m, np.asarray([[-0.25, 0.98, -0.25], [0.25, 0.98, -0.25], [-0.25, 0.98, 0.25], [0.25, 0.98, 0.25]], f32))
        sd.prims.append(BilinearPatchMesh(quad_pos, [[0, 1, 2, 3]]))
        sd.prim_materials.append(white)
        emitter_id = len(sd.prim_materials) - 1
        # same flux as the 4 / Y point light: Phi = 4 pi I = pi A L  =>  L = 4 I / A
        sd.lights = [host.area_light(table, f32(4.0 * 4.0 / 0.25) / f32(sp.photometric(table)), emitter_id, two_sided=False)]
    else:
        lt = rfw * host.Transform.translate((0.0, 0.9, 0.0))
        sd.lights = [host.light(host.PM_LIGHT_POINT, table, f32(4.0) / f32(sp.photometric(table)), lt.m)]
    if medium is not None:
        sigma_a, sigma_s, g = medium
        lo = _xf_points(rfw.m, np.asarray([[-1.0, -1.0, -1.0]], f32))[0]
        hi = _xf_points(rfw.m, np.asarray([[1.0, 1.0, 1.0]], f32))[0]
        sd.medium = host.homogeneous_medium(sp.constant_table(sigma_a), sp.constant_table(sigma_s), 1.0, g,
                                            bounds=tuple(np.minimum(lo, hi)) + tuple(np.maximum(lo, hi)))
    return sd


def c2_scene(resolution=(1024, 1024), spectra=None):
    """BASELINE.json configs[1] as SURVEY 8(d) writes it: the Cornell box as 36 triangles - 5 walls, 2 boxes and the ceiling quad
    EMITTER (two triangles, one DiffuseAreaLight each: the extension the reference lacks) - all diffuse, no point light."""
    return cornell_scene(resolution, spectra, glass_sphere=False, area_light="triangles")


def c5_scene(resolution=(1024, 1024), spectra=None):
    """BASELINE.json configs[4] / SURVEY 8(d) C5: the C2 box + ceiling quad emitter + homogeneous medium filling the box
    (sigma_s = 1.0, sigma_a = 0.05 per unit length, grey; g = 0.6); rendered with PM_INTEGRATOR_VOL_PATH at max_depth 16."""
    return cornell_scene(resolution, spectra, glass_sphere=True, area_light=True, medium=(0.05, 1.0, 0.6))
