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
