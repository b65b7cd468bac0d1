"""Pins for the parts of the oracle that the reference holds no test for ("parity unpinned by the reference"):
  * third-party arithmetic restated from published algorithms is checked against PUBLISHED known answers
    (PCG32 demo vector of the PCG paper's C library; MurmurHash3_x86_128 via an independent pure-Python port that
    reproduces the published digest of the pangram);
  * BVH build + traversal is checked against the brute-force loop over all primitives (every split method);
  * the any-hit walk is checked against the closest-hit walk.
No GPU needed."""
import ctypes as C
import struct

import numpy as np

M = 0xFFFFFFFF


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & M


def _fmix(h):
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M
    h ^= h >> 16
    return h


def murmur3_x86_128(data, seed=0):
    """Appleby's MurmurHash3_x86_128, written independently of oracle/ref_render.h (general length)."""
    c1, c2, c3, c4 = 0x239B961B, 0xAB0E9789, 0x38B34AE5, 0xA1E38B93
    h1 = h2 = h3 = h4 = seed
    n = len(data)
    nb = n // 16
    for i in range(nb):
        k1, k2, k3, k4 = struct.unpack_from("<4I", data, 16 * i)
        k1 = (k1 * c1) & M; k1 = _rotl(k1, 15); k1 = (k1 * c2) & M; h1 ^= k1
        h1 = _rotl(h1, 19); h1 = (h1 + h2) & M; h1 = (h1 * 5 + 0x561CCD1B) & M
        k2 = (k2 * c2) & M; k2 = _rotl(k2, 16); k2 = (k2 * c3) & M; h2 ^= k2
        h2 = _rotl(h2, 17); h2 = (h2 + h3) & M; h2 = (h2 * 5 + 0x0BCAA747) & M
        k3 = (k3 * c3) & M; k3 = _rotl(k3, 17); k3 = (k3 * c4) & M; h3 ^= k3
        h3 = _rotl(h3, 15); h3 = (h3 + h4) & M; h3 = (h3 * 5 + 0x96CD1C35) & M
        k4 = (k4 * c4) & M; k4 = _rotl(k4, 18); k4 = (k4 * c1) & M; h4 ^= k4
        h4 = _rotl(h4, 13); h4 = (h4 + h1) & M; h4 = (h4 * 5 + 0x32AC3B17) & M
    tail = data[nb * 16:]
    k = [0, 0, 0, 0]
    for i, b in enumerate(tail):
        k[i // 4] ^= b << (8 * (i % 4))
    k1, k2, k3, k4 = k
    t = len(tail)
    if t > 12:
        k4 = (k4 * c4) & M; k4 = _rotl(k4, 18); k4 = (k4 * c1) & M; h4 ^= k4
    if t > 8:
        k3 = (k3 * c3) & M; k3 = _rotl(k3, 17); k3 = (k3 * c4) & M; h3 ^= k3
    if t > 4:
        k2 = (k2 * c2) & M; k2 = _rotl(k2, 16); k2 = (k2 * c3) & M; h2 ^= k2
    if t > 0:
        k1 = (k1 * c1) & M; k1 = _rotl(k1, 15); k1 = (k1 * c2) & M; h1 ^= k1
    h1 ^= n; h2 ^= n; h3 ^= n; h4 ^= n
    h1 = (h1 + h2) & M; h1 = (h1 + h3) & M; h1 = (h1 + h4) & M
    h2 = (h2 + h1) & M; h3 = (h3 + h1) & M; h4 = (h4 + h1) & M
    h1, h2, h3, h4 = _fmix(h1), _fmix(h2), _fmix(h3), _fmix(h4)
    h1 = (h1 + h2) & M; h1 = (h1 + h3) & M; h1 = (h1 + h4) & M
    h2 = (h2 + h1) & M; h3 = (h3 + h1) & M; h4 = (h4 + h1) & M
    return h1, h2, h3, h4


def test_pcg32_published_vector(ora):
    """pcg32_srandom_r(42, 54) of the PCG C library's demo prints these first six outputs; rand_pcg's
    Lcg64Xsh32::new(state, stream) is documented as the same initialisation."""
    st = (C.c_uint64 * 2)()
    ora.lib().ora_pcg32_new(42, 54, st)
    got = [ora.lib().ora_pcg32_next(st) for _ in range(6)]
    assert got == [0xA15C02B7, 0x7B47F409, 0xBA1D3330, 0x83D2F293, 0xBFA4784B, 0xCBED606E]


def test_murmur3_pixel_hash(ora):
    # the independent port reproduces the published x86_128 digest of the pangram ...
    h = murmur3_x86_128(b"The quick brown fox jumps over the lazy dog")
    assert "".join(f"{x:08x}" for x in h) == "2f1583c3ecee2c675d7bf66ce5e91d2c"
    # ... and the oracle's 8-byte specialisation equals it (low 64 bits of the u128 = h2<<32 | h1, rng.rs:47-50)
    rng = np.random.default_rng(0)
    for _ in range(200):
        x, y = (int(v) for v in rng.integers(-2**31, 2**31, 2))
        seed = int(rng.integers(0, 2**32))
        h = murmur3_x86_128(struct.pack("<ii", x, y), seed)
        assert ora.lib().ora_hash_pixel(x, y, seed) == ((h[1] << 32) | h[0])


def test_sampler_stream_properties(ora):
    """IndependentSampler: every (pixel, sample, dimension) restarts PCG32 from (hash(pixel, seed), sample*65536+dim);
    floats are (u32 >> 9) * 2^-23 in [0, 1 - 2^-23] (util/rng.rs:19,34-36; sampling/sampler.rs:104-137)."""
    out = np.zeros(64, np.float32)
    ora.lib().ora_sampler_stream(7, 9, 0, 3, 0, 64, out.ctypes.data)
    assert (out >= 0).all() and (out <= np.float32(1 - 2.0**-23)).all()
    assert np.array_equal(out * np.float32(2**23), np.round(out * np.float32(2**23)))     # multiples of 2^-23
    st = (C.c_uint64 * 2)()
    ora.lib().ora_pcg32_new(ora.lib().ora_hash_pixel(7, 9, 0), 3 * 65536, st)
    ref = [np.float32((ora.lib().ora_pcg32_next(st) >> 9) * 2.0**-23) for _ in range(64)]
    assert np.array_equal(out, np.asarray(ref, np.float32))
    out2 = np.zeros(64, np.float32)
    ora.lib().ora_sampler_stream(7, 9, 0, 4, 0, 64, out2.ctypes.data)
    assert not np.array_equal(out, out2)


def _soup(pm, n_blobs=4, per=256, seed=11):
    return pm.scenes.blob_mesh(n_blobs, per, seed=seed, blob_sigma=0.35, tri_size=0.08)


def test_bvh_equals_brute_force_all_split_methods(ora, pm):
    pos, idx = _soup(pm)
    rays = np.concatenate([pm.scenes.primary_rays(48, 48), pm.scenes.incoherent_rays(2048, seed=5)])
    ref = None
    for method in (ora.SPLIT_MIDDLE, ora.SPLIT_SAH, ora.SPLIT_EQUAL_COUNTS):
        for max_prims in (1, 4, 255):
            sc = ora.Scene()
            sc.add_triangle_mesh(pos, idx)
            sc.build(method, max_prims)
            if ref is None:
                ref, _ = sc.intersect(rays, brute_force=True)
                assert (ref["prim_id"] != 0xFFFFFFFF).sum() > 500
            hits, ctr = sc.intersect(rays)
            same_t = hits["t"].view(np.uint32) == ref["t"].view(np.uint32)
            assert same_t.all(), (method, max_prims)
            # ids may differ only where two primitives are hit at exactly the same t (tie class)
            diff = hits["prim_id"] != ref["prim_id"]
            assert diff.sum() <= 2, (method, max_prims, int(diff.sum()))
            occ, _ = sc.intersect_p(rays)
            assert np.array_equal(occ.astype(bool), ref["prim_id"] != 0xFFFFFFFF)


def test_triangle_is_the_degenerate_patch(ora, pm):
    """TriangleMesh is an extension: each triangle is fed to the reference's bilinear-patch test as (v0,v1,v2,v2)."""
    pos, idx = _soup(pm, 2, 128, seed=4)
    quads = np.concatenate([idx, idx[:, 2:3]], axis=1)
    rays = pm.scenes.primary_rays(64, 64)
    a, b = ora.Scene(), ora.Scene()
    a.add_triangle_mesh(pos, idx)
    b.add_patch_mesh(pos, quads)
    a.build()
    b.build()
    ha, _ = a.intersect(rays)
    hb, _ = b.intersect(rays)
    assert np.array_equal(ha.view(np.uint32), hb.view(np.uint32))


def test_mixed_scene_bounds_and_hits(ora, pm):
    """main.rs scene (3 spheres + floor patch): world bounds = union of primitive bounds; BVH == brute force."""
    import helpers
    sd = pm.scenes.main_rs_scene((32, 32))
    sc = helpers.oracle_scene_from(sd, ora)
    lo, hi = sc.bounds()
    assert (lo < hi).all()
    rays = pm.scenes.incoherent_rays(4096, seed=3, radius=8.0)
    h1, _ = sc.intersect(rays)
    h2, _ = sc.intersect(rays, brute_force=True)
    assert np.array_equal(h1["t"].view(np.uint32), h2["t"].view(np.uint32))
    assert np.array_equal(h1["prim_id"], h2["prim_id"])
    assert len(set(h1["prim_id"].tolist()) - {0xFFFFFFFF}) == 4
