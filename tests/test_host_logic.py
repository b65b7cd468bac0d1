"""Host-side logic of the product, checked on the CPU: the wide-BVH builder (through a host-only context, device -1),
the host constructors (pmh_*) against the oracle's restatement of the reference constructors, and the shard maths of
the multi-GPU split."""
import ctypes as C

import numpy as np
import pytest


def _agg(pm, prims, **kw):
    return pm.GpuAggregate(prims, device_id=-1, **kw)


def test_wide_tree_walk_equals_reference_traversal(pm, ora):
    """Walking the GPU-layout tree (80-byte 8-wide quantised nodes) on the CPU in kernel order with the oracle's
    primitive tests gives the reference traversal's hits bit for bit, with far fewer node visits."""
    pos, idx = pm.scenes.blob_mesh(16, 512, seed=2, blob_sigma=0.25, tri_size=0.05)
    agg = _agg(pm, [pm.TriangleMesh(pos, idx)])
    assert agg.info.n_prims == len(idx) and agg.info.node_bytes == 80
    rays = np.concatenate([pm.scenes.primary_rays(64, 64), pm.scenes.incoherent_rays(4096, seed=9)])
    wide, wc = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, flat=agg.info.flat_prims != 0)
    sc = ora.Scene()
    sc.add_triangle_mesh(pos, idx)
    sc.build(ora.SPLIT_SAH, 4)
    ref, rc = sc.intersect(rays)
    assert (ref["prim_id"] != 0xFFFFFFFF).sum() > 1000
    assert np.array_equal(wide["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert np.array_equal(wide["u"].view(np.uint32), ref["u"].view(np.uint32))
    assert np.array_equal(wide["v"].view(np.uint32), ref["v"].view(np.uint32))
    assert (wide["prim_id"] != ref["prim_id"]).sum() <= 1          # exact-t tie class only
    assert wc["nodes_visited"] < rc["nodes_visited"]
    occ, _ = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, any_hit=True, flat=agg.info.flat_prims != 0)
    assert np.array_equal(occ.astype(bool), ref["prim_id"] != 0xFFFFFFFF)


def test_wide_tree_invariants(pm):
    """every primitive is referenced by exactly one leaf; child counts and bases are consistent; de-quantised child
    boxes contain their primitives (outward rounding)."""
    pos, idx = pm.scenes.blob_mesh(8, 300, seed=6, blob_sigma=0.3, tri_size=0.06)
    agg = _agg(pm, [pm.TriangleMesh(pos, idx)])
    nb = agg.node_buffer()
    n_nodes = agg.info.n_nodes
    assert len(nb) == n_nodes * 80
    node_dt = np.dtype([("o", "<f4", 3), ("e", "u1", 3), ("imask", "u1"), ("child_base", "<u4"), ("prim_base", "<u4"),
                        ("meta", "u1", 8), ("qlo", "u1", (3, 8)), ("qhi", "u1", (3, 8))])
    assert node_dt.itemsize == 80
    nodes = nb.view(node_dt)
    prim = agg.prim_buffer().view(np.uint32).reshape(-1, 4)
    tri_units = 3
    seen = np.zeros(len(idx), np.int32)
    reached = np.zeros(n_nodes, np.int32)
    reached[0] = 1
    tri = pos[idx.reshape(-1)].reshape(-1, 3, 3)
    for i in range(n_nodes):
        nd = nodes[i]
        scale = np.array([np.uint32(int(e) << 23).view(np.float32) for e in nd["e"]], np.float32)
        n_int = bin(int(nd["imask"])).count("1")
        for c in range(n_int):
            reached[nd["child_base"] + c] += 1
        pbase = int(nd["prim_base"]) & 0x3FFFFFFF
        for s in range(8):
            meta = int(nd["meta"][s])
            if (nd["imask"] >> s) & 1:
                continue
            cnt, off = meta >> 5, meta & 31
            if cnt == 0:
                continue
            lo = nd["o"] + nd["qlo"][:, s].astype(np.float32) * scale
            hi = nd["o"] + nd["qhi"][:, s].astype(np.float32) * scale
            for j in range(cnt):
                rec = prim[pbase + (off + j) * tri_units: pbase + (off + j + 1) * tri_units].reshape(-1)
                pid = int(rec[9])
                seen[pid] += 1
                v = rec[:9].view(np.float32).reshape(3, 3)
                assert np.array_equal(v, tri[pid])
                assert (v >= lo - 0).all() and (v <= hi + 0).all(), (i, s, lo, hi, v)
    assert (seen == 1).all()
    assert (reached == 1).all()


def test_mixed_types_join_node(pm, ora):
    import helpers
    sd = pm.scenes.main_rs_scene((16, 16))
    agg = _agg(pm, sd.prims)
    rays = pm.scenes.incoherent_rays(4096, seed=1, radius=8.0)
    wide, _ = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, flat=agg.info.flat_prims != 0)
    sc = helpers.oracle_scene_from(sd, ora)
    ref, _ = sc.intersect(rays)
    assert np.array_equal(wide["prim_id"], ref["prim_id"])
    assert np.array_equal(wide["t"].view(np.uint32), ref["t"].view(np.uint32))
    lo, hi = agg.bounds()
    olo, ohi = sc.bounds()
    assert np.array_equal(lo, olo) and np.array_equal(hi, ohi)


def test_empty_and_errors(pm):
    with pytest.raises(pm.PmError):      # BVHAggregate::new panics on an empty list (aggregates.rs:60-64 reduce().unwrap())
        _agg(pm, [])
    with pytest.raises(pm.PmError):
        _agg(pm, [pm.TriangleMesh(np.zeros((3, 3), np.float32), [[0, 1, 7]])])      # index out of range
    with pytest.raises(TypeError):
        _agg(pm, [object()])


def test_host_constructors_match_oracle(pm, ora):
    L, O = pm._capi.lib(), ora.lib()
    rng = np.random.default_rng(0)
    for _ in range(50):
        m = rng.normal(size=(4, 4)).astype(np.float32)
        a, b = np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32)
        assert L.pmh_mat4_inverse(m.ctypes.data, a.ctypes.data) == 0
        assert O.ora_mat4_inverse(m.ctypes.data, b.ctypes.data) == 1
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    L.pmh_gaussian.restype = C.c_float
    L.pmh_gaussian.argtypes = [C.c_float] * 3
    L.pmh_fast_exp.restype = C.c_float
    L.pmh_fast_exp.argtypes = [C.c_float]
    for x in np.linspace(-30, 30, 301, dtype=np.float32):
        assert np.float32(L.pmh_fast_exp(float(x))).view(np.uint32) == np.float32(O.ora_fast_exp(float(x))).view(np.uint32)
        assert np.float32(L.pmh_gaussian(float(x), 0.5, 2.0)).view(np.uint32) == np.float32(O.ora_gaussian(float(x), 0.5, 2.0)).view(np.uint32)


def test_look_at_and_camera(pm):
    """Transform::look_at(...).inverse() places the camera at `pos` (transform.rs:150-188); CameraTransform makes
    render space = world translated by -pos (camera/base.rs:59-94)."""
    w2c = pm.host.Transform.look_at((3.0, 4.0, 1.5), (0.5, 0.5, 0.0), (0.0, 0.0, 1.0))
    c2w = w2c.inverse()
    assert np.allclose(c2w.m[:3, 3], [3.0, 4.0, 1.5])
    assert np.allclose(w2c.m @ c2w.m, np.eye(4), atol=1e-6)
    cam, rfw = pm.host.make_camera(c2w, 45.0, (40, 30))
    assert np.allclose(rfw.m[:3, 3], [-3.0, -4.0, -1.5])
    assert np.allclose(rfw.m[:3, :3], np.eye(3))


def test_gaussian_filter_table(pm):
    """GaussianFilter tabulation for FilterSampler (imaging/filter.rs:183-214): 32 samples per unit radius, values
    max(0, gaussian(x) - gaussian(r)) * same in y, evaluated at Bounds2f::lerp'd points (with the reference's x-for-y quirk)."""
    sp = pm.host.Spectra()
    film = pm.host.make_film(sp, (8, 8))
    assert (film["filter_nx"], film["filter_ny"]) == (48, 48)
    v = film["filter_vals"].reshape(48, 48)
    assert (v >= 0).all() and v.max() > 0
    assert np.allclose(v, v.T, rtol=1e-6)


def test_wave_schedule(pm):
    """the sample-index waves of ImageTileIntegrate::image_tile_render (integrators/base.rs:85-127), restated here literally"""
    def reference(spp):
        out, wave_start, wave_end, next_wave_size = [], 0, 1, 1
        while wave_start < spp:
            out.append((wave_start, wave_end))
            wave_start = wave_end
            wave_end = min(spp, wave_end + next_wave_size)
            next_wave_size = min(2 * next_wave_size, 64)
        return out
    ws = pm.WavefrontIntegrator.wave_schedule
    assert ws(256) == [(0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 192), (192, 256)]
    for spp in (1, 2, 3, 7, 64, 65, 200, 1000):
        w = ws(spp)
        assert w == reference(spp)
        assert w[0][0] == 0 and w[-1][1] == spp and all(a[1] == b[0] for a, b in zip(w, w[1:]))
        assert max(b - a for a, b in w) <= 64


def test_shard_maths(pm):
    d = pm.distributed
    for spp in (1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            r = [d.shard_samples(spp, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == spp
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    for bounds in ((0, 0, 40, 24), (3, 5, 50, 102), (0, 0, 8, 8)):
        for world in (1, 2, 3, 8):
            w = [d.shard_rows(bounds, k, world) for k in range(world)]
            assert w[0][1] == bounds[1] and w[-1][3] == bounds[3]
            assert all(w[k][3] == w[k + 1][1] for k in range(world - 1))
            assert all(a[0] == bounds[0] and a[2] == bounds[2] for a in w)


def _read_exr(path):
    """minimal reader for single-part scanline uncompressed FLOAT EXR files (what pmh_write_exr writes)"""
    import struct
    b = open(path, "rb").read()
    assert b[:4] == bytes([0x76, 0x2F, 0x31, 0x01]) and b[4] == 2 and b[5:8] == b"\x00\x00\x00"
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\x00", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\x00", pos); typ = b[pos:e].decode(); pos = e + 1
        (size,) = struct.unpack_from("<i", b, pos); pos += 4
        attrs[name] = (typ, b[pos:pos + size]); pos += size
    pos += 1
    assert attrs["compression"][1] == b"\x00" and attrs["lineOrder"][1] == b"\x00"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    ch, p = [], 0
    cl = attrs["channels"][1]
    while cl[p] != 0:
        e = cl.index(b"\x00", p); ch.append(cl[p:e].decode()); p = e + 1
        assert struct.unpack_from("<i", cl, p)[0] == 2      # FLOAT
        p += 16
    assert ch == ["B", "G", "R"]
    offs = struct.unpack_from(f"<{h}Q", b, pos)
    img = np.zeros((h, w, 3), np.float32)
    for y in range(h):
        yy, size = struct.unpack_from("<2i", b, offs[y])
        assert yy == y and size == 12 * w
        line = np.frombuffer(b, np.float32, 3 * w, offs[y] + 8).reshape(3, w)
        img[y, :, 2], img[y, :, 1], img[y, :, 0] = line[0], line[1], line[2]
    assert offs[-1] + 8 + 12 * w == len(b)
    return img


def test_exr_writer_round_trip(pm, tmp_path):
    """Image::write_exr counterpart (imaging/image.rs:48-57): what is written parses back bit for bit"""
    rng = np.random.default_rng(0)
    img = rng.normal(size=(7, 13, 3)).astype(np.float32)
    img[0, 0] = [np.inf, 0.0, -0.0]
    path = tmp_path / "t.exr"
    assert pm._capi.lib().pmh_write_exr(str(path).encode(), 13, 7, img.ctypes.data) == 0
    back = _read_exr(path)
    assert np.array_equal(back.view(np.uint32), img.view(np.uint32))
    assert pm._capi.lib().pmh_write_exr(str(tmp_path / "no" / "dir.exr").encode(), 13, 7, img.ctypes.data) != 0
