"""PM_SPLIT_HLBVH: the wide BVH built ON THE DEVICE (pm_gpu_build.hip: Morton codes, radix sort, radix tree, refit,
collapse).  The tree differs from the host-built one, the hits must not: closest-hit results are tree-independent."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check_tree(agg, pos, idx):
    """every triangle in exactly one leaf; every node reached once; child boxes contain their primitives"""
    nb = agg.node_buffer()
    n_nodes = agg.info.n_nodes
    node_dt = np.dtype([("o", "<f4", 3), ("e", "u1", 3), ("imask", "u1"), ("child_base", "<u4"), ("prim_base", "<u4"),
                        ("meta", "u1", 8), ("qlo", "u1", (3, 8)), ("qhi", "u1", (3, 8))])
    nodes = nb.view(node_dt)
    assert len(nodes) == n_nodes
    prim = agg.prim_buffer().view(np.uint32).reshape(-1, 4)
    tri = pos[idx.reshape(-1)].reshape(-1, 3, 3)
    seen = np.zeros(len(idx), np.int32)
    reached = np.zeros(n_nodes, np.int32)
    reached[0] = 1
    scale = (nodes["e"].astype(np.uint32) << 23).view(np.float32)
    for i in range(n_nodes):
        nd = nodes[i]
        n_int = bin(int(nd["imask"])).count("1")
        reached[nd["child_base"]: nd["child_base"] + n_int] += 1
        pbase = int(nd["prim_base"]) & 0x3FFFFFFF
        assert n_int == 0 or (int(nd["prim_base"]) >> 30) == 2
        for s in range(8):
            meta = int(nd["meta"][s])
            if (nd["imask"] >> s) & 1 or meta == 0:
                continue
            cnt, off = meta >> 5, meta & 31
            lo = nd["o"] + nd["qlo"][:, s].astype(np.float32) * scale[i]
            hi = nd["o"] + nd["qhi"][:, s].astype(np.float32) * scale[i]
            for j in range(cnt):
                rec = prim[pbase + (off + j) * 3: pbase + (off + j + 1) * 3].reshape(-1)
                pid = int(rec[9])
                seen[pid] += 1
                v = rec[:9].view(np.float32).reshape(3, 3)
                assert np.array_equal(v, tri[pid])
                assert (v >= lo).all() and (v <= hi).all(), (i, s)
    assert (seen == 1).all()
    assert (reached == 1).all()


def test_device_built_tree(pm, ora):
    pos, idx = pm.scenes.blob_mesh(16, 1024, seed=5, blob_sigma=0.2, tri_size=0.04)
    agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)], split_method=pm._capi.PM_SPLIT_HLBVH)
    sah = pm.GpuAggregate([pm.TriangleMesh(pos, idx)], split_method=pm._capi.PM_SPLIT_SAH)
    assert agg.info.n_prims == len(idx) and agg.info.build_seconds > 0
    lo, hi = agg.bounds()
    slo, shi = sah.bounds()
    assert np.array_equal(lo, slo) and np.array_equal(hi, shi)
    _check_tree(agg, pos, idx)
    rays = np.concatenate([pm.scenes.primary_rays(256, 256), pm.scenes.incoherent_rays(1 << 16, seed=3)])
    sc = ora.Scene()
    sc.add_triangle_mesh(pos, idx)
    sc.build(ora.SPLIT_MIDDLE, 255)
    ref, _ = sc.intersect(rays)
    assert (ref["prim_id"] != 0xFFFFFFFF).sum() > 10000
    agg.set_counting(True)
    agg.reset_counters()
    got = agg.intersect(rays)
    c = agg.counters()
    agg.set_counting(False)
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)), f
    assert (got["prim_id"] != ref["prim_id"]).sum() <= 1
    occ_ref, _ = sc.intersect_p(rays)
    assert np.array_equal(agg.intersect_p(rays), occ_ref)
    # the CPU walk of the very same (device-built) tree reproduces the kernel's counters
    _, cw = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), rays, flat=agg.info.flat_prims != 0)
    assert (c["nodes_visited"], c["prims_tested"], c["exact_tests"]) == (cw["nodes_visited"], cw["prims_tested"], cw["exact_tests"])
    # a linear BVH is coarser than binned SAH, but not absurdly so
    sah.set_counting(True)
    sah.reset_counters()
    sah.intersect(rays)
    assert c["nodes_visited"] < 3 * sah.counters()["nodes_visited"]


def test_fallbacks(pm):
    """small or mixed scenes take the host builder even when HLBVH is asked for"""
    pos, idx = pm.scenes.blob_mesh(1, 256, seed=1)
    small = pm.GpuAggregate([pm.TriangleMesh(pos, idx)], split_method=pm._capi.PM_SPLIT_HLBVH)
    assert small.info.n_prims == 256
    sd = pm.scenes.main_rs_scene((16, 16))
    mixed = pm.GpuAggregate(sd.prims, split_method=pm._capi.PM_SPLIT_HLBVH)
    assert mixed.info.n_prims == 4
    rays = pm.scenes.primary_rays(32, 32)
    assert len(small.intersect(rays)) == len(rays)
