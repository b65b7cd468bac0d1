"""Exact-t ties between triangles (found by the full-size C4 determinism probe in round 3: two triangles of the 21 M-triangle soup
that one camera ray hits at bit-equal t = 3.121102).  The tree walk resolves such a tie to the LOWER primitive id - with that
triangle's own (u, v) - whatever order it meets the two in, in the timed kernels (parked leaves, pooled exact tests: the order
of a ray's tests depends on its wave) and in the counting kernels alike; the CPU mirror follows the same rule."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

A = np.array([[-0.9208327531814575, 0.4419354498386383, 2.944183826446533], [-0.9082875847816467, 0.45111405849456787, 2.95322847366333],
              [-0.9126601219177246, 0.44195812940597534, 2.9527363777160645]], np.float32)
B = np.array([[-0.9237823486328125, 0.44476571679115295, 2.957437038421631], [-0.9089869856834412, 0.4474371075630188, 2.95039701461792],
              [-0.9224495887756348, 0.4516768753528595, 2.9435372352600098]], np.float32)
D = np.array([0xbe959604, 0x3e12b737, 0x3f721147], np.uint32).view(np.float32)


def _single(ora, pm, tri, ray):
    """the reference traversal's hit of one triangle alone"""
    sc = ora.Scene()
    sc.add_triangle_mesh(tri.reshape(-1, 3), np.array([[0, 1, 2]], np.uint32))
    sc.build(ora.SPLIT_MIDDLE, 255)
    return sc.intersect(ray)[0][0]


@pytest.mark.parametrize("order", ["AB", "BA"])
@pytest.mark.parametrize("n_fill", [200, 5000])
def test_tie_goes_to_the_lower_id_with_its_own_uv(pm, ora, order, n_fill):
    rng = np.random.default_rng(n_fill)
    fill = (rng.uniform(-1.0, 1.0, (n_fill, 1, 3)) * [1.5, 1.5, 0.3] + [0.0, 0.0, 6.0] + rng.uniform(-0.02, 0.02, (n_fill, 3, 3))).astype(np.float32)
    tris = np.concatenate([fill, np.stack([A, B] if order == "AB" else [B, A])])
    pos = tris.reshape(-1, 3)
    idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
    agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)])
    assert agg.info.flat_prims == 0
    r = np.zeros(1, pm.RAY_DTYPE)
    r["d"] = D
    r["tmax"] = np.inf
    cpu, _ = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), r)
    ha, hb = _single(ora, pm, A, r), _single(ora, pm, B, r)
    assert ha["t"] == hb["t"] and ha["t"].view(np.uint32) == 1078444067 and ha["u"] != hb["u"]          # a bit-equal tie of two triangles
    want = ha if order == "AB" else hb              # the first of the pair has the lower id
    assert cpu["prim_id"][0] == n_fill and (cpu["t"][0], cpu["u"][0], cpu["v"][0]) == (want["t"], want["u"], want["v"])
    # the ray alone, in a full wave of copies, and scattered among other rays (other waves, other refill timings)
    other = pm.scenes.primary_rays(64, 64, eye=(0.0, 0.0, 0.0), fov_deg=30.0)
    for counting in (False, True):
        agg.set_counting(counting)
        for batch in (r, np.repeat(r, 64), np.repeat(r, 1000)):
            g = agg.intersect(batch)
            assert np.array_equal(g.view(np.uint32), np.repeat(cpu, len(batch)).view(np.uint32)), (counting, len(batch), g[:2], cpu)
        for pos_in in (0, 17, 63, 64, 1999):
            b = other.copy()
            b[pos_in] = r[0]
            g = agg.intersect(b)
            assert np.array_equal(g[pos_in:pos_in + 1].view(np.uint32), cpu.view(np.uint32)), (counting, pos_in, g[pos_in], cpu)
