"""Sphere::sample_with_context on the device against the oracle's restatement, function for function (round 3): the device
evaluates it through the pm_debug_sphere_sample hook of the C ABI for random receivers outside, far outside (Taylor branch), on and
inside the sphere; every output word must be bit-equal to the oracle's (correctly-rounded transcendental mode, as everywhere)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sphere_sample_with_context_bit_equal(pm, ora):
    host = pm.host
    sd = pm.scenes.cornell_scene((16, 16), glass_sphere=False)
    rfw = sd.render_from_world
    specs = [((-0.3, 0.55, 0.35), 0.18, False), ((0.6, 0.7, -0.5), 0.02, True), ((0.0, 0.0, 0.0), 6.0, True)]
    ids = []
    for centre, radius, reverse in specs:
        t = rfw * host.Transform.translate(centre)
        sd.prims.append(pm.Sphere(radius, -radius, radius, 360.0, t.m, t.m_inv, reverse))
        sd.prim_materials.append(0)
        ids.append((len(sd.prim_materials) - 1, t, radius, reverse))
    integ = pm.WavefrontIntegrator(1, True, True, sd, 1)
    integ.render()
    L = ora.lib()
    L.ora_sphere_sample_ctx_batch.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
    rng = np.random.default_rng(11)
    n = 1 << 20
    with ora.correctly_rounded():
        for prim_id, t, radius, reverse in ids:
            c = np.asarray(t.m, np.float32).reshape(4, 4)[:3, 3]
            # receivers: a third near the sphere (inside and outside), a third in the box, a third far away
            scale = np.concatenate([np.full(n // 3, 1.5 * radius), np.full(n // 3, 2.0), np.full(n - 2 * (n // 3), 200.0 * radius)]).astype(np.float32)
            ctx_p = (c[None, :] + (rng.random((n, 3), dtype=np.float32) * 2 - 1) * scale[:, None]).astype(np.float32)
            # p_origin: the point itself (medium) or nudged towards the centre (what offset_ray_origin_towards does, in effect)
            nudge = np.where(rng.random(n) < 0.5, 0.0, 1e-5).astype(np.float32)
            p_origin = (ctx_p + (c[None, :] - ctx_p) * nudge[:, None]).astype(np.float32)
            in8 = np.concatenate([ctx_p, p_origin, rng.random((n, 2), dtype=np.float32)], axis=1).astype(np.float32)
            got = integ.aggregate.debug_sphere_sample(prim_id, in8)
            want = np.zeros((n, 8), np.float32)
            s5 = np.array([radius, -radius, radius, 360.0, 1.0 if reverse else 0.0], np.float32)
            m = np.ascontiguousarray(t.m, np.float32); mi = np.ascontiguousarray(t.m_inv, np.float32)
            L.ora_sphere_sample_ctx_batch(s5.ctypes.data, m.ctypes.data, mi.ctypes.data, in8.ctypes.data, n, want.ctypes.data)
            ok = want[:, 0] == 1
            assert np.array_equal(got[:, 0], want[:, 0])
            inside = ((p_origin - c[None, :]) ** 2).sum(axis=1) <= radius * radius
            bad = np.argwhere((got[ok].view(np.uint32) != want[ok].view(np.uint32)).any(axis=1)).ravel()
            print(f"[sphere-sample] radius {radius}: {ok.sum()} samples, {int(inside.sum())} receivers inside, {len(bad)} rows differ")
            for b in bad[:5]:
                print("   in", in8[ok][b], "gpu", got[ok][b], "ora", want[ok][b])
            assert len(bad) == 0
