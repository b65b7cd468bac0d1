"""Traversal-stack overflow: the walk keeps PM_STACK_TOTAL = 48 entries per ray; beyond that it drops a
group of siblings and the call must return PM_ERR_STACK_OVERFLOW ("results invalid") - never crash, hang or silently return.

A ray pushes at most one entry per tree level, so provoking 48 needs a tree more than 48 wide levels deep with two hit
interior siblings on every level; the device LBVH cannot get there (binary depth <= 63 + log2 n, three binary levels per
wide level on average) and SAH trees of ordinary scenes are ~10 levels deep.  Instead of a contrived scene the test uses
`libpbrt_mi355_smallstack.so`: the SAME sources built with a 3-entry stack (`make smallstack`, -DPM_STACK_TOTAL=3 -DPM_LDS_STACK=2; measured: 8 entries are already enough for a 262 144-triangle SAH tree), on which
ordinary scenes overflow.  Each case runs in a child process (the library is chosen at import time).
"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = os.path.join(ROOT, "pbrt-rs_amd", "libpbrt_mi355_smallstack.so")


def _run(body, lib=SMALL):
    code = textwrap.dedent("""
        import importlib, sys
        import numpy as np
        sys.path.insert(0, %r)
        pm = importlib.import_module("pbrt-rs_amd")
    """ % ROOT) + textwrap.dedent(body)
    env = dict(os.environ, PBRT_MI355_LIB=lib)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=110, env=env)


def test_small_stack_build_exists():
    assert os.path.exists(SMALL), "run __graft_entry__.build() (make smallstack)"


def test_overflow_is_reported_by_trace_and_render():
    r = _run("""
        pos, idx = pm.scenes.blob_mesh(256, 1024, seed=1)
        agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)])
        rays = pm.scenes.incoherent_rays(1 << 18, seed=2)
        import torch
        d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
        d_out = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
        for name, call in (("closest", agg.intersect), ("any", agg.intersect_p), ("noalloc", lambda r: agg.intersect_noalloc(d_rays, d_out))):
            try:
                call(rays)
                print(name, "NO ERROR")
            except Exception as e:
                print(name, type(e).__name__, getattr(e, "status", None), e)
        # the context stays usable: a shallow scene afterwards traces without complaint
        p2, i2 = pm.scenes.blob_mesh(1, 128, seed=3, blob_sigma=0.3, tri_size=0.2)
        agg2 = pm.GpuAggregate([pm.TriangleMesh(p2, i2)])
        h = agg2.intersect(pm.scenes.primary_rays(64, 64))
        print("shallow hits", int((h["prim_id"] != pm.PM_MISS).sum()))
        # render: the C4-shaped scene walks the same tree
        sd = pm.scenes.blob_sky_scene(n_blobs=64, resolution=(64, 64))
        integ = pm.WavefrontIntegrator(5, True, True, sd, 2)
        try:
            integ.render()
            print("render NO ERROR")
        except Exception as e:
            print("render", type(e).__name__, e)
    """)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    for name in ("closest", "any", "noalloc", "render"):
        line = [l for l in out.splitlines() if l.startswith(name + " ")]
        assert line and "NO ERROR" not in line[0] and "overflow" in line[0].lower(), out
    shallow = [l for l in out.splitlines() if l.startswith("shallow hits")]
    assert shallow and int(shallow[0].split()[-1]) > 0, out


def test_same_calls_succeed_with_the_real_stack():
    r = _run("""
        pos, idx = pm.scenes.blob_mesh(256, 1024, seed=1)
        agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)])
        rays = pm.scenes.incoherent_rays(1 << 18, seed=2)
        import torch
        d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
        d_out = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
        agg.intersect(rays); agg.intersect_p(rays); agg.intersect_noalloc(d_rays, d_out)
        print("ok", agg.counters()["stack_overflows"])
    """, lib=os.path.join(ROOT, "pbrt-rs_amd", "libpbrt_mi355.so"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().endswith("ok 0")
