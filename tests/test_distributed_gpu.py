"""The N>1 path on the GPU box: two processes sharing the one card (gloo rendezvous; on a full node bench.py uses RCCL, one
rank per GPU).  Each rank builds the scene in its own context, renders its share with the HIP path through
pbrt-rs_amd/distributed.py::render_distributed, and the merged film must equal the single-process render:
  * sample split + all-reduce: equal to 1e-14 relative (one extra f64 add per pixel);
  * row split + all-gather: bit-identical.
Also runs bench.py itself under torch.distributed.run with two ranks (weak / samples and strong / rows)."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES, SPP = (72, 40), 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _integrator():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    pm = importlib.import_module("pbrt-rs_amd")
    return pm, pm.WavefrontIntegrator(5, True, True, pm.scenes.main_rs_scene(RES), SPP)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pm, integ = _integrator()
        for mode in ("samples", "rows"):
            film = pm.distributed.render_distributed(integ, mode)
            np.save(os.path.join(out_dir, f"{mode}_{rank}.npy"), film.cpu().numpy())
            np.save(os.path.join(out_dir, f"{mode}_loaded_{rank}.npy"), integ.film())     # what film_add put back
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu_split_and_merge(tmp_path, world, monkeypatch):
    """2 and 4 ranks sharing the one card (RES = 72 x 40: five 8-row units over four ranks = ragged row bands 16 / 8 / 8 / 8).
    The four-rank case renders on two lanes per rank (eight streams on the card); the single-process reference on one."""
    if world == 4:
        monkeypatch.setenv("PBRT_MI355_LANES", "2")       # (inherited by the spawned ranks)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    monkeypatch.setenv("PBRT_MI355_LANES", "1")
    pm, integ = _integrator()
    integ.render()
    whole = integ.film()
    assert whole[..., 3].min() > 0
    for r in range(world):
        for name in ("samples", "samples_loaded"):
            np.testing.assert_allclose(np.load(tmp_path / f"{name}_{r}.npy"), whole, rtol=1e-14, atol=0)
        for name in ("rows", "rows_loaded"):
            assert np.array_equal(np.load(tmp_path / f"{name}_{r}.npy").view(np.uint64), whole.view(np.uint64)), name


@pytest.mark.parametrize("extra", [[], ["--scaling", "strong", "--merge", "rows"]])
def test_bench_two_ranks(extra):
    """bench.py under torch.distributed.run, two ranks on the one card: one JSON line from rank 0, n_gpus = 2"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--res", "256", "--spp", "4", "--no-secondary", "--backend", "gloo"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["unit"] == "samples/s"
    assert d["scaling"] == ("strong" if extra else "weak")
    total_spp = 4 if extra else 8
    assert f"@ {total_spp} spp in total" in d["config"]["workload"]
    assert "roofline" in d and d["roofline"]["frac"] > 0


def test_bench_two_ranks_default_legs():
    """the DEFAULT N > 1 line (no --no-secondary): the weak C2 value plus the c4_strong leg - BASELINE config 4's split of a
    fixed sample count over the ranks, films merged by one all-reduce overlapped with the next frame - here at toy size"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--res", "256", "--spp", "4", "--blobs", "8", "--backend", "gloo"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 4000
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "@ 8 spp in total" in d["config"]["workload"]
    leg = d["c4_strong"]
    assert leg["value"] > 0 and leg["unit"] == "samples/s" and "8192 diffuse triangles" in leg["workload"]
    full = json.load(open(os.path.join(ROOT, d["full"])))
    assert full["c4_strong"]["scaling"] == "strong" and "@ 4 spp in total (2 sample indices per GPU)" in full["c4_strong"]["workload"]
    assert "overlapped" in full["config"]["merge"]
