"""The N>1 path on the CPU: two processes, gloo backend.  Each rank renders its share of the image with the ORACLE
renderer standing in for the GPU (the split/merge logic in pbrt-rs_amd/distributed.py is device-agnostic: it works on
torch tensors), merges with the same collectives bench.py uses, and the result must equal the single-process render:
  * sample split + all-reduce of the f64 film: equal to 1e-14 relative (one extra f64 add per pixel);
  * row split + all-gather of bands: bit-identical."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES, SPP = (40, 24), 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    pm = importlib.import_module("pbrt-rs_amd")
    import helpers
    import oracle as ora
    sd = pm.scenes.main_rs_scene(RES)
    return pm, ora, helpers, sd, helpers.oracle_renderer_from(sd, ora)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pm, ora, helpers, sd, oren = _setup()
        d = pm.distributed
        # sample split
        s0, s1 = d.shard_samples(SPP, rank, world)
        oren.clear()
        oren.render(helpers.render_params(pm, sd, SPP, samples=(s0, s1)), 1)
        film = torch.from_numpy(oren.film().copy())
        d.allreduce_film(film)
        # row split
        win = d.shard_rows(sd.film["pixel_bounds"], rank, world)
        oren.clear()
        oren.render(helpers.render_params(pm, sd, SPP, window=win), 1)
        film_rows = d.allgather_rows(torch.from_numpy(oren.film().copy()), sd.film["pixel_bounds"])
        np.save(os.path.join(out_dir, f"samples_{rank}.npy"), film.numpy())
        np.save(os.path.join(out_dir, f"rows_{rank}.npy"), film_rows.numpy())
        # the asynchronous forms bench.py uses to overlap the collective with the next frame's render: same results
        oren.clear()
        oren.render(helpers.render_params(pm, sd, SPP, samples=(s0, s1)), 1)
        film2 = torch.from_numpy(oren.film().copy())
        work = d.allreduce_film(film2, async_op=True)
        oren.clear()
        oren.render(helpers.render_params(pm, sd, SPP, window=win), 1)          # ("the next frame", while the collective runs)
        assemble, work2 = d.allgather_rows(torch.from_numpy(oren.film().copy()), sd.film["pixel_bounds"], async_op=True)
        work.wait()
        work2.wait()
        np.save(os.path.join(out_dir, f"samples_async_{rank}.npy"), film2.numpy())
        np.save(os.path.join(out_dir, f"rows_async_{rank}.npy"), assemble().numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_split_and_merge(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    pm, ora, helpers, sd, oren = _setup()
    oren.render(helpers.render_params(pm, sd, SPP), 2)
    whole = oren.film()
    assert whole[..., 3].min() > 0
    for r in range(world):
        a = np.load(tmp_path / f"samples_{r}.npy")
        np.testing.assert_allclose(a, whole, rtol=1e-14, atol=0)
        b = np.load(tmp_path / f"rows_{r}.npy")
        assert np.array_equal(b.view(np.uint64), whole.view(np.uint64))
        assert np.array_equal(np.load(tmp_path / f"samples_async_{r}.npy").view(np.uint64), a.view(np.uint64))
        assert np.array_equal(np.load(tmp_path / f"rows_async_{r}.npy").view(np.uint64), whole.view(np.uint64))
