#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X ray-intersection / path-tracing core.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  trace   C3: ~1M-triangle synthetic mesh, 4096x4096 primary rays, closest-hit (metric: Mrays/s)
A step = one pass of the hot path over one batch of synthetic rays that is already resident in HBM.
Rays shard across ranks with no data-path collective (every rank traces its own batch against its own
replica of the BVH) => weak scaling.

The oracle (oracle/) is used ONLY for the cpu_baseline leg, on rank 0 at N=1.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="trace", choices=["trace"])
    ap.add_argument("--res", type=int, default=4096, help="primary-ray image side (trace workload)")
    ap.add_argument("--blobs", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1 << 21, help="rays in the bounded CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline_trace(pos, idx, rays_sample, threads):
    """C++ restatement of the pbrt-rs CPU path (binary 64-B-node BVH, Middle split as in main.rs:217-221,
    per-node divides, per-ray heap stack) timed on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as ora
    sc = ora.Scene()
    sc.add_triangle_mesh(pos, idx)
    t0 = time.time()
    sc.build(ora.SPLIT_MIDDLE, 255)
    build_s = time.time() - t0
    t0 = time.time()
    hits, ctr = sc.intersect(rays_sample, nthreads=threads)
    dt = time.time() - t0
    return {"value": len(rays_sample) / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"first {len(rays_sample)} primary rays of the same batch, closest hit; "
                      f"binary-BVH build {build_s:.1f}s not included",
            "nodes_per_ray": ctr["nodes_visited"] / len(rays_sample), "prims_per_ray": ctr["prims_tested"] / len(rays_sample)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only path (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    pm = importlib.import_module("pbrt-rs_amd")

    # ---- synthetic scene + rays (same geometry on every rank; rank-dependent camera offset so batches differ)
    pos, idx = pm.scenes.blob_mesh(args.blobs, 1024, seed=1)
    agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)], device_id=local_rank)
    rays = pm.scenes.primary_rays(args.res, args.res, eye=(0.02 * rank, 0.0, -4.0))
    n_rays = len(rays)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_hits = torch.empty((n_rays, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    # ---- deterministic traversal counters (untimed pass with the counting kernel) -> algorithmic bytes
    agg.set_counting(True)
    agg.reset_counters()
    agg.intersect(d_rays, out=d_hits)
    ctr = agg.counters()
    agg.set_counting(False)
    n_node = ctr["nodes_visited"] / n_rays
    n_prim = ctr["prims_tested"] / n_rays
    s_node = agg.info.node_bytes
    s_prim = agg.info.prim_record_bytes[2]
    bytes_per_ray = 32 + 16 + n_node * s_node + n_prim * s_prim

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        agg.intersect(d_rays, out=d_hits)
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agg.intersect(d_rays, out=d_hits)      # blocking: returns after the library's stream has drained
        kernel_ms.append(agg.last_kernel_ms())
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    hit_frac = float((d_hits[:, 0].view(torch.int32) != -1).float().mean().item())

    if rank == 0:
        value = n_rays * world * args.steps / dt / 1e6
        k_ms = float(np.mean(kernel_ms))
        achieved = bytes_per_ray * n_rays / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "closest-hit Mrays/s", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3 closest-hit microbench: {len(idx)} triangles ({args.blobs} blobs x 1024), "
                                   f"{args.res}x{args.res} primary rays per GPU in 8x8 tile order",
                       "rays_per_gpu": n_rays, "hit_fraction": hit_frac, "bvh_nodes": int(agg.info.n_nodes),
                       "bvh_node_bytes": int(s_node), "tri_record_bytes": int(s_prim)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "trace_kernel<false,false>", "kernel_ms": k_ms,
                         "bytes_per_ray": bytes_per_ray, "nodes_per_ray": n_node, "prims_per_ray": n_prim},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, len(os.sched_getaffinity(0)))
            out["cpu_baseline"] = cpu_baseline_trace(pos, idx, rays[:args.cpu_sample], threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
