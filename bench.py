#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X ray-intersection / path-tracing core.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  path    C2 (default, BASELINE.json configs[1]): Cornell box, SimplePathIntegrator(max_depth 5, NEE, BSDF sampling),
          1024x1024 @ 64 spp per GPU.  A step = one whole Integrate::render() of the image plus — for N>1 — the
          film merge.  Metric: path samples/s.  N>1: sample split (rank r renders sample indices [64r, 64(r+1)) of a
          64*N-spp image), films merged by ONE RCCL all-reduce of the f64 film per step => weak scaling.
          The C3 closest-hit figure is measured in the same run and reported under "closest_hit".
  trace   C3: 1,048,576-triangle synthetic mesh, 4096x4096 primary rays per GPU, closest hit.  Metric: Mrays/s.
          Rays shard across ranks with no collective (every rank owns a replica of the BVH) => weak scaling.
Inputs (scene, BVH, rays) are resident in HBM before the timed region starts.

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
KERNEL_NAMES = ["raygen_kernel", "closest_kernel", "shade_kernel", "shadow_kernel", "accumulate_kernel"]


def pmc_traffic(kernel_regex, tag):
    """HBM bytes per launch of the kernel whose name matches `kernel_regex`, from the committed rocprofv3 --pmc passes of
    this same command (profiles/rNN/pmc_summary_<tag>.json, made by scripts/profile_gpu.sh + scripts/summarize_pmc.py):
    FETCH_SIZE doubled for wide loads as MI355X_MICROARCH.md prescribes, plus WRITE_SIZE.  None when no summary is
    committed.  The newest round's summary that holds the kernel wins."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_summary_{tag}.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        for name, v in d.items():
            if re.search(kernel_regex, name) and "fetch_bytes_x2_wide_loads" in v and "write_bytes" in v:
                return (v["fetch_bytes_x2_wide_loads"] + v["write_bytes"]) / max(1, v["dispatches"]), os.path.relpath(path, ROOT)
    return None, None


# timed (non-counting) variants of the kernel classes, as rocprofv3 names them: the first template argument of the
# render kernels (COUNT) and the second of trace_kernel / trace_flat_kernel is false
PMC_NAME = {"raygen_kernel": r"^raygen_kernel", "closest_kernel": r"^closest(_flat)?_kernel<false", "shade_kernel": r"^shade_kernel",
            "shadow_kernel": r"^shadow(_flat)?_kernel<false", "accumulate_kernel": r"^accumulate_kernel"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="path", choices=["path", "trace", "c4"])
    ap.add_argument("--res", type=int, default=None, help="image side (path: 1024, trace: 4096)")
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel per GPU (path workload)")
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--blobs", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="path workload: skip the C3 closest-hit figure")
    ap.add_argument("--cpu-sample", type=int, default=1 << 23, help="rays in the bounded CPU-baseline sample (trace; about 20 core-seconds)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time bound (path)")
    ap.add_argument("--max-leaf", type=int, default=4, help="max_prims_in_node of BVHAggregate::new (wide leaves hold <= 4)")
    ap.add_argument("--split", default="sah", choices=["sah", "hlbvh"],
                    help="BVH build: binned SAH on the host (default) or the linear BVH built on the device (triangle scenes)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse "
                                                       "several ranks on one GPU)")
    return ap.parse_args()


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as ora
    return ora


def cpu_baseline_trace(pos, idx, rays_sample, threads):
    """C++ restatement of the pbrt-rs CPU path (binary 64-B-node BVH, Middle split as in main.rs:217-221,
    per-node divides, per-ray heap stack) timed on this box's host cores."""
    ora = _oracle()
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


def cpu_baseline_path(pm, sd, spp, max_depth, threads, seconds):
    """The oracle's Integrate::render() (row-parallel like the reference's rayon loop, base.rs:50-63) on a bounded
    sample of the same workload: the same image, one sample index (one sample of every pixel) at a time, until
    `seconds` have elapsed."""
    ora = _oracle()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    oren = helpers.oracle_renderer_from(sd, ora)
    x0, y0, x1, y1 = sd.film["pixel_bounds"]
    band = (x0, y0, x1, y1)
    n, s, t0 = 0, 0, time.time()
    while s < spp and time.time() - t0 < seconds:
        p = helpers.render_params(pm, sd, spp, max_depth, True, True, 0, window=band, samples=(s, s + 1))
        n += oren.render(p, threads)["samples"]
        s += 1
    dt = time.time() - t0
    return {"value": n / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"rows {band[1]}..{band[3]} of the same {x1 - x0}x{y1 - y0} image, sample indices 0..{s} "
                      f"({n} path samples, {dt:.1f} s)"}


def run_trace(args, pm, torch, dist, rank, local_rank, world, steps, warmup, want_cpu):
    res = args.res or 4096
    pos, idx = pm.scenes.blob_mesh(args.blobs, 1024, seed=1)
    split = pm._capi.PM_SPLIT_HLBVH if args.split == "hlbvh" else pm._capi.PM_SPLIT_SAH
    agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)], max_prims_in_node=args.max_leaf, device_id=local_rank, split_method=split)
    rays = pm.scenes.primary_rays(res, res, eye=(0.02 * rank, 0.0, -4.0))
    n_rays = len(rays)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_hits = torch.empty((n_rays, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    # deterministic traversal counters (untimed pass with the counting kernel) -> algorithmic bytes
    agg.set_counting(True)
    agg.reset_counters()
    agg.intersect(d_rays, out=d_hits)
    ctr = agg.counters()
    agg.set_counting(False)
    n_node, n_prim = ctr["nodes_visited"] / n_rays, ctr["prims_tested"] / n_rays
    s_node, s_prim = agg.info.node_bytes, agg.info.prim_record_bytes[2]
    bytes_per_ray = 32 + 16 + n_node * s_node + n_prim * s_prim

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        agg.intersect(d_rays, out=d_hits)
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        agg.intersect(d_rays, out=d_hits)      # blocking: returns after the library's stream has drained
        kernel_ms.append(agg.last_kernel_ms())
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    hit_frac = float((d_hits[:, 0].view(torch.int32) != -1).float().mean().item())
    k_ms = float(np.mean(kernel_ms))
    achieved = bytes_per_ray * n_rays / (k_ms * 1e-3) / 1e9
    out = {
        "metric": "closest-hit Mrays/s", "value": n_rays * world * steps / dt / 1e6, "unit": "Mrays/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C3 closest-hit microbench: {len(idx)} triangles ({args.blobs} blobs x 1024), "
                               f"{res}x{res} primary rays per GPU in 8x8 tile order",
                   "rays_per_gpu": n_rays, "hit_fraction": hit_frac, "bvh_nodes": int(agg.info.n_nodes),
                   "bvh_build": args.split, "bvh_build_seconds": float(agg.info.build_seconds),
                   "bvh_node_bytes": int(s_node), "tri_record_bytes": int(s_prim)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(r"^trace(_flat)?_kernel<false, false", "trace")[0],
                     "traffic_source": pmc_traffic(r"^trace(_flat)?_kernel<false, false", "trace")[1],
                     "kernel": "trace_kernel<false,false>", "kernel_ms": k_ms,
                     "bytes_per_ray": bytes_per_ray, "nodes_per_ray": n_node, "prims_per_ray": n_prim,
                     "note": "achieved = DEMAND bytes (every node / record fetch of the walk counted as a miss, SURVEY 8d) per "
                             "second; this tree fits the L2s + Infinity Cache, so the HBM side moves far less (traffic = PMC "
                             "bytes per launch) and the kernel is bound by instruction issue, not by HBM"},
    }
    if not args.no_secondary:
        # SURVEY.md 8(d) C3's other two ray sets, each with its own counters and event-timed launches (not `value`)
        hits_host = np.empty(n_rays, dtype=pm.HIT_DTYPE)
        hits_host.view(np.float32).reshape(-1, 4)[:] = d_hits.cpu().numpy()
        sets = [("shadow_any_hit", pm.scenes.shadow_rays(rays, hits_host), True,
                 "any hit from every primary hit point towards a point light at (0,3,0), un-normalised direction, "
                 "t_max = 1 - 1e-4 (primary misses carry t_max = 0)"),
                ("incoherent_closest_hit", pm.scenes.incoherent_rays(n_rays, seed=2 + rank), False,
                 "closest hit, origins uniform on a sphere of radius 1.9 around the mesh, directions cosine-distributed "
                 "about the inward normal, in generation order (no sorting)")]
        for key, rset, any_hit, what in sets:
            note(f"secondary ray set: {key}")
            d_r = torch.from_numpy(rset.view(np.float32).reshape(-1, 8)).cuda()
            d_o = (torch.empty(n_rays, dtype=torch.uint8, device="cuda") if any_hit
                   else torch.empty((n_rays, 4), dtype=torch.float32, device="cuda"))
            call = agg.intersect_p if any_hit else agg.intersect
            agg.set_counting(True)
            agg.reset_counters()
            call(d_r, out=d_o)
            c2 = agg.counters()
            agg.set_counting(False)
            for _ in range(max(1, warmup)):
                call(d_r, out=d_o)
            ms = []
            for _ in range(steps):
                call(d_r, out=d_o)
                ms.append(agg.last_kernel_ms())
            ms = float(np.mean(ms))
            live = int((rset["tmax"] > 0).sum())
            nn, npr = c2["nodes_visited"] / n_rays, c2["prims_tested"] / n_rays
            bpr = 32 + (1 if any_hit else 16) + nn * s_node + npr * s_prim
            ach = bpr * n_rays / (ms * 1e-3) / 1e9
            frac_out = (float(d_o.float().mean().item()) if any_hit
                        else float((d_o[:, 0].view(torch.int32) != -1).float().mean().item()))
            out[key] = {"value": n_rays / (ms * 1e-3) / 1e6, "unit": "Mrays/s (this GPU, event-timed kernel)", "rays": what,
                        "kernel_ms": ms, "live_rays": live, "occluded_or_hit_fraction": frac_out,
                        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": ach / HBM_PEAK_GBS, "bytes_per_ray": bpr, "nodes_per_ray": nn,
                                     "prims_per_ray": npr, "fallback_walks": c2["fallback_walks"]}}
            del d_r, d_o
    if want_cpu and rank == 0:
        threads = min(16, len(os.sched_getaffinity(0)))
        out["cpu_baseline"] = cpu_baseline_trace(pos, idx, rays[:args.cpu_sample], threads)
    del agg
    return out


def kernel_bytes(st, info, sd):
    """Algorithmic HBM bytes of one render() per kernel class: structure-of-arrays path state that each
    kernel must read/write once per item + the scene data its rays need.  Tree walk: BVH node / leaf-record bytes from
    the traversal counters (demand traffic, every fetch counted).  Flat sweep (scenes of <= 64 primitives): the flat table
    is read once per WAVE of 64 rays through scalar loads, so a ray's share is 1/64 of the entries it was offered, plus
    the 48..80-byte record of every exact test it ran (those are per-lane loads)."""
    rec = info.prim_buffer_bytes / max(1, info.n_prims)     # mean leaf-record size of this scene
    node = info.node_bytes
    S, C, SH = st["samples"], st["closest_rays"], st["shadow_rays"]
    shaded = C                       # one shade item per closest-hit ray
    if info.flat_prims:
        ex_c = st["exact_tests"] * C / max(1, C + SH)       # the exact-test counter is not split by kernel class
        ex_s = st["exact_tests"] - ex_c
        scene_c = st["closest_prims"] * 64 / 64.0 + ex_c * rec
        scene_s = st["shadow_prims"] * 64 / 64.0 + ex_s * rec
    else:
        scene_c = st["closest_nodes"] * node + st["closest_prims"] * rec
        scene_s = st["shadow_nodes"] * node + st["shadow_prims"] * rec
    b = [0.0] * 5
    b[0] = S * (7 * 16 + 8)                                          # ray_o ray_d beta L lambda pdf rng + misc
    b[1] = C * (4 + 8 + 32 + 16) + scene_c
    b[2] = shaded * (4 + 8 + 16 + 16 + 16 + 16 + 16 + 8)             # queue misc hit beta lambda ray_d rng(16 r + 8 w)
    b[2] += (C - S) * (32 + 16 + 8 + 4) + SH * (48 + 4) + shaded * 72   # continuation + shadow-ray writes + shading record
    b[3] = SH * (4 + 32) + scene_s + SH * 48
    b[4] = S * (48 + 8) + (S / max(1, st["spp_rendered"])) * 64
    return b


def note(msg):
    """progress on stderr (stdout carries only the JSON line)"""
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def run_path(args, pm, torch, dist, rank, local_rank, world, steps, warmup):
    note("building scene")
    if args.workload == "c4":      # BASELINE.json configs[3]: 20,971,520 triangles, 3840x2160 @ 256 spp, uniform sky
        rw, rh = (args.res or 3840), ((args.res or 3840) * 9) // 16
        n_blobs = args.blobs if args.blobs != 1024 else 20480
        sd = pm.scenes.blob_sky_scene(n_blobs, (rw, rh))
        spp = args.spp if args.spp != 64 else 256
        name = f"C4 {n_blobs * 1024} diffuse triangles under a uniform sky: {rw}x{rh} @ {spp} spp per GPU"
    else:
        rw = rh = args.res or 1024
        sd = pm.scenes.cornell_scene((rw, rh))
        spp = args.spp
        name = f"C2 Cornell box path tracing: {rw}x{rh} @ {spp} spp per GPU"
    spp_total = spp * world
    integ = pm.WavefrontIntegrator(args.max_depth, True, True, sd, spp_total, seed=0, device_id=local_rank,
                                   split_method=pm._capi.PM_SPLIT_HLBVH if args.split == "hlbvh" else pm._capi.PM_SPLIT_SAH,
                                   max_prims_in_node=args.max_leaf)
    s0, s1 = pm.distributed.shard_samples(spp_total, rank, world)
    film = torch.empty((rh, rw, 4), dtype=torch.float64, device="cuda")

    def step():
        integ.clear()
        st = integ.render(samples=(s0, s1))
        if world > 1:
            integ.film(out=film)
            pm.distributed.allreduce_film(film)
        return st

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed counting pass -> algorithmic bytes per kernel class
    note(f"scene ready ({integ.aggregate.info.n_prims} primitives, {integ.aggregate.info.n_nodes} nodes); counting pass")
    integ.aggregate.set_counting(True)
    st_c = step()
    integ.aggregate.set_counting(False)
    st_c["spp_rendered"] = s1 - s0
    bytes_k = kernel_bytes(st_c, integ.aggregate.info, sd)
    note("timed steps")
    for _ in range(warmup):
        step()
    barrier()
    ms_k = np.zeros(5)
    launches = np.zeros(5)
    t0 = time.perf_counter()
    for _ in range(steps):
        st = step()
        ms_k += np.asarray(st["ms_by_kernel"])
        launches += np.asarray(st["launches_by_kernel"])
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_k /= steps
    launches /= steps
    samples = st["samples"]
    dom = int(np.argmax(ms_k))
    achieved = bytes_k[dom] / (ms_k[dom] * 1e-3) / 1e9
    total_bytes = float(sum(bytes_k))
    out = {
        "metric": "path samples/s", "value": samples * world * steps / dt, "unit": "samples/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name}, SimplePathIntegrator max_depth {args.max_depth}, light + BSDF sampling, "
                               f"Gaussian filter, 4 wavelengths/sample",
                   "bvh_nodes": int(integ.aggregate.info.n_nodes), "bvh_build": args.split,
                   "traversal": ("flat sweep of %d primitives (no tree walk)" % integ.aggregate.info.flat_prims
                                 if integ.aggregate.info.flat_prims else "wide-BVH walk"),
                   "bvh_build_seconds": float(integ.aggregate.info.build_seconds),
                   "samples_per_gpu": samples, "closest_rays": st["closest_rays"], "shadow_rays": st["shadow_rays"],
                   "passes": st["passes"], "merge": "none" if world == 1 else ("RCCL" if args.backend == "nccl" else args.backend) + " all-reduce of the f64 film per step",
                   "bytes_per_sample": total_bytes / samples,
                   "closest_nodes_per_ray": st_c["closest_nodes"] / max(1, st_c["closest_rays"]),
                   "closest_prims_per_ray": st_c["closest_prims"] / max(1, st_c["closest_rays"]),
                   "shadow_nodes_per_ray": st_c["shadow_nodes"] / max(1, st_c["shadow_rays"]),
                   "shadow_prims_per_ray": st_c["shadow_prims"] / max(1, st_c["shadow_rays"]),
                   "exact_tests_per_ray": st_c["exact_tests"] / max(1, st_c["closest_rays"] + st_c["shadow_rays"]),
                   "fallback_walks": st_c["fallback_walks"],
                   "kernel_ms": dict(zip(KERNEL_NAMES, [float(x) for x in ms_k])),
                   "kernel_GBps": dict(zip(KERNEL_NAMES, [float(b / (m * 1e-3) / 1e9) if m > 0 else 0.0
                                                          for b, m in zip(bytes_k, ms_k)]))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(PMC_NAME[KERNEL_NAMES[dom]], "path")[0] if args.workload == "path" else None,
                     "traffic_source": pmc_traffic(PMC_NAME[KERNEL_NAMES[dom]], "path")[1] if args.workload == "path" else None,
                     "kernel": KERNEL_NAMES[dom],
                     "kernel_ms": float(ms_k[dom] / max(1.0, launches[dom])), "launches_per_step": float(launches[dom]),
                     "bytes_per_launch": bytes_k[dom] / max(1.0, launches[dom]),
                     "whole_render": {"achieved": total_bytes / (float(ms_k.sum()) * 1e-3) / 1e9,
                                      "frac": total_bytes / (float(ms_k.sum()) * 1e-3) / 1e9 / HBM_PEAK_GBS},
                     "per_kernel_frac": {k: float(b / (m * 1e-3) / 1e9 / HBM_PEAK_GBS) if m > 0 else 0.0
                                         for k, b, m in zip(KERNEL_NAMES, bytes_k, ms_k)}},
    }
    if integ.aggregate.info.flat_prims and KERNEL_NAMES[dom] in ("closest_kernel", "shadow_kernel"):
        out["roofline"]["note"] = ("the traversal kernels of this %d-primitive scene are instruction-bound by design: every wave "
                                   "sweeps all primitives from wave-uniform (scalar-loaded) records, so their algorithmic HBM "
                                   "bytes are a few tens of bytes per ray; the HBM-bound kernel of the frame is shade_kernel "
                                   "(per_kernel_frac)" % integ.aggregate.info.flat_prims)
    return out, sd, integ


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only path (no CPU fallback)")
    local_rank %= torch.cuda.device_count()       # (rehearsing N ranks on fewer GPUs; a no-op on a full node)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    pm = importlib.import_module("pbrt-rs_amd")
    want_cpu = world == 1 and not args.no_cpu_baseline

    if args.workload == "trace":
        out = run_trace(args, pm, torch, dist, rank, local_rank, world, args.steps, args.warmup, want_cpu)
    else:
        out, sd, integ = run_path(args, pm, torch, dist, rank, local_rank, world, args.steps, args.warmup)
        if want_cpu and rank == 0 and args.workload == "path":
            threads = min(16, len(os.sched_getaffinity(0)))
            out["cpu_baseline"] = cpu_baseline_path(pm, sd, args.spp, args.max_depth, threads, args.cpu_seconds)
        del integ
        if not args.no_secondary and args.workload == "path":
            tr = run_trace(args, pm, torch, dist, rank, local_rank, world, max(3, args.steps), 1, False)
            out["closest_hit"] = {"value": tr["value"], "unit": tr["unit"], "workload": tr["config"]["workload"],
                                  "ms_per_step": tr["ms_per_step"], "roofline": tr["roofline"]}
            for key in ("shadow_any_hit", "incoherent_closest_hit"):
                if key in tr:
                    out["closest_hit"][key] = tr[key]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
