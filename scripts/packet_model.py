#!/usr/bin/env python3
"""PREDICT the gain of a wave-packet walk before building it.  CPU only.  Builds the C3 scene's
GPU-layout tree on a host-only context, takes the bench's ray sets in the bench's order (8x8 pixel tiles = one wave),
and compares
  * the walk the kernel runs today: persistent waves, every lane walks its own ray (oracle ora_wide_wave_model_persistent:
    wave node steps and prim trips per 64 rays), with
  * a packet walk: the 64 rays of a wave share one stack, a node is visited when ANY lane wants it
    (oracle ora_wide_packet_model: union node steps and union prim trips per 64 rays).
usage: packet_model.py [blobs=1024] [res=4096] [waves=8192]   (a strided sample of whole waves is modelled)"""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as ora
pm = importlib.import_module("pbrt-rs_amd")
blobs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
res = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_waves = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
pos, idx = pm.scenes.blob_mesh(blobs, 1024, seed=1)
agg = pm.GpuAggregate([pm.TriangleMesh(pos, idx)], device_id=-1)
nb, pb = agg.node_buffer(), agg.prim_buffer()
prim = pm.scenes.primary_rays(res, res)
# sample: runs of 64 consecutive waves (what a persistent wave's refills see), strided over the image
total_waves = len(prim) // 64
run = 64
starts = np.linspace(0, total_waves - run, max(1, n_waves // run)).astype(np.int64)
sel = (starts[:, None] * 64 + np.arange(run * 64)[None, :]).reshape(-1)
prim_s = np.ascontiguousarray(prim[sel])
hits, _ = ora.wide_trace(nb, pb, prim_s)
sets = {"primary": prim_s, "shadow": pm.scenes.shadow_rays(prim_s, hits), "incoherent": pm.scenes.incoherent_rays(len(prim_s), seed=2)}
L = ora.lib()
out = {"scene": f"{blobs} blobs x 1024 triangles", "res": res, "rays_modelled": int(len(prim_s))}
for name, rays in sets.items():
    for any_hit in ((0, 1) if name == "shadow" else (0,)):
        if name == "shadow" and not any_hit:
            continue
        n = len(rays)
        o6 = np.zeros(6, np.uint64)
        L.ora_wide_wave_model_persistent(C.c_void_p(nb.ctypes.data), C.c_void_p(pb.ctypes.data), C.c_void_p(rays.ctypes.data),
                                         C.c_uint64(n), any_hit, 64, 24, 1, C.c_void_p(o6.ctypes.data))
        rec = {"per_lane_walk": {"wave_node_steps_per_64": float(o6[2]) * 64 / n, "wave_prim_trips_per_64": float(o6[3]) * 64 / n,
                                 "lane_node_steps_per_ray": float(o6[4]) / n, "lane_prim_tests_per_ray": float(o6[5]) / n,
                                 "lane_util_node_steps": float(o6[4]) / (64.0 * float(o6[2]))}}
        if not any_hit:
            o8 = np.zeros(8, np.uint64)
            L.ora_wide_packet_model(C.c_void_p(nb.ctypes.data), C.c_void_p(pb.ctypes.data), C.c_void_p(rays.ctypes.data),
                                    C.c_uint64(n), C.c_void_p(o8.ctypes.data), 8)
            w = float(o8[0])
            rec["packet_walk"] = {"union_node_steps_per_64": float(o8[1]) / w, "union_prim_trips_per_64": float(o8[2]) / w,
                                  "lane_participation": float(o8[3]) / (64.0 * float(o8[1])),
                                  "lane_prim_participation": float(o8[4]) / max(1.0, 64.0 * float(o8[2])),
                                  "mixed_octant_waves": float(o8[5]) / w, "max_stack": int(o8[6]),
                                  "steps_under_16_lanes": float(o8[7]) / float(o8[1])}
            # cost model in VALU issue slots per 64 rays.  Today's step: 320 instructions per node step, ~70 per prim trip
            # (profiles/r02 closest_kernel: 38.9 node trips x 320 + 9.3 prim trips x 70 per 64 primary rays).  Packet step:
            # node through the scalar path (one fetch per wave), 48 ubyte converts + 24 set-up + 8 x 13 slab ops + stack on
            # the scalar unit: ~190; prim trip with a wave-uniform triangle: ~60.
            today = rec["per_lane_walk"]["wave_node_steps_per_64"] * 320 + rec["per_lane_walk"]["wave_prim_trips_per_64"] * 70
            packet = rec["packet_walk"]["union_node_steps_per_64"] * 190 + rec["packet_walk"]["union_prim_trips_per_64"] * 60
            rec["predicted_speedup_valu"] = today / packet
        out[name + ("_any" if any_hit else "")] = rec
print(json.dumps(out, indent=1))
