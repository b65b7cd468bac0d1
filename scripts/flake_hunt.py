"""debug: find the closest-hit ray of the C4 frame whose result depends on its wave (pixel 290,630)."""
import importlib, os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
pm = importlib.import_module("pbrt-rs_amd")
import helpers, oracle as ora
PX, PY = 290, 630
sd = pm.scenes.blob_sky_scene(20480, (3840, 2160))
integ = pm.WavefrontIntegrator(5, True, True, sd, 2, split_method=pm._capi.PM_SPLIT_HLBVH)
agg = integ.aggregate
t0 = time.time()
oren = helpers.oracle_renderer_from(sd, ora)
print("oracle tree", time.time() - t0, flush=True)
L = ora.lib()
L.ora_renderer_path_log.restype = C.c_int
L.ora_renderer_path_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
p = helpers.render_params(pm, sd, 2, 5)
with ora.correctly_rounded():
    logs = []
    for dy in range(-8, 8):
        for dx in range(-8, 8):
            for s in (0, 1):
                buf = np.zeros((64, 10), np.float32)
                n = L.ora_renderer_path_log(oren._h, C.addressof(p), PX + dx, PY + dy, s, buf.ctypes.data, 64)
                for i in range(n):
                    logs.append((dx, dy, s, i) + tuple(buf[i]))
    for s in (0, 1):
        print("GPU single-lane sample", s, integ.debug_sample(PX, PY, s)[:4], "oracle", oren.sample(p, PX, PY, s)[:4])
logs = np.array(logs, np.float32)
print("rays logged", len(logs))
mine = logs[(logs[:, 0] == 0) & (logs[:, 1] == 0)]
for r in mine:
    print("path ray: s=%d i=%d kind=%d o=%s d=%s tmax=%g -> id/occ=%d t=%g" % (r[2], r[3], r[11], r[4:7], r[7:10], r[10], np.float32(r[12]).view(np.uint32), r[13]))
clo = logs[logs[:, 11] == 0]
rays = np.zeros(len(clo), pm.RAY_DTYPE)
rays["o"] = clo[:, 4:7]; rays["d"] = clo[:, 7:10]; rays["tmax"] = clo[:, 10]
nb, pb = agg.node_buffer(), agg.prim_buffer()
cpu, _ = ora.wide_trace(nb, pb, rays, nthreads=16)
ref_id = clo[:, 12].copy().view(np.uint32)
print("mirror vs reference traversal: id mismatches", int((cpu["prim_id"] != ref_id).sum()), "t mismatches", int((cpu["t"].view(np.uint32) != clo[:, 13].copy().view(np.uint32)).sum()))
rng = np.random.default_rng(1)
bad = {}
for it in range(300):
    perm = rng.permutation(len(rays))
    reps = np.concatenate([perm, rng.permutation(len(rays)), rng.permutation(len(rays))])
    g = agg.intersect(np.ascontiguousarray(rays[reps]))
    mism = (g["prim_id"] != cpu["prim_id"][reps]) | (g["t"].view(np.uint32) != cpu["t"][reps].view(np.uint32))
    for k in np.nonzero(mism)[0]:
        bad.setdefault(int(reps[k]), []).append((int(g["prim_id"][k]), float(g["t"][k])))
print("rays whose GPU result differed from the mirror in some composition:", {k: (len(v), v[:3]) for k, v in bad.items()})
for k in bad:
    print("ray", k, "pixel d", clo[k, :4], "o", rays["o"][k], "d", rays["d"][k], "mirror", cpu[k], "reference id", ref_id[k], "t", clo[k, 13])
    np.save(os.path.join(ROOT, "gpurun_out", f"flake_ray_{k}.npy"), rays[k:k + 1])
