"""Dev probe (GPU): do two renders on two streams overlap usefully?  Two contexts (each has its own stream and path state)
render half of the C2 frame's sample indices each from two host threads; compared with one context rendering all of them.
Says whether a two-stream frame inside pm_render is worth building (a VALU-bound sweep beside a memory-waiting shade kernel)."""
import importlib
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")

spp = int(os.environ.get("SPP", "64"))
steps = int(os.environ.get("STEPS", "8"))
scene = os.environ.get("SCENE", "c2")
res = int(os.environ.get("RES", "1024"))
parts = int(os.environ.get("PARTS", "2"))
mk = {"c2": pm.scenes.c2_scene, "c5": pm.scenes.c5_scene, "c2pl": pm.scenes.cornell_scene}[scene]
integrator = pm._capi.PM_INTEGRATOR_VOL_PATH if scene == "c5" else pm._capi.PM_INTEGRATOR_SIMPLE_PATH
depth = 16 if scene == "c5" else 5


def make():
    return pm.WavefrontIntegrator(depth, True, True, mk((res, res)), spp, seed=0, integrator=integrator)


one = make()
for _ in range(2):
    one.clear(); one.render()
t0 = time.perf_counter()
for _ in range(steps):
    one.clear(); st = one.render()
t_one = (time.perf_counter() - t0) / steps
print(f"one context, {spp} spp: {t_one * 1e3:.2f} ms wall per frame (kernel_ms {st['kernel_ms']:.2f}), by class {[round(x, 2) for x in st['ms_by_kernel']]}", flush=True)

# the same frame as `parts` sequential calls of one context (what splitting alone costs)
t0 = time.perf_counter()
for _ in range(steps):
    one.clear()
    for k in range(parts):
        one.render(samples=(k * spp // parts, (k + 1) * spp // parts))
t_seq = (time.perf_counter() - t0) / steps
print(f"one context, {parts} sequential calls: {t_seq * 1e3:.2f} ms wall per frame", flush=True)

ctxs = [one] + [make() for _ in range(parts - 1)]
for k, c in enumerate(ctxs):
    c.clear(); c.render(samples=(k * spp // parts, (k + 1) * spp // parts))


def worker(k, c, n):
    for _ in range(n):
        c.clear()
        c.render(samples=(k * spp // parts, (k + 1) * spp // parts))


t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(k, c, steps)) for k, c in enumerate(ctxs)]
for t in th:
    t.start()
for t in th:
    t.join()
t_two = (time.perf_counter() - t0) / steps
print(f"{parts} contexts on {parts} threads, {spp // parts} spp each: {t_two * 1e3:.2f} ms wall per frame "
      f"({t_one / t_two:.3f} x one context, {t_seq / t_two:.3f} x the sequential split)", flush=True)
