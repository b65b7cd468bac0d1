"""GPU box: randomised differential test of the two trace entry points against the oracle (test infrastructure), beyond what
tests/ pins down: random mixes of triangles, bilinear patches and (partial, rotated, scaled) spheres at random sizes and
densities, through every tree layout and walk variant the library has (flat sweep on / off, one tree / a tree per type, SAH /
device LBVH), with primary, incoherent, aimed, axis-parallel, short and degenerate rays.

    python scripts/fuzz_trace.py [--iters N] [--seed S]   ->  gpurun_out/fuzz_trace.log, exit 1 on a real mismatch

What counts as a mismatch: a primitive id that differs where the two t are NOT bit-equal (bit-equal t with different ids is
the documented exact-tie class: the reference keeps whichever its tree meets first; two overlapping spheres within
2e-5 of each other's t are the other documented order-dependent class), any t / u / v bit difference on
an equal id (u, v of spheres: 1e-5), any any-hit flag that differs from the oracle's, any counter that differs from the CPU
walk of the exported tree, any fallback walk."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
MISS = 0xFFFFFFFF


def rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_scene(pm, ora, rng):
    """random primitive lists, the product aggregate and the oracle's scene; returns (agg, sc, n_prims, sphere id range, centres)"""
    kinds = rng.choice(["tri", "patch", "sphere", "tri+patch", "tri+sphere", "patch+sphere", "all"])
    spec, centres = [], []
    nid = 0
    sigma = float(rng.choice([0.02, 0.06, 0.15, 0.4]))
    if kinds in ("tri", "tri+patch", "tri+sphere", "all"):
        nb, per = int(rng.choice([1, 2, 8, 32])), int(rng.choice([3, 16, 100, 512, 1024]))
        pos, idx = pm.scenes.blob_mesh(nb, per, seed=int(rng.integers(1 << 30)), blob_sigma=sigma, tri_size=float(rng.choice([0.004, 0.012, 0.05, 0.3])))
        spec.append(("tri", pos, idx))
        centres.append(pos.reshape(-1, 3, 3).mean(1))
        nid += len(idx)
    if kinds in ("patch", "tri+patch", "patch+sphere", "all"):
        nb, per = int(rng.choice([1, 2, 8, 32])), int(rng.choice([2, 16, 100, 512]))
        pos, idx = pm.scenes.blob_patch_mesh(nb, per, seed=int(rng.integers(1 << 30)), blob_sigma=sigma, size=float(rng.choice([0.004, 0.012, 0.05, 0.3])),
                                             warp=float(rng.choice([0.0, 0.35, 1.0])))
        spec.append(("patch", pos, idx))
        centres.append(pos.reshape(-1, 4, 3).mean(1))
        nid += len(idx)
    sph = range(nid, nid)
    if kinds in ("sphere", "tri+sphere", "patch+sphere", "all"):
        ns = int(rng.choice([1, 3, 20, 150]))
        sph = range(nid, nid + ns)
        cs = []
        for _ in range(ns):
            scale = np.diag(rng.uniform(0.5, 2.0, 3)) if rng.random() < 0.4 else np.eye(3)
            m = np.eye(4)
            m[:3, :3] = rot(rng) @ scale if rng.random() < 0.7 else scale
            m[:3, 3] = rng.uniform(-1.0, 1.0, 3)
            r = float(rng.uniform(0.01, 0.25))
            if rng.random() < 0.4:
                zmin, zmax, phi = -r * rng.uniform(0.2, 1.0), r * rng.uniform(0.2, 1.0), float(rng.uniform(60, 360))
            else:
                zmin, zmax, phi = -r, r, 360.0
            spec.append(("sphere", r, float(zmin), float(zmax), phi, m.astype(np.float32), np.linalg.inv(m).astype(np.float32)))
            cs.append(m[:3, 3])
        centres.append(np.array(cs))
        nid += ns
    prims = []
    sc = ora.Scene()
    for p in spec:
        if p[0] == "sphere":
            _, r, zmin, zmax, phi, m, mi = p
            prims.append(pm.Sphere(r, zmin, zmax, phi, m, mi))
            sc.add_sphere(r, zmin, zmax, phi, m, mi)
        elif p[0] == "patch":
            prims.append(pm.BilinearPatchMesh(p[1], p[2]))
            sc.add_patch_mesh(p[1], p[2])
        else:
            prims.append(pm.TriangleMesh(p[1], p[2]))
            sc.add_triangle_mesh(p[1], p[2])
    split, mpn = int(rng.choice([0, 1, 2, 3])), int(rng.choice([1, 4, 255]))     # PM_SPLIT_HLBVH / SAH / MIDDLE / EQUAL_COUNTS
    agg = pm.GpuAggregate(prims, max_prims_in_node=mpn, split_method=split)
    kinds = f"{kinds}/s{split}m{mpn}"
    sc.build(int(rng.choice([ora.SPLIT_MIDDLE, ora.SPLIT_SAH])), 255)
    return kinds, agg, sc, nid, sph, np.concatenate(centres).astype(np.float32)


def make_rays(pm, rng, centres):
    from importlib import import_module
    RAY = import_module("pbrt-rs_amd")._capi.RAY_DTYPE
    parts = [pm.scenes.primary_rays(128, 128), pm.scenes.incoherent_rays(40_000, seed=int(rng.integers(1 << 30)))]
    # aimed at primitives from random origins (inside the scene too), some with a finite t_max around the target's distance
    n = 40_000
    k = rng.integers(0, len(centres), n)
    o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    d = (centres[k] + rng.normal(0, 0.01, (n, 3)).astype(np.float32) - o).astype(np.float32)
    aimed = np.zeros(n, dtype=RAY)
    aimed["o"], aimed["d"] = o, d                           # un-normalised: the hit is near t = 1
    aimed["tmax"] = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0.5, 1.5, n)).astype(np.float32)
    parts.append(aimed)
    # axis-parallel (zero direction components), from a grid of origins
    n = 6_000
    ax = np.zeros(n, dtype=RAY)
    a = rng.integers(0, 3, n)
    o = rng.uniform(-1.2, 1.2, (n, 3)).astype(np.float32)
    o[np.arange(n), a] = -3.0
    dd = np.zeros((n, 3), np.float32)
    dd[np.arange(n), a] = rng.choice([1.0, 2.5], n)
    two = rng.random(n) < 0.3                               # one more non-zero component for some
    dd[two, (a[two] + 1) % 3] = rng.normal(0, 0.2, two.sum())
    neg = rng.random(n) < 0.5
    o[neg] = -o[neg]; dd[neg] = -dd[neg]
    ax["o"], ax["d"], ax["tmax"] = o, dd, np.inf
    parts.append(ax)
    # degenerate: t_max 0, negative, NaN components, zero direction, huge / tiny directions
    n = 2_000
    dg = np.zeros(n, dtype=RAY)
    dg["o"] = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    dg["d"] = rng.normal(size=(n, 3)).astype(np.float32)
    dg["tmax"] = np.inf
    c = rng.integers(0, 6, n)
    dg["tmax"][c == 0] = 0.0
    dg["tmax"][c == 1] = -1.0
    dg["d"][c == 2] = 0.0
    dg["d"][c == 3] *= np.float32(1e18)
    dg["d"][c == 4] *= np.float32(1e-18)
    dg["tmax"][c == 5] = np.float32(1e-3)
    parts.append(dg)
    return np.concatenate(parts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_trace.log"))
    a = ap.parse_args()
    pm = importlib.import_module("pbrt-rs_amd")
    import oracle as ora
    ora.lib()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    log = open(a.out, "w")

    def say(s):
        print(s, flush=True)
        log.write(s + "\n"); log.flush()

    bad = 0
    totals = dict(rays=0, ties=0, sph_order=0, scenes=0)
    for it in range(a.iters):
        rng = np.random.default_rng([a.seed, it])
        env = {"PBRT_MI355_FLAT_MAX": str(rng.choice(["0", "64"])), "PBRT_MI355_UNIFIED_TREE": str(rng.choice(["0", "1"])),
               "PBRT_MI355_LBVH_LEAF": str(rng.choice(["0", "4"]))}
        os.environ.update(env)
        t0 = time.time()
        kinds, agg, sc, n_prims, sph, centres = make_scene(pm, ora, rng)
        rays = make_rays(pm, rng, centres)
        ref, _ = sc.intersect(rays)
        gpu = agg.intersect(rays)
        hit = ref["prim_id"] != MISS
        same_t = gpu["t"].view(np.uint32) == ref["t"].view(np.uint32)
        id_diff = gpu["prim_id"] != ref["prim_id"]
        ties = id_diff & same_t & hit & (gpu["prim_id"] != MISS)
        # two OVERLAPPING spheres hit within each other's t interval: the interval test rejects a sphere whose t interval reaches
        # above the current t_max, so the reference's own answer depends on the order its tree meets them in
        sph_ids = np.arange(sph.start, sph.stop, dtype=np.uint32)
        with np.errstate(invalid="ignore", divide="ignore"):
            sph_order = id_diff & ~ties & np.isin(gpu["prim_id"], sph_ids) & np.isin(ref["prim_id"], sph_ids) \
                & (np.abs(gpu["t"] - ref["t"]) <= 2e-5 * np.abs(ref["t"]))
        real = id_diff & ~ties & ~sph_order
        eq = ~id_diff & hit
        t_bad = eq & ~same_t
        is_sph = np.isin(ref["prim_id"], sph_ids) & eq
        flat = eq & ~is_sph
        uv_bad = flat & ((gpu["u"].view(np.uint32) != ref["u"].view(np.uint32)) | (gpu["v"].view(np.uint32) != ref["v"].view(np.uint32)))
        suv_bad = is_sph & ~(np.isclose(gpu["u"], ref["u"], rtol=1e-5, atol=2e-6) & np.isclose(gpu["v"], ref["v"], rtol=1e-5, atol=2e-6))
        # any hit: the rays themselves (t_max as given) and shadow rays off the hits
        occ_ref, _ = sc.intersect_p(rays)
        occ_bad = agg.intersect_p(rays).astype(np.uint8) != occ_ref
        sh = pm.scenes.shadow_rays(rays, ref, light=tuple(rng.uniform(-2, 2, 3)))
        sh_ref, _ = sc.intersect_p(sh)
        sh_bad = agg.intersect_p(sh).astype(np.uint8) != sh_ref
        # counters against the CPU walk of the exported tree
        agg.set_counting(True)
        agg.reset_counters()
        sub = np.ascontiguousarray(rays[:: 3])
        gc = agg.intersect(sub)
        c = agg.counters()
        cpu, cw = ora.wide_trace(agg.node_buffer(), agg.prim_buffer(), sub, flat=agg.info.flat_prims != 0)
        agg.set_counting(False)
        ctr_bad = (c["nodes_visited"], c["prims_tested"]) != (cw["nodes_visited"], cw["prims_tested"]) or c["fallback_walks"] != 0 \
            or not np.array_equal(gc["prim_id"], cpu["prim_id"]) or not np.array_equal(gc["t"].view(np.uint32), cpu["t"].view(np.uint32))
        n_bad = int(real.sum() + t_bad.sum() + uv_bad.sum() + suv_bad.sum() + occ_bad.sum() + sh_bad.sum()) + int(ctr_bad)
        totals["rays"] += len(rays); totals["ties"] += int(ties.sum()); totals["sph_order"] += int(sph_order.sum()); totals["scenes"] += 1
        say(f"it {it:3d} {kinds:20s} prims {n_prims:6d} flat {agg.info.flat_prims:3d} nodes {agg.info.n_nodes:6d} {env} hit {hit.mean():.2f} "
            f"ties {int(ties.sum())} sphere-order {int(sph_order.sum())} real {int(real.sum())} t {int(t_bad.sum())} uv {int(uv_bad.sum())} suv {int(suv_bad.sum())} "
            f"any {int(occ_bad.sum())} shadow {int(sh_bad.sum())} counters {'BAD' if ctr_bad else 'ok'}  [{time.time() - t0:.1f} s]")
        if n_bad:
            bad += 1
            for name, m in (("real", real), ("t", t_bad), ("uv", uv_bad), ("any", occ_bad)):
                for i in np.flatnonzero(m)[:3]:
                    say(f"    {name} ray {i}: o {rays['o'][i].tolist()} d {rays['d'][i].tolist()} tmax {rays['tmax'][i]} "
                        f"gpu ({gpu['prim_id'][i]}, {gpu['t'][i]!r}) ref ({ref['prim_id'][i]}, {ref['t'][i]!r})")
            if ctr_bad:
                say(f"    counters gpu {c} cpu {cw}")
        del agg, sc
    say(f"{totals['scenes']} scenes, {totals['rays']} rays, {totals['ties']} exact ties, {totals['sph_order']} overlapping-sphere order cases, {bad} scenes with mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
