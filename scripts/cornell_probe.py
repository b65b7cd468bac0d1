"""Dev probe (GPU): per-kernel time of the C2 Cornell render with / without the glass sphere, and leaf sizes."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pm = importlib.import_module("pbrt-rs_amd")

spp = int(os.environ.get("SPP", "16"))
for glass in (True, False):
    for leaf in (4, 1):
        sd = pm.scenes.cornell_scene((1024, 1024), glass_sphere=glass)
        integ = pm.WavefrontIntegrator(5, True, True, sd, spp, seed=0, max_prims_in_node=leaf)
        integ.aggregate.set_counting(True)
        integ.clear(); st_c = integ.render()
        integ.aggregate.set_counting(False)
        integ.clear(); integ.render()
        integ.clear(); st = integ.render()
        ms = [round(x, 2) for x in st["ms_by_kernel"]]
        print(f"glass={glass} leaf={leaf} nodes={integ.aggregate.info.n_nodes} prims={integ.aggregate.info.n_prims} "
              f"rays={st['closest_rays']/1e6:.1f}M/{st['shadow_rays']/1e6:.1f}M ms={ms} total={st['kernel_ms']:.1f} closest n/p={st_c['closest_nodes']/st_c['closest_rays']:.2f}/"
              f"{st_c['closest_prims']/st_c['closest_rays']:.2f} shadow n/p={st_c['shadow_nodes']/max(1,st_c['shadow_rays']):.2f}/"
              f"{st_c['shadow_prims']/max(1,st_c['shadow_rays']):.2f} exact/ray={st_c['exact_tests']/(st_c['closest_rays']+st_c['shadow_rays']):.2f}",
              flush=True)
        del integ
