#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel.

usage: summarize_pmc.py <pass1.csv> <pass2.csv> ...      (one CSV per --pmc pass of the SAME command)

Per kernel: dispatches per pass, every counter as its PER-DISPATCH mean (a counter collected in several passes is averaged
over them), and derived figures:
  fetch_bytes_x2_wide_loads / write_bytes   HBM-side bytes per dispatch.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 a wide
                                            coalesced read is tallied at half its bytes (MI355X_MICROARCH.md, HBM), so the fetch
                                            side is doubled
  l2_hit_rate                               TCC_HIT / (TCC_HIT + TCC_MISS)
  valu_issue_frac                           SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): a wave64 VALU
                                            instruction occupies its 32-lane-per-cycle SIMD for 2 cycles
  issue_frac_with_scalar                    (SQ_INSTS_VALU x 2 + SQ_INSTS_SALU) / the same cycles: the scalar instructions of the
                                            exec-mask bookkeeping counted at one issue cycle each (the flat sweep responded
                                            linearly to instructions removed at 0.8 of this figure)
  active_lane_frac                          SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64 (calibrated: kernels that run full
                                            waves - raygen, accumulate - read 1.00)
  wait_frac                                 SQ_WAIT_ANY / SQ_WAVE_CYCLES
"""
import collections
import csv
import json
import sys

N_SIMD, N_XCD = 1024, 8


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main(paths):
    sums = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))   # kernel -> counter -> file -> sum
    disp = collections.defaultdict(lambda: collections.defaultdict(set))                                      # kernel -> file -> dispatch ids
    for p in paths:
        with open(p) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                sums[k][row["Counter_Name"]][p] += float(row["Counter_Value"])
                disp[k][p].add(row["Dispatch_Id"])
    out = {}
    for k, ctrs in sums.items():
        if k.startswith("__amd") or k.startswith("at::"):
            continue
        d = {"dispatches": max(len(v) for v in disp[k].values())}
        m = {}
        for c, per_file in sorted(ctrs.items()):
            m[c] = sum(v / len(disp[k][p]) for p, v in per_file.items()) / len(per_file)
        d.update(m)
        if "FETCH_SIZE" in m:
            d["fetch_bytes_raw"] = m["FETCH_SIZE"] * 1024
            d["fetch_bytes_x2_wide_loads"] = m["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in m:
            d["write_bytes"] = m["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
            d["l2_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        if "SQ_INSTS_VALU" in m and m.get("GRBM_GUI_ACTIVE", 0) > 0:
            d["valu_issue_frac"] = m["SQ_INSTS_VALU"] * 2.0 / (N_SIMD * m["GRBM_GUI_ACTIVE"] / N_XCD)
        if "SQ_INSTS_VALU" in m and "SQ_INSTS_SALU" in m and m.get("GRBM_GUI_ACTIVE", 0) > 0:
            # scalar instructions counted at one issue cycle each beside the vector ones
            d["issue_frac_with_scalar"] = (m["SQ_INSTS_VALU"] * 2.0 + m["SQ_INSTS_SALU"]) / (N_SIMD * m["GRBM_GUI_ACTIVE"] / N_XCD)
            d["scalar_per_vector_inst"] = m["SQ_INSTS_SALU"] / max(1.0, m["SQ_INSTS_VALU"])
        if m.get("SQ_ACTIVE_INST_VALU", 0) > 0 and "SQ_THREAD_CYCLES_VALU" in m:
            d["active_lane_frac"] = m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"] / 64.0
        if m.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_WAIT_ANY" in m:
            d["wait_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
        out[k] = d
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:])
