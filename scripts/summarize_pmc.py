#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: sum of every counter over the dispatches of a kernel,
dispatch count, and derived HBM-side bytes (FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B per the rocprofv3
definition: FETCH_SIZE = TCC_EA0_RDREQ*64/1024; on gfx950 a wide coalesced read is tallied at half its bytes, see
MI355X_MICROARCH.md §HBM — the summary prints the raw value and the doubled one)."""
import csv
import collections
import json
import sys


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "").strip()


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for p in paths:
        with open(p) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[(k, p)].add(row["Dispatch_Id"])
    out = {}
    for k, ctrs in agg.items():
        if k.startswith("__amd") or k.startswith("at::"):
            continue
        n = max(len(v) for (kk, _), v in calls.items() if kk == k)
        d = {"dispatches": n}
        d.update({c: v for c, v in sorted(ctrs.items())})
        if "FETCH_SIZE" in ctrs:
            d["fetch_bytes_raw"] = ctrs["FETCH_SIZE"] * 1024
            d["fetch_bytes_x2_wide_loads"] = ctrs["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in ctrs:
            d["write_bytes"] = ctrs["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in ctrs and "TCC_MISS_sum" in ctrs:
            d["l2_hit_rate"] = ctrs["TCC_HIT_sum"] / max(1.0, ctrs["TCC_HIT_sum"] + ctrs["TCC_MISS_sum"])
        if "SQ_WAVE_CYCLES" in ctrs:
            wc = ctrs["SQ_WAVE_CYCLES"]
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if c in ctrs:
                    d[c + "/WAVE_CYCLES"] = ctrs[c] / max(1.0, wc)
        out[k] = d
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:])
