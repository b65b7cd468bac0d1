#!/usr/bin/env python3
"""Write the sRGB RGB->spectrum coefficient table in the text format the reference loads from
src/util/rgbspectrum_srgb.txt (util/data.rs:1684-1702): line 1 the resolution `64`, line 2 a JSON array of the 64 z nodes,
then a JSON array of 3*64^3*3 coefficients in CoefficientTable order.  The file is missing from the reference snapshot
(the binary panics without it); with this one a `cargo run` of the reference becomes possible on a machine that has Rust.

usage: python scripts/gen_rgbspectrum_table.py out/rgbspectrum_srgb.txt
"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pm = importlib.import_module("pbrt-rs_amd")
z, co = pm.host.Spectra().rgb_table()
with open(sys.argv[1], "w") as f:
    f.write("64\n")
    f.write(json.dumps([float(v) for v in z]) + "\n")
    f.write(json.dumps([float(v) for v in co]))
print("wrote", sys.argv[1], os.path.getsize(sys.argv[1]), "bytes")
