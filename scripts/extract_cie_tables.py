"""One-off: extract the CIE 1931 2-degree observer (X,Y,Z at 1 nm, 360..830 nm) and the CIE D65 illuminant
(5 nm, 300..830 nm) NUMERIC tables from the reference's data file into a small binary fixture.

These are published CIE standard data (every spectral renderer carries the same numbers); only the numbers
are kept, as float64 parsed from the decimal literals.  Run here (the reference tree is not on the GPU box).
"""
import re
import sys
import numpy as np

src = open("/root/reference/src/util/data.rs").read()


def table(name):
    m = re.search(r"pub const %s[^=]*=\s*\[(.*?)\n\];" % name, src, re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return np.array([float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body)], dtype=np.float64)


x, y, z, lam, d65 = (table(n) for n in ("CIE_X", "CIE_Y", "CIE_Z", "CIE_LAMBDA", "CIE_ILLUM_D6500"))
assert x.size == y.size == z.size == lam.size == 471 and d65.size == 214, (x.size, lam.size, d65.size)
assert lam[0] == 360 and lam[-1] == 830
np.savez(sys.argv[1], cie_x=x, cie_y=y, cie_z=z, cie_lambda=lam, d65_interleaved=d65)
print("ok", x[:3], d65[:6])

# the same numbers as raw little-endian float32 for the C++ host mirror (include/pbrt_mi355.hpp Spectra):
# "PMCIE1\0\0", cie_x[471], cie_y[471], cie_z[471], uint32 n, d65_interleaved[n]
import struct
with open(sys.argv[1].replace(".npz", ".bin"), "wb") as f:
    f.write(b"PMCIE1\0\0")
    for a in (x, y, z):
        f.write(a.astype(np.float32).tobytes())
    f.write(struct.pack("<I", d65.size))
    f.write(d65.astype(np.float32).tobytes())
