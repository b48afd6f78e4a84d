"""Writes tests/golden/mc_tables.bin from the REFERENCE's own marching-cubes tables: builds
oracle/_ref (oracle/ref.mk compiles /root/reference/src/vacancy/marching_cubes_lut.cc unmodified)
and runs its dumper.  int32 kEdgeTable[256] followed by int32 kTriTable[256][16], little endian.
Run in the build container (the reference is not on the GPU box):
    python tests/golden/make_mc_tables.py"""
import os
import subprocess

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
subprocess.run(["make", "-C", os.path.join(root, "oracle"), "-f", "ref.mk"], check=True)
out = os.path.join(here, "mc_tables.bin")
subprocess.run([os.path.join(root, "oracle", "_ref", "ref_lut_dump"), out], check=True)
t = np.fromfile(out, "<i4")
assert t.shape == (256 + 256 * 16,)
print("wrote", out, "edge sum", int(t[:256].sum()), "triangles", int((t[256:] >= 0).sum()) // 3)
