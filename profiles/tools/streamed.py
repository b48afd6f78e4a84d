"""Runs ON the GPU box: the streamed entry point (vcy_carve_batch_silhouettes: host silhouettes -> staging -> DMA -> device
SDF -> fused carve), wall time and the two sides' device times.  usage: streamed.py [n] [views] [w] [h]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vacancy_amd import carver as vc, synth  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
h = int(sys.argv[4]) if len(sys.argv) > 4 else 720
views, masks = synth.sphere_views(n, nv, w, h)
c = vc.VoxelCarver(synth.sphere_option(n, UpdateOption()))
assert c.Init()
rows = []
for rep in range(8):
    c.reset(); c.sync()
    t = time.perf_counter()
    assert c.CarveBatchSilhouettes(views, masks)
    rows.append(((time.perf_counter() - t) * 1e3,) + c.last_stream_ms())
for r in rows:
    print("wall %.3f ms: producer %.3f, carve %.3f (library wall %.3f) overlap %.3f" % (r[0], r[1], r[2], r[3], max(r[1], r[2]) / r[3]))
