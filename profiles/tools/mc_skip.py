"""Runs ON the GPU box: marching cubes of the bench scene with and without the brick minima ("mcskip" 1 / 0),
alternating in one process; kernel ms (HIP events) of every call, mesh hash, and per-kernel times when run under
rocprofv3.   usage: python profiles/tools/mc_skip.py [n]"""
import hashlib
import sys
sys.path.insert(0, ".")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nv = 32
opt = synth.sphere_option(n, UpdateOption())
views, masks = synth.sphere_views(n, nv, 1280, 720)
sdf0 = vc.make_sdf(masks[0])
c = vc.VoxelCarver(opt)
assert c.Init()
d = [c.upload_sdf(sdf0)] * nv
assert c.CarveBatchDevice(vc.VoxelCarver.prepare_batch(views, d))
c.set_param("meshkeys", 0)
for skip in (1, 0, 1, 0):
    c.set_param("mcskip", skip)
    c.ExtractIsoSurface(0.0, True)
    ms = []
    for it in range(5):
        m = c.ExtractIsoSurface(0.0, True)
        ms.append(m["device_ms"])
    h = hashlib.sha1(m["vertices"].tobytes() + m["faces"].tobytes()).hexdigest()[:12]
    cells = float(n - 1) ** 3
    med = sorted(ms)[len(ms) // 2]
    print("mcskip %d: median %.3f ms (%s)  %.0f Mcells/s  frac of 8 TB/s (4 B/cell) %.3f  mesh %s %d verts"
          % (skip, med, " ".join("%.3f" % x for x in ms), cells / med / 1e3, cells * 4 / (med * 1e-3) / 8e12, h, len(m["vertices"])))
