"""Runs ON the GPU box: marching cubes with the one-sweep cell search and with the bit planes in memory
("mcsweep" 1 / 0) on sphere scenes of several grid sizes; kernel ms (HIP events), best of 4 after a warm-up call.
usage: python profiles/tools/mc_sizes.py [n ...]      (default 256 512 1024 2048)"""
import sys
sys.path.insert(0, ".")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption

sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048]
for n in sizes:
    nv = 8
    opt = synth.sphere_option(n, UpdateOption())
    views, masks = synth.sphere_views(n, nv, 1280, 720)
    c = vc.VoxelCarver(opt)
    assert c.Init(), vc.last_error()
    d = [c.upload_sdf(vc.make_sdf(masks[0]))] * nv
    assert c.CarveBatchDevice(vc.VoxelCarver.prepare_batch(views, d))
    c.set_param("meshkeys", 0)
    row = []
    for sweep in (1, 0, 1, 0):
        c.set_param("mcsweep", sweep)
        c.ExtractIsoSurface(0.0, True)
        best = min(c.ExtractIsoSurface(0.0, True)["device_ms"] for _ in range(4))
        row.append("sweep %d: %.3f ms" % (sweep, best))
    cells = float(n - 1) ** 3
    print("%5d^3  %s   (%.0f Mcells)" % (n, "   ".join(row), cells / 1e6))
    c.close()
