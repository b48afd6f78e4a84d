import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from vacancy_amd import carver as vc, synth
from vacancy_amd.capi import UpdateOption
n, nv = 1024, 32
views, masks = synth.sphere_views(n, nv, 1280, 720)
opt = synth.sphere_option(n, UpdateOption())
sdf0 = vc.make_sdf(masks[0])
c = vc.VoxelCarver(opt); assert c.Init()
d = [c.upload_sdf(sdf0) for _ in range(nv)]
for defer in (1, 0):
    c.set_param("defer", defer)
    for name, fn in (("vcy_carve_device x32", lambda i: c.CarveDevice(views[i], d[i])), ("vcy_carve (host SDF) x32", lambda i: c.Carve(views[i], sdf0)),
                     ("vcy_carve_silhouette x32", lambda i: c.CarveSilhouette(views[i], masks[i]))):
        best = 1e9
        for it in range(3):
            c.reset(); c.sync(); t0 = time.perf_counter()
            for i in range(nv):
                assert fn(i)
            c.sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
        print("defer=%d %-28s %.2f ms wall = %.0f Mvoxel*views/s" % (defer, name, best, n ** 3 * nv / best / 1e3))
