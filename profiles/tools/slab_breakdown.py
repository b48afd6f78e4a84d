"""Runs ON the GPU box: where the time of a z-slab launch goes.  The benchmark scene (1024^3 x 32 views at 1280x720),
carved slab by slab as a rank of a G-GPU run would: for every slab the event time of the step, its pre-pass and carve
kernel (vcy_last_carve_ms), the host time of the launch call itself (everything is asynchronous: that call's duration
is the host-side preparation) and the wall time to the sync.  usage: slab_breakdown.py [G] [k] [bounds ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vacancy_amd import carver as vc, synth, dist as vdist  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402

n, nv = 1024, 32
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bounds = [int(x) for x in sys.argv[3:]]
views, masks = synth.sphere_views(n, nv, 1280, 720)
opt = synth.sphere_option(n, UpdateOption())
sdf0 = vc.make_sdf(masks[0])
if bounds:
    slabs = list(zip(bounds[:-1], bounds[1:]))
else:
    slabs = [vdist.slab_range(n, s, G * k) for s in range(G * k)]
tot = 0.0
for z0, z1 in slabs:
    c = vc.VoxelCarver(opt, device_id=0, z_range=(z0, z1))
    assert c.Init()
    d = [c.upload_sdf(sdf0)] * nv
    batch = vc.VoxelCarver.prepare_batch(views, d)
    c.set_param("carvetimer", 1)
    rows = []
    for it in range(5):
        c.reset()
        c.sync()
        t0 = time.perf_counter()
        c.timer_begin()
        c.CarveBatchDevice(batch)
        t1 = time.perf_counter()
        ms = c.timer_end()
        t2 = time.perf_counter()
        pre, ker = c.last_carve_ms()
        rows.append((ms, pre, ker, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
    best = min(rows[1:])
    tot += best[0]
    print("z [%4d, %4d): step %.3f ms = pre-pass %.3f + kernel %.3f + other %.3f | host call %.3f ms, wall %.3f ms"
          % (z0, z1, best[0], best[1], best[2], best[0] - best[1] - best[2], best[3], best[4]))
    c.free_device(d[0])
    c.close()
print("sum of slabs %.3f ms" % tot)
