"""Runs ON the GPU box: how the duration of the SAME carve launch depends on what the GPU did just before.
The benchmark scene on the whole grid and on one z-slab of an 8-GPU run (128 slices), M steps
 (a) queued back to back without any host synchronisation (the event log of "carvetimer" is read afterwards),
 (b) with a device synchronisation after every step (the GPU idles while the host comes round), and
 (c) as (b) with an extra host pause per step.
Prints the carve kernel's time per step."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vacancy_amd import carver as vc, synth  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402

n, nv = 1024, 32
views, masks = synth.sphere_views(n, nv, 1280, 720)
opt = synth.sphere_option(n, UpdateOption())
sdf0 = vc.make_sdf(masks[0])


def series(zr, steps, mode, pause=0.0):
    c = vc.VoxelCarver(opt, device_id=0, z_range=zr)
    assert c.Init()
    d = c.upload_sdf(sdf0)
    batch = vc.VoxelCarver.prepare_batch(views, [d] * nv)
    c.reset(); c.CarveBatchDevice(batch); c.sync()   # allocations, uploads
    time.sleep(0.2)                                   # a cold start: the GPU has idled
    c.set_param("carvetimer", 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        c.reset()
        c.CarveBatchDevice(batch)
        if mode != "a":
            c.sync()
            if pause:
                time.sleep(pause)
    c.sync()
    wall = (time.perf_counter() - t0) * 1e3
    log = c.carve_log()
    c.free_device(d); c.close()
    return wall, log


def show(name, wall, log):
    k = [r[2] for r in log]
    p = [r[1] for r in log]
    span = log[-1][0] + p[-1] + k[-1]
    print("%s: wall %.2f ms, device span %.2f ms, %d steps" % (name, wall, span, len(log)))
    print("   kernel ms: first 10 %s" % [round(x, 3) for x in k[:10]])
    print("              every 10th %s" % [round(x, 3) for x in k[::10]])
    print("   mean of the last 10: kernel %.3f, pre-pass %.3f; step period %.3f"
          % (sum(k[-10:]) / 10, sum(p[-10:]) / 10, (log[-1][0] - log[-11][0]) / 10))


for zr, steps in (((448, 576), 200), ((0, 1024), 60)):
    print("== z-range", zr)
    show("(a) back to back", *series(zr, steps, "a"))
    show("(b) sync after every step", *series(zr, steps, "b"))
    show("(c) sync + 1 ms host pause", *series(zr, steps, "c", 0.001))
