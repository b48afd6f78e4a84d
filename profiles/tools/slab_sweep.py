"""Runs ON the GPU box: what a z-slab launch of the benchmark scene costs beyond its share of the whole grid.
 (1) slabs of growing thickness around the grid centre: kernel time against thickness -> the fixed cost of a launch;
 (2) the eight slabs of an 8-GPU run back to back on ONE stream (no idle gaps: does an idle GPU cost the next launch?)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vacancy_amd import carver as vc, synth, dist as vdist  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402

n, nv = 1024, 32
views, masks = synth.sphere_views(n, nv, 1280, 720)
opt = synth.sphere_option(n, UpdateOption())
sdf0 = vc.make_sdf(masks[0])


def carver(z0, z1):
    c = vc.VoxelCarver(opt, device_id=0, z_range=(z0, z1))
    assert c.Init()
    c.set_param("carvetimer", 1)
    return c


print("(1) thickness sweep around z = 512")
for half in (32, 64, 128, 256, 512):
    c = carver(512 - half, 512 + half)
    d = c.upload_sdf(sdf0)
    batch = vc.VoxelCarver.prepare_batch(views, [d] * nv)
    best = None
    for it in range(5):
        c.reset(); c.sync(); c.timer_begin(); c.CarveBatchDevice(batch); ms = c.timer_end()
        pre, ker = c.last_carve_ms()
        if it and (best is None or ms < best[0]):
            best = (ms, pre, ker)
    print("  %4d slices: step %.3f  pre-pass %.3f  kernel %.3f  other %.3f" % (2 * half, best[0], best[1], best[2], best[0] - best[1] - best[2]))
    c.free_device(d); c.close()

print("(2) eight slabs back to back on one stream")
cs = [carver(*vdist.slab_range(n, s, 8)) for s in range(8)]
for c in cs[1:]:
    c.use_stream_of(cs[0])
d = cs[0].upload_sdf(sdf0)
batch = vc.VoxelCarver.prepare_batch(views, [d] * nv)
for it in range(4):
    for c in cs:
        c.reset()
    cs[0].sync()
    t0 = time.perf_counter()
    cs[0].timer_begin()
    for c in cs:
        c.CarveBatchDevice(batch)
    ms = cs[0].timer_end()
    wall = (time.perf_counter() - t0) * 1e3
    parts = [c.last_carve_ms() for c in cs]
    print("  all eight: %.3f ms (wall %.3f); kernels %s sum %.3f; pre-pass sum %.3f"
          % (ms, wall, [round(p[1], 3) for p in parts], sum(p[1] for p in parts), sum(p[0] for p in parts)))
cs[0].free_device(d)
for c in reversed(cs):
    c.close()
