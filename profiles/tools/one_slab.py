"""Runs ON the GPU box: the benchmark scene carved on ONE z-slab of 64 slices (a central and an outer one, what a rank of
an 8-GPU run with 2 slabs per GPU holds): step time, pre-pass / kernel split; under rocprofv3 the per-kernel timeline."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vacancy_amd import carver as vc, synth
from vacancy_amd.capi import UpdateOption
n, nv = 1024, 32
views, masks = synth.sphere_views(n, nv, 1280, 720)
opt = synth.sphere_option(n, UpdateOption())
sdf0 = vc.make_sdf(masks[0])
for zr in ((448, 512), (0, 64)):
    c = vc.VoxelCarver(opt, device_id=0, z_range=zr); assert c.Init()
    d = [c.upload_sdf(sdf0)] * nv
    batch = vc.VoxelCarver.prepare_batch(views, d)
    c.set_param("carvetimer", 1)
    for it in range(4):
        c.reset(); c.timer_begin(); c.CarveBatchDevice(batch); ms = c.timer_end()
        print(zr, "step %.3f ms, prepass/kernel %s" % (ms, c.last_carve_ms()))
    c.close()
