"""Carve rate of the benchmark scene (1024^3 x 32 views at 1280x720) in option combinations bench.py does not
cover: nearest-neighbour sampling, update_outside = kMax, an update limit in reach, general weights."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vacancy_amd import synth  # noqa: E402
from vacancy_amd import carver as vc  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402

n, nv, w, h = 1024, 32, 1280, 720
for name, kw in (("default", {}), ("nearest neighbour", dict(sdf_interp=0)), ("update_outside max", dict(update_outside=1)),
                 ("max_update 20 (limit in reach)", dict(voxel_max_update_num=20)),
                 ("average, weight 0.5", dict(voxel_update=1, voxel_update_weight=0.5)),
                 ("average nn + truncation", dict(voxel_update=1, sdf_interp=0, use_truncation=True))):
    uo = UpdateOption(**kw)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
    c = vc.VoxelCarver(opt)
    assert c.Init(), vc.last_error()
    batch = vc.VoxelCarver.prepare_batch(views, [c.upload_sdf(sdf0)] * nv)
    best = 1e9
    for it in range(4):
        c.reset()
        c.timer_begin()
        assert c.CarveBatchDevice(batch), vc.last_error()
        best = min(best, c.timer_end())
    print("%-34s %7.2f ms  %9.0f Mvoxel*views/s" % (name, best, n ** 3 * nv / best / 1e3))
    del c
