"""Runs ON the GPU box: the state prefetch into the L2 ("l2ahead": distance in units of 8 workgroups, 0 = off) for the
launches it is made for -- one view per launch in the weighted-average mode over a carved 1024^3 grid -- and for the second
of two fused launches of 16 views in that mode.  Prints ms per view / per launch and a hash of the state (must not change).
usage: python profiles/tools/ab_l2ahead.py [n] [distances ...]"""
import hashlib, sys
sys.path.insert(0, ".")
import numpy as np
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dists = [int(x) for x in sys.argv[2:]] or [0, 16, 32, 64, 128, 255, 0, 64]
uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1)
opt = synth.sphere_option(n, uo)
views, masks = synth.sphere_views(n, 32, 1280, 720)
c = vc.VoxelCarver(opt); assert c.Init()
d = c.upload_sdf(vc.make_sdf(masks[0], use_truncation=True, band=0.1))
ids = np.arange(0, n ** 3, max(1, n ** 3 // 200003), dtype=np.int64)
for ahead in dists:
    c.set_param("l2ahead", ahead)
    c.set_param("defer", 0)
    for rep in range(2):
        c.reset(); c.sync()
        c.set_param("carvetimer", 1)
        for i in range(16):
            assert c.CarveDevice(views[i], d)
        log = c.carve_log()
    ker = [r[2] for r in log]
    s, u = c.download_voxels(ids)
    h = hashlib.sha1(s.tobytes() + u.tobytes()).hexdigest()[:10]
    # two fused launches of 16 views: the second reads the carved state
    c.set_param("defer", 1)
    ba = vc.VoxelCarver.prepare_batch(views[:16], [d] * 16); bb = vc.VoxelCarver.prepare_batch(views[16:], [d] * 16)
    for rep in range(2):
        c.reset(); c.sync(); c.set_param("carvetimer", 1)
        assert c.CarveBatchDevice(ba) and c.CarveBatchDevice(bb)
        log2 = c.carve_log()
    print("l2ahead %3d: single-view launches, kernel ms per view after the first: mean %.3f min %.3f max %.3f (first %.3f)  state %s | "
          "fused 16 + 16: %.3f + %.3f ms" % (ahead, sum(ker[1:]) / 15, min(ker[1:]), max(ker[1:]), ker[0], h, log2[0][2], log2[1][2]), flush=True)
