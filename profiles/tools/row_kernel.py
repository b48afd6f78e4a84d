"""Runs ON the GPU box: the reference's call pattern on the bench scene -- one view per launch ("defer" 0) -- with the
few-view flavour of the fused kernel ("rowkernel" -1: a wave walks the bricks of a row segment) against the
workgroup-per-block kernel ("rowkernel" 0); carve ms per view from the library's event log (pre-pass + kernel), the
whole-grid state hash of both after all views.
usage: [PARAM=rowkernel VALUES=0,-1] python profiles/tools/row_kernel.py [n] [mode ...]
(any other knob of vcy_set_param can be compared the same way, e.g. PARAM=oneview VALUES=0,1)"""
import os
import sys
sys.path.insert(0, ".")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
modes = sys.argv[2:] or ["tsdf", "default"]
nv = 32
for mode in modes:
    uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, 1280, 720)
    sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
    cs = []
    PARAM = os.environ.get("PARAM", "rowkernel")
    for rk in [int(x) for x in os.environ.get("VALUES", "0,-1").split(",")]:
        c = vc.VoxelCarver(opt)
        assert c.Init()
        c.set_param("defer", 0)
        c.set_param(PARAM, rk)
        cs.append((rk, c, c.upload_sdf(sdf0)))
    for rep in range(3):
        for rk, c, d in cs:
            c.reset()
            c.set_param("carvetimer", 1)
            for i in range(nv):
                assert c.CarveDevice(views[i], d)
            c.sync()
            log = c.carve_log()
            pre = [r[1] for r in log]
            ker = [r[2] for r in log]
            tot = sum(pre) + sum(ker)
            print(("%-7s " + PARAM + " %2d rep %d: total %.2f ms | first view %.3f (kernel %.3f) | others avg %.3f (kernel %.3f, min %.3f max %.3f) -> %.0f Mvoxel*views/s")
                  % (mode, rk, rep, tot, pre[0] + ker[0], ker[0], (tot - pre[0] - ker[0]) / (nv - 1), sum(ker[1:]) / (nv - 1),
                     min(ker[1:]), max(ker[1:]), float(n) ** 3 * nv / tot / 1e3))
    if len(cs) > 1:
        print("%-7s voxels differing between the two kernels after %d views: %d" % (mode, nv, cs[0][1].state_diff(cs[1][1])))
    for _, c, d in cs:
        c.free_device(d)
        c.close()
