"""Runs ON the GPU box: single-view launches with their footprints from the pre-pass's records ("prologue" 0) and computed
in the carve kernel's own prologue ("prologue" 1): the first view on a fresh 1024^3 grid (default and weighted average),
and one view per launch in the weighted-average mode over the carved grid.  usage: python profiles/tools/ab_prologue_single.py"""
import hashlib, sys
sys.path.insert(0, ".")
import numpy as np
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n = 1024
views, masks = synth.sphere_views(n, 32, 1280, 720)
ids = np.arange(0, n ** 3, n ** 3 // 200003, dtype=np.int64)
for mode in ("default", "tsdf"):
    uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
    c = vc.VoxelCarver(synth.sphere_option(n, uo)); assert c.Init()
    d = c.upload_sdf(vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band))
    c.set_param("defer", 0)
    for prologue in (0, 1, 0, 1):
        c.set_param("prologue", prologue)
        first = []
        for rep in range(4):
            c.reset(); c.sync(); c.set_param("carvetimer", 1)
            assert c.CarveDevice(views[rep], d)
            log = c.carve_log(); first.append((log[-1][1], log[-1][2]))
        c.reset(); c.sync(); c.set_param("carvetimer", 1)
        for i in range(12):
            assert c.CarveDevice(views[i], d)
        log = c.carve_log()
        s, u = c.download_voxels(ids)
        h = hashlib.sha1(s.tobytes() + u.tobytes()).hexdigest()[:10]
        print("%-7s prologue %d: first view on a fresh grid pre-pass + kernel ms %s | views 2..12 over the carved grid: pre-pass %.3f kernel %.3f ms per view | state %s"
              % (mode, prologue, " ".join("%.3f+%.3f" % f for f in first), sum(r[1] for r in log[1:]) / 11, sum(r[2] for r in log[1:]) / 11, h), flush=True)
    c.close()
