"""Runs ON the GPU box: single-view launches over the bench scene ("defer" 0) with the carve kernel launched over the
listed workgroups only ("livesync" 1: the host waits for the list's length) and over every workgroup (0); device ms per
view from the event log, and the wall time of the 32-view loop (the wait costs host time, the empty workgroups device time).
usage: python profiles/tools/per_view_livesync.py [n] [mode]"""
import sys, time
sys.path.insert(0, ".")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else "default"
nv = 32
uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
opt = synth.sphere_option(n, uo)
views, masks = synth.sphere_views(n, nv, 1280, 720)
c = vc.VoxelCarver(opt); assert c.Init()
d = c.upload_sdf(vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band))
c.set_param("defer", 0)
for ls in (1, 0, 1, 0):
    c.set_param("livesync", ls)
    c.reset(); c.sync()
    c.set_param("carvetimer", 1)
    t = time.perf_counter()
    for i in range(nv):
        assert c.CarveDevice(views[i], d)
    c.sync()
    wall = (time.perf_counter() - t) * 1e3
    log = c.carve_log()
    dev = [r[1] + r[2] for r in log]
    print("%s livesync %d: device ms per view after the first %.3f (pre-pass %.3f + kernel %.3f); loop wall %.2f ms = %.3f per view"
          % (mode, ls, sum(dev[1:]) / (nv - 1), sum(r[1] for r in log[1:]) / (nv - 1), sum(r[2] for r in log[1:]) / (nv - 1), wall, wall / nv))
