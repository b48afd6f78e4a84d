"""Runs ON the GPU box: the FIRST single-view launch on a fresh grid (every voxel stored once, nothing read) with each wave
storing its own 16-byte pieces ("coopstore" 0) and with the cooperative write-back (-1: the library's rule takes it
for a single-view launch on a fresh slab).  usage: python profiles/tools/first_view.py [n] [mode]"""
import sys
sys.path.insert(0, ".")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else "default"
uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
opt = synth.sphere_option(n, uo)
views, masks = synth.sphere_views(n, 32, 1280, 720)
c = vc.VoxelCarver(opt); assert c.Init()
d = c.upload_sdf(vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band))
c.set_param("defer", 0)
for coop in (0, -1, 0, -1):
    c.set_param("coopstore", coop)
    ms = []
    for rep in range(4):
        c.reset(); c.sync()
        c.set_param("carvetimer", 1)
        assert c.CarveDevice(views[rep], d)
        log = c.carve_log()
        ms.append(log[-1][2])
    print("%s coopstore %d: first view on a fresh grid, carve kernel ms: %s" % (mode, coop, " ".join("%.3f" % x for x in ms)))
