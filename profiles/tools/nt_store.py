"""Runs ON the GPU box: single-view launches with the cooperative write-back as ordinary and as streaming stores
("ntstore" 0 / 1): the first view on a fresh grid and the views after it, both modes, the workgroup-per-block kernel
("rowkernel" 0).  Alternating on one box.   usage: python profiles/tools/nt_store.py [n]"""
import sys
sys.path.insert(0, ".")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for mode in ("tsdf", "default"):
    uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
    views, masks = synth.sphere_views(n, 32, 1280, 720)
    c = vc.VoxelCarver(synth.sphere_option(n, uo)); assert c.Init()
    d = c.upload_sdf(vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band))
    c.set_param("defer", 0)
    c.set_param("rowkernel", 0)
    for nt in (0, 1, 0, 1, 0, 1):
        c.set_param("ntstore", nt)
        first, rest = [], []
        for rep in range(3):
            c.reset(); c.sync()
            c.set_param("carvetimer", 1)
            for i in range(8):
                assert c.CarveDevice(views[i], d)
            c.sync()
            log = c.carve_log()
            first.append(log[0][2])
            rest.append(sum(r[2] for r in log[1:]) / 7.0)
        print("%-7s ntstore %d: first view kernel ms %s | later views avg %s" % (mode, nt, " ".join("%.3f" % x for x in first), " ".join("%.3f" % x for x in rest)))
    c.close()
