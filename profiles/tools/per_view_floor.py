"""Runs ON the GPU box: what a single-view launch costs when NOTHING is live -- after a few views, launches with an image
whose every pixel lies below anything the grid holds (kMax: every pair is dropped, ub <= brick minimum).  With the
live-workgroup list the carve kernel is 512 K workgroups that read list[0] and leave; without it every wave reads its
record and its brick's minimum first.
usage: python profiles/tools/per_view_floor.py [n]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
opt = synth.sphere_option(n, UpdateOption())
views, masks = synth.sphere_views(n, 32, 1280, 720)
c = vc.VoxelCarver(opt); assert c.Init()
sdf = vc.make_sdf(masks[0])
d = c.upload_sdf(sdf)
low = c.upload_sdf(np.full_like(sdf, -1.0e30))
c.set_param("defer", 0)
for ll in (1, 0, 1):
    c.set_param("livelist", ll)
    c.reset()
    for i in range(6):
        assert c.CarveDevice(views[i], d)
    c.set_param("carvetimer", 1)
    for i in range(6, 14):
        assert c.CarveDevice(views[i], low)
    log = c.carve_log()[2:]
    print("livelist %d: a launch in which no pair is live: pre-pass %.3f ms (window maxima, records, list) + carve kernel %.3f ms"
          % (ll, sum(r[1] for r in log) / len(log), sum(r[2] for r in log) / len(log)))
