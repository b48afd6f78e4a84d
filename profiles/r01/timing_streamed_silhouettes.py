import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from vacancy_amd import carver as vc, synth
from vacancy_amd.capi import UpdateOption
n, nv = 1024, 32
views, masks = synth.sphere_views(n, nv, 1280, 720)
opt = synth.sphere_option(n, UpdateOption())
c = vc.VoxelCarver(opt); assert c.Init()
for it in range(3):
    c.reset(); c.sync(); t0 = time.perf_counter()
    assert c.CarveBatchSilhouettes(views, masks)
    c.sync(); print("batch silhouettes wall ms", round((time.perf_counter() - t0) * 1e3, 2))
