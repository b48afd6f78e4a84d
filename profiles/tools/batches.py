"""Runs ON the GPU box: the 32 views of the bench scene carved in B batches of 32 / B views (fused launches over a grid
the earlier batches have carved), with the cooperative write-back forced on / off / by the library's rule.
usage: python profiles/tools/batches.py [n] [mode]"""
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
sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
c = vc.VoxelCarver(opt); assert c.Init()
d = c.upload_sdf(sdf0)
for B in (2, 4, 8):
    per = nv // B
    batches = [vc.VoxelCarver.prepare_batch(views[i * per:(i + 1) * per], [d] * per) for i in range(B)]
    for coop in (0, 1, -1, 0, 1):
        c.set_param("coopstore", coop)
        ms = []
        for rep in range(6):
            c.reset(); c.sync()
            t = time.perf_counter()
            for b in batches:
                assert c.CarveBatchDevice(b)
            c.sync()
            ms.append((time.perf_counter() - t) * 1e3)
        print("%s %d batches of %d views, coopstore %2d: %.3f ms (min of 6; median %.3f)" % (mode, B, per, coop, min(ms), sorted(ms)[3]))
