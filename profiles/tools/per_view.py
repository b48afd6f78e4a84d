"""Runs ON the GPU box: the reference's call pattern on the bench scene -- one view per launch ("defer" 0) -- with the
live-workgroup list on and off; kernel ms per view (HIP events) and the state hash after all views.
usage: python profiles/tools/per_view.py [n] [mode] [coopstore values, e.g. 0,1]"""
import hashlib
import sys
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
c = vc.VoxelCarver(opt)
assert c.Init()
d = c.upload_sdf(sdf0)
c.set_param("defer", 0)
coops = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [-1]
for ll, coop in [(ll, co) for ll in (1, 0, 1, 0) for co in coops]:
    c.set_param("livelist", ll)
    c.set_param("coopstore", coop)
    c.reset()
    ms = []
    for i in range(nv):
        c.timer_begin()
        assert c.CarveDevice(views[i], d)
        ms.append(c.timer_end())
    ids = (n // 2) * n * n + (n // 2) * n + __import__("numpy").arange(0, n, dtype="int64")
    s_, u_ = c.download_voxels(ids)
    h = hashlib.sha1(s_.tobytes() + u_.tobytes()).hexdigest()[:10]
    tot = sum(ms)
    print("%s coopstore %d livelist %d: total %.2f ms  first %.2f  others avg %.3f  min %.3f max %.3f  -> %.0f Mvoxel*views/s  (centre row %s)"
          % (mode, coop, ll, tot, ms[0], (tot - ms[0]) / (nv - 1), min(ms[1:]), max(ms[1:]), float(n) ** 3 * nv / tot / 1e3, h))
