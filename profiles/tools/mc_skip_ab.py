import os, sys
sys.path.insert(0, os.environ.get("VCY_ROOT", "."))
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n, nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 32
opt = synth.sphere_option(n, UpdateOption())
views, masks = synth.sphere_views(n, nv, 1280, 720)
sdf0 = vc.make_sdf(masks[0])
c = vc.VoxelCarver(opt)
assert c.Init()
d = [c.upload_sdf(sdf0)] * nv
assert c.CarveBatchDevice(vc.VoxelCarver.prepare_batch(views, d))
c.set_param("meshkeys", 0)
for rs in (1, 0, 1, 0):
    c.set_param("mcskip", rs)
    c.ExtractIsoSurface(0.0, True)
    t = sorted(c.ExtractIsoSurface(0.0, True)["device_ms"] for _ in range(7))
    print("n %d mcskip %d: device ms min %.3f median %.3f" % (n, rs, t[0], t[3]))
