import sys
sys.path.insert(0, ".")
import numpy as np
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 6
uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1)
opt = synth.sphere_option(n, uo)
views, masks = synth.sphere_views(n, 32, 1280, 720)
sdf0 = vc.make_sdf(masks[0], use_truncation=True, band=0.1)
res = []
for coop in (0, 1):
    c = vc.VoxelCarver(opt); assert c.Init()
    d = c.upload_sdf(sdf0)
    c.set_param("defer", 0); c.set_param("coopstore", coop)
    per = []
    for i in range(nv):
        assert c.CarveDevice(views[i], d)
        s, u = c.download()
        per.append((s.copy(), u.copy()))
    res.append(per); c.close()
for i in range(nv):
    s0, u0 = res[0][i]; s1, u1 = res[1][i]
    ds = np.flatnonzero(s0.view(np.uint32) != s1.view(np.uint32)); du = np.flatnonzero(u0 != u1)
    print("view", i, "sdf diffs", ds.size, "cnt diffs", du.size)
    if ds.size:
        z, r = np.divmod(ds, n * n); y, x = np.divmod(r, n)
        print("  x range", x.min(), x.max(), "y", y.min(), y.max(), "z", z.min(), z.max())
        print("  x%32 histogram", np.bincount(x % 32, minlength=32))
        print("  y%8", np.bincount(y % 8, minlength=8), "z%8", np.bincount(z % 8, minlength=8))
        for j in ds[:6]:
            print("   idx", j, (j % n, (j // n) % n, j // (n * n)), s0.flat[j], s1.flat[j], u0.flat[j], u1.flat[j])
        break
