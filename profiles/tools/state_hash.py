"""Development aid: hashes of the carved state of a small version of the bench scene (512^3, 32 views at
640x360: the same voxel footprint as the 1024^3 benchmark), for the default, cull 0 and TSDF modes.
Kernel variants under test (VCY_HIP_LIB=...) must print the hashes of the tested production build."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from vacancy_amd import synth  # noqa: E402
from vacancy_amd import carver as vc  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402


def main():
    n, nv, w, h = 512, 32, 640, 360
    out = []
    for name, mode, cull in (("default", "default", 1), ("cull0", "default", 0), ("tsdf", "tsdf", 1)):
        uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
        opt = synth.sphere_option(n, uo)
        views, masks = synth.sphere_views(n, nv, w, h)
        sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
        c = vc.VoxelCarver(opt)
        assert c.Init(), vc.last_error()
        c.set_param("cull", cull)
        d = [c.upload_sdf(sdf0)] * nv
        assert c.CarveBatchDevice(vc.VoxelCarver.prepare_batch(views, d)), vc.last_error()
        s, u = c.download()
        out.append("%s:%s" % (name, hashlib.sha1(s.tobytes() + u.tobytes()).hexdigest()[:12]))
        del c
    print(" ".join(out))


if __name__ == "__main__":
    main()
