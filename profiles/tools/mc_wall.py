"""Runs ON the GPU box: what a marching-cubes CALL costs from entry to the mesh in host memory (what the reference's
timer brackets, marching_cubes.cc:65-66,226-227) next to its kernels, for the bunny (resolution 2.5) and sphere grids,
with the mesh written straight to host memory by the last kernel ("mcdirect") and staged + copied.
usage: python profiles/tools/mc_wall.py [n ...]       (default: bunny 256 512 1024)"""
import os
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption


def bunny(res):
    import bunny_data as B
    views = B.bunny_views(lambda t, q: synth.affine_inverse(synth.pose_from_tum(t, q)))
    c = vc.VoxelCarver(B.bunny_option(res))
    assert c.Init(), vc.last_error()
    for v, m in zip(views, B.load_masks()):
        assert c.CarveSilhouette(v, m)
    return c, "bunny res %g (%d x %d x %d)" % ((res,) + tuple(c.dims))


def sphere(n):
    nv = 8
    c = vc.VoxelCarver(synth.sphere_option(n, UpdateOption()))
    assert c.Init(), vc.last_error()
    views, masks = synth.sphere_views(n, nv, 1280, 720)
    d = [c.upload_sdf(vc.make_sdf(masks[0]))] * nv
    assert c.CarveBatchDevice(vc.VoxelCarver.prepare_batch(views, d))
    return c, "%d^3 sphere" % n


args = sys.argv[1:] or ["bunny", "256", "512", "1024"]
for a in args:
    c, label = bunny(2.5) if a == "bunny" else sphere(int(a))
    c.set_param("meshkeys", 0)
    c.sync()
    for direct in (0, 32 << 20, 256 << 20, 0, 32 << 20):
        c.set_param("mcdirect", direct)
        c.ExtractIsoSurface(0.0, True)
        runs = [c.ExtractIsoSurface(0.0, True) for _ in range(9)]
        wall = sorted(m["wall_ms"] for m in runs)
        dev = sorted(m["device_ms"] for m in runs)
        m = runs[-1]
        mb = (m["vertices"].nbytes + m["faces"].nbytes) / 1e6
        print("%-28s mcdirect %9d: wall median %.3f min %.3f ms | kernels median %.3f min %.3f ms | mesh %.2f MB (%d v, %d f)"
              % (label, direct, wall[len(wall) // 2], wall[0], dev[len(dev) // 2], dev[0], mb, len(m["vertices"]), len(m["faces"])))
    if os.environ.get("VCY_MC_TIMING_ONCE"):
        c.set_param("mctiming", 1)
        for direct in (0, 32 << 20):
            c.set_param("mcdirect", direct)
            for _ in range(3):
                c.ExtractIsoSurface(0.0, True)
        c.set_param("mctiming", 0)
    c.close()
