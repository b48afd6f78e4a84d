"""Runs ON the GPU box: the two slabs a rank of an 8-GPU run holds (slabs r and r + 8 of 16), carved on ONE stream one
after the other, and on a stream each (concurrently); wall ms per step incl. the final sync."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vacancy_amd import carver as vc, synth, dist as vdist
from vacancy_amd.capi import UpdateOption
n, nv = 1024, 32
views, masks = synth.sphere_views(n, nv, 1280, 720)
opt = synth.sphere_option(n, UpdateOption())
sdf0 = vc.make_sdf(masks[0])
for G, k in ((8, 2), (4, 2)):
    for r in (0, G // 2 - 1):
        for shared in (1, 0, 1, 0):
            cs = []
            for _, z0, z1 in vdist.slabs_of_rank(n, r, G, k):
                c = vc.VoxelCarver(opt, device_id=0, z_range=(z0, z1)); assert c.Init()
                if cs and shared:
                    c.use_stream_of(cs[0])
                cs.append(c)
            d = cs[0].upload_sdf(sdf0)
            batch = vc.VoxelCarver.prepare_batch(views, [d] * nv)
            best = 1e9
            for it in range(6):
                for c in cs: c.reset()
                for c in cs: c.sync()
                t0 = time.perf_counter()
                for c in cs: assert c.CarveBatchDevice(batch)
                for c in cs: c.sync()
                best = min(best, (time.perf_counter() - t0) * 1e3)
            print("G=%d k=%d rank %d: %s -> %.3f ms per step" % (G, k, r, "one stream " if shared else "two streams", best))
            cs[0].free_device(d)
            for c in reversed(cs): c.close()
