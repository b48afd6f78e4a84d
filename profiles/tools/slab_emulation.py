import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vacancy_amd import carver as vc, synth, dist as vdist
from vacancy_amd.capi import UpdateOption
n, nv = 1024, 32
views, masks = synth.sphere_views(n, nv, 1280, 720)
opt = synth.sphere_option(n, UpdateOption())
sdf0 = vc.make_sdf(masks[0])
def time_slab(z0, z1):
    c = vc.VoxelCarver(opt, device_id=0, z_range=(z0, z1)); assert c.Init()
    d = [c.upload_sdf(sdf0)] * nv
    batch = vc.VoxelCarver.prepare_batch(views, d)
    best = 1e9; wall = 1e9
    for it in range(4):
        c.reset(); c.sync(); t0 = time.perf_counter(); c.timer_begin(); c.CarveBatchDevice(batch); ms = c.timer_end(); w = (time.perf_counter() - t0) * 1e3
        best = min(best, ms); wall = min(wall, w)
    c.free_device(d[0]); c.close()
    return best, wall
t1, w1 = time_slab(0, n)
print("1 GPU: %.3f ms (wall %.3f)" % (t1, w1))
for G in (2, 4, 8):
    for k in (1, 2, 4):
        per_rank = []
        for r in range(G):
            tot = 0.0; wtot = 0.0
            for _, z0, z1 in vdist.slabs_of_rank(n, r, G, k):
                t, w = time_slab(z0, z1); tot += t; wtot += w
            per_rank.append((tot, wtot))
        mx = max(p[0] for p in per_rank); mxw = max(p[1] for p in per_rank)
        print("G=%d k=%d: max rank %.3f ms (wall %.3f) -> speedup %.2f (wall %.2f); ranks %s" % (G, k, mx, mxw, t1 / mx, w1 / mxw, [round(p[0], 2) for p in per_rank]))
