"""Runs ON the GPU box: the slab planner against the truth, and what its cuts are worth.
 (1) (brick, view) pairs per brick layer the carve kernel really processes ("paircount") vs the planner's estimate;
 (2) G z-slabs cut equally and cut by the planner, each carved at steady clocks (STEPS steps queued back to back, the
     mean step period of the last 10): per-rank times, spread, predicted speed-up over the whole grid on one GPU.
usage: plan_check.py [G ...]   (default 2 4 8)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vacancy_amd import carver as vc, synth, dist as vdist  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402

n, nv = 1024, 32
STEPS = int(os.environ.get("VCY_PLAN_STEPS", "60"))
Gs = [int(x) for x in sys.argv[1:]] or [2, 4, 8]
mode = os.environ.get("VCY_PLAN_MODE", "default")
uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
views, masks = synth.sphere_views(n, nv, 1280, 720)
opt = synth.sphere_option(n, uo)
sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)


def steady(z0, z1, steps=STEPS):
    c = vc.VoxelCarver(opt, device_id=0, z_range=(z0, z1))
    assert c.Init()
    d = c.upload_sdf(sdf0)
    batch = vc.VoxelCarver.prepare_batch(views, [d] * nv)
    c.reset(); c.CarveBatchDevice(batch); c.sync()
    c.set_param("carvetimer", 1)
    for _ in range(steps):
        c.reset()
        c.CarveBatchDevice(batch)
    c.sync()
    log = c.carve_log()
    c.free_device(d); c.close()
    period = (log[-1][0] - log[-11][0]) / 10
    return period, sum(r[1] for r in log[-10:]) / 10, sum(r[2] for r in log[-10:]) / 10


# (1) truth vs estimate
c = vc.VoxelCarver(opt, device_id=0)
assert c.Init()
d = c.upload_sdf(sdf0)
batch = vc.VoxelCarver.prepare_batch(views, [d] * nv)
c.set_param("paircount", 1)
c.reset(); c.CarveBatchDevice(batch); c.sync()
proc, total, per_layer = c.last_carve_pairs()
c.set_param("paircount", 0)
print("pairs processed %d of %d = %.4f" % (proc, total, proc / total))
c.free_device(d); c.close()

p = vc.VoxelCarver(opt, device_id=0, z_range=(0, 8))
assert p.Init()
d = p.upload_sdf(sdf0)
batch = vc.VoxelCarver.prepare_batch(views, [d] * nv)
for stride in (1, 2, 4):
    t0 = time.perf_counter()
    bounds, cost = p.plan_z_slabs(batch, None, 8, stride=stride, brick_cost=1e-9)
    ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    bounds, cost = p.plan_z_slabs(batch, None, 8, stride=stride, brick_cost=1e-9)
    ms2 = (time.perf_counter() - t0) * 1e3
    est = cost  # (brick_cost ~ 0: the pairs alone)
    print("stride %d: plan call %.2f ms (first), %.2f ms (again); estimated pairs %.0f = %.4f of the truth; "
          "max layer error %.3f of the mean layer; corr %.5f"
          % (stride, ms, ms2, est.sum(), est.sum() / proc, np.abs(est - per_layer).max() / per_layer.mean(),
             np.corrcoef(est, per_layer)[0, 1]))
print("layer: truth / estimate (stride 2), every 8th layer")
bounds, cost = p.plan_z_slabs(batch, None, 8, stride=2, brick_cost=1e-9)
print("  " + "  ".join("%d:%.0fk/%.0fk" % (l, per_layer[l] / 1e3, cost[l] / 1e3) for l in range(0, len(cost), 8)))
np.save(os.path.join(os.environ.get("VCY_OUT", "."), "pairs_truth.npy"), per_layer)
np.save(os.path.join(os.environ.get("VCY_OUT", "."), "pairs_est.npy"), cost)

# (2) what the cuts are worth
t1, pre1, ker1 = steady(0, n)
print("whole grid: step %.3f ms (pre-pass %.3f, kernel %.3f)" % (t1, pre1, ker1))
for G in Gs:
    for name, b in (("equal", [vdist.slab_range(n, s, G)[0] for s in range(G)] + [n]),
                    ("planned", p.plan_z_slabs(batch, None, G)[0])):
        rows = [steady(b[s], b[s + 1]) for s in range(G)]
        per = [r[0] for r in rows]
        print("G=%d %-8s cuts %s" % (G, name, b))
        print("     ranks %s  max %.3f  spread %.1f %%  -> speed-up %.2f  (kernels %s)"
              % ([round(x, 3) for x in per], max(per), 100 * (max(per) - min(per)) / (sum(per) / G), t1 / max(per),
                 [round(r[2], 3) for r in rows]))
p.free_device(d); p.close()
