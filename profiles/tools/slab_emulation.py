"""Runs ON the GPU box: the benchmark grid (1024^3 x 32 views at 1280x720) carved slab by slab on ONE GPU, as the ranks
of a G-GPU run would carve it -- each slab at STEADY clocks (STEPS steps queued back to back, the mean step period of
the last 10; a rank of the real run is in that state after bench.py's warm-up, profiles/r04/clock_ramp.txt).
For G = 2, 4, 8: slabs of equal thickness and slabs cut by the planner (vcy_plan_z_slabs), one per GPU; per-rank
step times, spread and the predicted speed-up over the whole grid on one GPU."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vacancy_amd import carver as vc, synth, dist as vdist  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402

n, nv = 1024, 32
STEPS = int(os.environ.get("VCY_PLAN_STEPS", "60"))
views, masks = synth.sphere_views(n, nv, 1280, 720)
opt = synth.sphere_option(n, UpdateOption())
sdf0 = vc.make_sdf(masks[0])


def steady(z0, z1):
    c = vc.VoxelCarver(opt, device_id=0, z_range=(z0, z1))
    assert c.Init()
    d = c.upload_sdf(sdf0)
    batch = vc.VoxelCarver.prepare_batch(views, [d] * nv)
    c.reset(); c.CarveBatchDevice(batch); c.sync()
    c.set_param("carvetimer", 1)
    for _ in range(STEPS):
        c.reset()
        c.CarveBatchDevice(batch)
    c.sync()
    log = c.carve_log()
    c.free_device(d); c.close()
    return (log[-1][0] - log[-11][0]) / 10, sum(r[1] for r in log[-10:]) / 10, sum(r[2] for r in log[-10:]) / 10


t1, pre1, ker1 = steady(0, n)
print("1 GPU: step %.3f ms (pre-pass %.3f + carve kernel %.3f)" % (t1, pre1, ker1))
for G in (2, 4, 8):
    for k in (1, 2):
        S = G * k
        bounds, _, info = vdist.plan_bounds(opt, 0, views, [sdf0] * nv, S)
        for name, b in (("equal", vdist.equal_bounds(n, S)), ("planned", bounds)):
            if name == "equal" and k == 2 and G == 2:
                continue
            slab = [steady(b[s], b[s + 1]) for s in range(S)]
            ranks = [sum(slab[s][0] for s in range(r, S, G)) for r in range(G)]
            mean = sum(ranks) / G
            print("G=%d k=%d %-7s: max rank %.3f ms -> speed-up %.2f; spread (max - min) / mean %.1f %%; ranks %s; cuts %s%s"
                  % (G, k, name, max(ranks), t1 / max(ranks), 100 * (max(ranks) - min(ranks)) / mean,
                     [round(x, 3) for x in ranks], b,
                     "; plan %.2f ms, predicted spread %.1f %%" % (info["plan_ms"], 100 * info["predicted_spread"])
                     if name == "planned" else ""))
