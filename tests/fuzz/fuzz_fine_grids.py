"""Long-running differential fuzz (not collected by pytest): benchmark-like geometry (40..90 voxels per
side, 0.15..1.3 pixels per voxel, 3..11 views, smooth / noisy / silhouette SDFs, every update mode) carved
as one context and as two z-slab contexts, against the oracle bit for bit.
usage: python tests/fuzz/fuzz_fine_grids.py FIRST_SEED LAST_SEED [modes]   (round 1: seeds 0..1000 over the kernel versions of the round, 0 mismatches; round 2: seeds 0..1200, and 1200..2500 with the extra modes, on the final kernels: 0 mismatches)"""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from vacancy_amd import carver as vc, synth
from vacancy_amd.capi import CarverOption, UpdateOption
lo, hi = int(sys.argv[1]), int(sys.argv[2])
extra_modes = len(sys.argv) > 3 and sys.argv[3] == "modes"  # also nearest neighbour, weights != 1, update limits
bad = 0
t0 = time.time()
for seed in range(lo, hi):
    rng = np.random.RandomState(7000 + seed)
    dims = rng.randint(40, 90, 3)
    res = 1.0
    centre = rng.uniform(-30, 30, 3)
    half = dims * res / 2.0
    bb_min = (centre - half).astype(np.float32)
    bb_max = (bb_min + np.float32(res) * dims + np.float32(0.25)).astype(np.float32)
    # (sampling, weight and update limit from their own stream, so that the scenes of a seed stay what they were)
    rng2 = np.random.RandomState(9000 + seed)
    uo = UpdateOption(voxel_update=int(rng.randint(0, 2)), update_outside=int(rng.randint(0, 2)),
                      use_truncation=bool(rng.randint(0, 2)), truncation_band=float(rng.choice([0.1, 0.35])),
                      sdf_interp=int(rng2.randint(0, 2)) if extra_modes else 1,
                      voxel_update_weight=float(rng2.choice([1.0, 1.0, 0.5, 2.25])) if extra_modes else 1.0,
                      voxel_max_update_num=int(rng2.choice([255, 255, 255, 3, 1000])) if extra_modes else 255)
    opt = CarverOption(bb_min=[float(x) for x in bb_min], bb_max=[float(x) for x in bb_max], resolution=res, update_option=uo)
    nviews = int(rng.randint(3, 12))
    extent = float(np.linalg.norm(half))
    views, sdfs = [], []
    for _ in range(nviews):
        w, h = int(rng.randint(150, 420)), int(rng.randint(120, 330))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        dist = extent * rng.uniform(1.3, 4.0)
        pos = centre + d * dist
        up = (0.0, 1.0, 0.0) if abs(d[1]) < 0.9 else (1.0, 0.0, 0.0)
        w2c = synth.affine_inverse(synth.lookat_c2w(pos, centre + rng.uniform(-0.1, 0.1, 3) * half, up)).astype(np.float32)
        ppv = rng.uniform(0.15, 1.3)           # pixels per voxel at the centre
        f = float(ppv * dist / res)
        fx, fy = (f, f) if rng.rand() < 0.6 else (f, f * float(rng.uniform(0.9, 1.1)))
        rmin = rmax = None
        if rng.rand() < 0.3:
            rmin = (int(rng.randint(0, w // 4)), int(rng.randint(0, h // 4))); rmax = (int(rng.randint(3 * w // 4, w)) - 1, int(rng.randint(3 * h // 4, h)) - 1)
        views.append(vc.make_view(w2c, np.float32(fx), np.float32(fy), np.float32(w / 2 - 0.5), np.float32(h / 2 - 0.5), w, h, rmin, rmax, False))
        yy, xx = np.mgrid[0:h, 0:w]
        style = rng.randint(0, 3)
        r0 = min(w, h) * rng.uniform(0.1, 0.45)
        if style == 0:
            mask = ((np.hypot(xx - w / 2, yy - h / 2) < r0) * 255).astype(np.uint8)
            img = O.make_sdf(mask, rmin, rmax, True, bool(uo.use_truncation), uo.truncation_band)
        elif style == 1:
            img = (np.hypot(xx - w / 2, yy - h / 2) - r0) / max(w, h) + rng.normal(0, 0.002, (h, w))
        else:
            img = np.sin(xx / rng.uniform(5, 40)) * np.cos(yy / rng.uniform(5, 40)) * rng.uniform(0.2, 1.5)
        sdfs.append(np.ascontiguousarray(img, np.float32))
    orc = O.OracleGrid(opt)
    for v, s in zip(views, sdfs):
        orc.carve(v, s)
    os_, ou = orc.download()
    nz = orc.dims[2]
    cut = int(rng.randint(2, nz))
    for zr in (None, (0, cut), (cut, nz)):
        dev = vc.VoxelCarver(opt, z_range=zr) if zr else vc.VoxelCarver(opt)
        assert dev.Init()
        d = [dev.upload_sdf(s) for s in sdfs]
        assert dev.CarveBatchDevice(views, d)
        ds, du = dev.download()
        sl = slice(None) if zr is None else slice(zr[0] * orc.dims[0] * orc.dims[1], zr[1] * orc.dims[0] * orc.dims[1])
        eo, eu = os_[sl], ou[sl]
        nan_d, nan_o = np.isnan(ds), np.isnan(eo)
        ok = np.array_equal(du, eu) and np.array_equal(nan_d, nan_o) and np.array_equal(np.where(nan_d, 0, ds.view(np.uint32)), np.where(nan_o, 0, eo.view(np.uint32)))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "zr", zr, int((du != eu).sum()), int((ds.view(np.uint32) != eo.view(np.uint32)).sum()))
        dev.close()
print("fuzz2 seeds %d..%d done, %d mismatches, %.0f s" % (lo, hi, bad, time.time() - t0))
