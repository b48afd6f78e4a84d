"""Long-running differential fuzz (not collected by pytest): the random scenes of test_gpu_random.py
for an arbitrary seed range, fused kernel with both tile sizes against the oracle.
usage: python tests/fuzz/fuzz_random_scenes.py FIRST_SEED LAST_SEED   (round 1: seeds 100..4200 over the kernel versions of the round, 0 mismatches; round 2: seeds 0..3000 on the final kernels, 0 mismatches)"""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from vacancy_amd import carver as vc
import test_gpu_random as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for seed in range(lo, hi):
    opt, views, sdfs, rng = T._random_case(seed)
    orc = O.OracleGrid(opt)
    for v, s in zip(views, sdfs):
        orc.carve(v, s)
    os_, ou = orc.download()
    for tile in (0, 2):
        dev = vc.VoxelCarver(opt); assert dev.Init()
        dev.set_param("tile", tile)
        d = [dev.upload_sdf(s) for s in sdfs]
        assert dev.CarveBatchDevice(views, d)
        ds, du = dev.download()
        nan_d, nan_o = np.isnan(ds), np.isnan(os_)
        ok = np.array_equal(du, ou) and np.array_equal(nan_d, nan_o) and np.array_equal(np.where(nan_d, 0, ds.view(np.uint32)), np.where(nan_o, 0, os_.view(np.uint32)))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "tile", tile, int((du != ou).sum()), int((ds.view(np.uint32) != os_.view(np.uint32)).sum()))
        dev.close()
print("seeds %d..%d done, %d mismatches, %.0f s" % (lo, hi, bad, time.time() - t0))
