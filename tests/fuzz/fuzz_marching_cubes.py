"""Long-running differential fuzz of marching cubes alone (not collected by pytest): arbitrary uploaded state
(noise, smooth runs, invalid holes, untouched voxels, values in the 1e-5 snap band) on grids whose rows are 1 .. 8
whole 64-voxel words (the one-sweep cell search) or ragged (the bit-plane path), as one context and as two
z-slab contexts with the halo installed, "mcsweep" 1 and 0, float and non-float iso levels, both interpolation
modes -- every mesh against the oracle array for array.
usage: python tests/fuzz/fuzz_marching_cubes.py FIRST_SEED LAST_SEED   (round 2: seeds 0..2500 over the kernel versions of the round, 0..2300 on the final ones: 0 mismatches)"""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from vacancy_amd import carver as vc
from vacancy_amd.capi import CarverOption, UpdateOption

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()


def same(a, b):
    return (a["vertices"].shape == b["vertices"].shape and a["faces"].shape == b["faces"].shape and
            np.array_equal(a["keys"], b["keys"]) and np.array_equal(a["faces"], b["faces"]) and
            np.array_equal(a["vertices"].view(np.uint32), b["vertices"].view(np.uint32)))


for seed in range(lo, hi):
    rng = np.random.RandomState(31000 + seed)
    nx = int(rng.choice([64, 64, 128, 128, 256, 512, 100, 192]))
    ny, nz = int(rng.randint(2, 90)), int(rng.randint(3, 80))
    if nx >= 256:
        ny, nz = min(ny, 50), min(nz, 40)
    dims = (nx, ny, nz)
    uo = UpdateOption(voxel_max_update_num=int(rng.choice([200, 255, 70000])))  # u8 / u16 / u32 counters
    opt = CarverOption(bb_min=[0.0, 0.0, 0.0], bb_max=[float(d) for d in dims], resolution=1.0, update_option=uo)
    orc = O.OracleGrid(opt)
    assert orc.dims == dims, (orc.dims, dims)
    n = orc.n
    x = np.arange(n) % nx
    y = (np.arange(n) // nx) % ny
    z = np.arange(n) // (nx * ny)
    style = rng.randint(0, 3)
    if style == 0:
        sdf = rng.uniform(-1, 1, n)
    elif style == 1:
        sdf = np.sin(x * rng.uniform(0.02, 0.3)) * np.cos(y * rng.uniform(0.02, 0.3) + z * rng.uniform(0.0, 0.3)) + rng.uniform(-0.3, 0.3)
    else:
        c = np.array(dims) * rng.uniform(0.3, 0.7, 3)
        sdf = (np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) - min(dims) * rng.uniform(0.2, 0.6)) / max(dims)
        sdf += rng.normal(0, 0.01, n) * (rng.rand() < 0.5)
    sdf = sdf.astype(np.float32)
    sdf[rng.rand(n) < rng.choice([0.0, 0.01, 0.1])] = np.finfo(np.float32).min
    snap = rng.rand(n) < 0.02
    sdf[snap] = rng.uniform(-2e-5, 2e-5, int(snap.sum())).astype(np.float32)
    cnt = (rng.rand(n) < rng.choice([1.0, 0.98, 0.7])).astype(np.int32)
    orc.upload(sdf, cnt)
    iso = float(rng.choice([0.0, 0.0, 0.25, 0.3, -0.1]))
    interp = bool(rng.randint(0, 2))
    ref = orc.marching_cubes(iso, interp)
    whole = vc.VoxelCarver(opt)
    assert whole.Init(), vc.last_error()
    whole.upload(sdf, cnt)
    for sweep in (1, 0):
        whole.set_param("mcsweep", sweep)
        if not same(whole.ExtractIsoSurface(iso, interp), ref):
            bad += 1
            print("MISMATCH seed", seed, dims, "whole sweep", sweep, "iso", iso, interp)
    cut = int(rng.randint(2, nz)) if nz > 2 else 0
    if cut:
        sl = nx * ny
        lower = vc.VoxelCarver(opt, z_range=(0, cut)); upper = vc.VoxelCarver(opt, z_range=(cut, nz))
        assert lower.Init() and upper.Init(), vc.last_error()
        lower.upload(sdf[:cut * sl], cnt[:cut * sl]); upper.upload(sdf[cut * sl:], cnt[cut * sl:])
        upper.halo_install_host(lower.halo_pack_host())
        for c, (z0, z1) in ((lower, (0, cut)), (upper, (cut, nz))):
            sref = O.marching_cubes_slab(orc, z0, z1, iso, interp)
            for sweep in (1, 0):
                c.set_param("mcsweep", sweep)
                m = c.ExtractIsoSurface(iso, interp)
                if not (same(m, sref) and m["n_foreign"] == sref["n_foreign"]):
                    bad += 1
                    print("MISMATCH seed", seed, dims, "slab", (z0, z1), "sweep", sweep, "iso", iso, interp)
        lower.close(); upper.close()
    whole.close()
print("fuzz mc seeds %d..%d done, %d mismatches, %.0f s" % (lo, hi, bad, time.time() - t0))
