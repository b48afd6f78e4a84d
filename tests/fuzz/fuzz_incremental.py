"""Long-running differential fuzz (not collected by pytest) of the paths round 3 added: the views of a scene are
carved in GROUPS of 1..5 per call on a state that is already carved (footprint records, drop against the brick
minima before any state is read, live-workgroup list on / off, several chunks of brick layers), with extractions
in between (bricks outside the surface skipped: "mcskip" 1, against 0 and against the oracle at random iso
levels) and writes that bypass the fused kernel (vcy_upload, the per-view kernel).  State against the oracle bit
for bit after every group, as one context and as two z-slab contexts.
usage: python tests/fuzz/fuzz_incremental.py FIRST_SEED LAST_SEED [modes]   (round 3: seeds 0..520, 1 560 contexts, chunked launches mixed in from 400 on, and 1000..1150 with the extra modes: 0 mismatches)"""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from vacancy_amd import carver as vc, synth
from vacancy_amd.capi import CarverOption, UpdateOption

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()


def same_state(ds, du, eo, eu):
    nan_d, nan_o = np.isnan(ds), np.isnan(eo)
    return (np.array_equal(du, eu) and np.array_equal(nan_d, nan_o) and
            np.array_equal(np.where(nan_d, 0, ds.view(np.uint32)), np.where(nan_o, 0, eo.view(np.uint32))))


def same_mesh(a, b):
    return (a["vertices"].shape == b["vertices"].shape and a["faces"].shape == b["faces"].shape and
            np.array_equal(a["keys"], b["keys"]) and np.array_equal(a["faces"], b["faces"]) and
            np.array_equal(a["vertices"].view(np.uint32), b["vertices"].view(np.uint32)))


for seed in range(lo, hi):
    if seed > lo and (seed - lo) % 10 == 0:
        print("  ... seed %d, %d mismatches so far, %.0f s" % (seed, bad, time.time() - t0), flush=True)
    rng = np.random.RandomState(17000 + seed)
    dims = rng.randint(40, 100, 3)
    if rng.rand() < 0.3:
        dims[0] = int(rng.choice([64, 128]))  # rows of whole 64-voxel words: the brick-row pass of marching cubes
    if seed >= 2000:  # (round 4) rows of whole bricks, which the vector loads / stores and the cooperative write-back need
        dims[0] = int(np.random.RandomState(29000 + seed).choice([48, 56, 64, 72, 80, 96, 104, 128]))
    centre = rng.uniform(-30, 30, 3)
    half = dims / 2.0
    bb_min = (centre - half).astype(np.float32)
    bb_max = (bb_min + np.float32(1.0) * dims + np.float32(0.25)).astype(np.float32)
    rng2 = np.random.RandomState(19000 + seed)  # (its own stream: the scenes of a seed stay what they were)
    extra = len(sys.argv) > 3 and sys.argv[3] == "modes"  # also nearest neighbour, weights != 1, update limits in reach
    uo = UpdateOption(voxel_update=int(rng.randint(0, 2)), update_outside=int(rng.randint(0, 2)),
                      use_truncation=bool(rng.randint(0, 2)), truncation_band=float(rng.choice([0.1, 0.35])),
                      sdf_interp=int(rng2.randint(0, 2)) if extra else 1,
                      voxel_update_weight=float(rng2.choice([1.0, 1.0, 0.5, 2.25])) if extra else 1.0,
                      voxel_max_update_num=int(rng2.choice([255, 255, 3, 6, 1000])) if extra else 255)
    opt = CarverOption(bb_min=[float(x) for x in bb_min], bb_max=[float(x) for x in bb_max], resolution=1.0, update_option=uo)
    nviews = int(rng.randint(6, 16))
    extent = float(np.linalg.norm(half))
    views, sdfs = [], []
    for _ in range(nviews):
        w, h = int(rng.randint(150, 420)), int(rng.randint(120, 330))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        dist = extent * rng.uniform(1.3, 4.0)
        pos = centre + d * dist
        up = (0.0, 1.0, 0.0) if abs(d[1]) < 0.9 else (1.0, 0.0, 0.0)
        w2c = synth.affine_inverse(synth.lookat_c2w(pos, centre + rng.uniform(-0.1, 0.1, 3) * half, up)).astype(np.float32)
        f = float(rng.uniform(0.15, 1.3) * dist)
        views.append(vc.make_view(w2c, np.float32(f), np.float32(f), np.float32(w / 2 - 0.5), np.float32(h / 2 - 0.5), w, h, None, None, False))
        yy, xx = np.mgrid[0:h, 0:w]
        style = rng.randint(0, 3)
        r0 = min(w, h) * rng.uniform(0.1, 0.45)
        if style == 0:
            mask = ((np.hypot(xx - w / 2, yy - h / 2) < r0) * 255).astype(np.uint8)
            img = O.make_sdf(mask, None, None, True, bool(uo.use_truncation), uo.truncation_band)
        elif style == 1:
            img = (np.hypot(xx - w / 2, yy - h / 2) - r0) / max(w, h) + rng.normal(0, 0.002, (h, w))
        else:
            img = np.sin(xx / rng.uniform(5, 40)) * np.cos(yy / rng.uniform(5, 40)) * rng.uniform(0.2, 1.5)
        sdfs.append(np.ascontiguousarray(img, np.float32))
    # the groups, and what happens between them (the same script for every context of this seed)
    script, i = [], 0
    while i < nviews:
        g = int(rng.randint(1, 6))
        script.append((i, min(nviews, i + g), int(rng.randint(0, 2)), rng.rand() < 0.4, float(rng.choice([0.0, 0.0, 0.05, -0.1, 0.3])),
                       rng.rand() < 0.12, rng.rand() < 0.12))
        i += g
    nz = int(dims[2])
    cut = int(rng.randint(2, nz - 1))
    for zr in (None, (0, cut), (cut, nz)):
        dev = vc.VoxelCarver(opt, z_range=zr) if zr else vc.VoxelCarver(opt)
        assert dev.Init()
        dev.set_param("mcskip", 2)  # the brick-row pass on these small grids too (the library's rule: rows of 1024 voxels and more)
        # (round 4) write-back through LDS in whole row segments: the library's rule, never, wherever the layout allows
        dev.set_param("coopstore", int(np.random.RandomState(23000 + seed).choice([-1, 0, 1, 1])))
        # (round 6) the few-view flavour of the fused kernel (a wave walks the bricks of a row segment): its rule, never,
        # launches of up to 2 views only -- drawn from a stream of its own so that the scenes of earlier rounds' seeds stay
        dev.set_param("rowkernel", int(np.random.RandomState(61000 + seed).choice([-1, 0, 0, 2])))
        dev.set_param("listrecords", int(np.random.RandomState(64000 + seed).choice([1, 1, 0])))  # live-list entries with / without the records
        dev.set_param("eagerstate", int(np.random.RandomState(63000 + seed).choice([-1, -1, 0, 1])))  # state requested next to the footprint record
        dev.set_param("oneview", int(np.random.RandomState(62000 + seed).choice([1, 1, 0])))  # the instance compiled for ONE view, or the general one
        dev.set_param("defer", int(rng.randint(0, 2)))
        dev.set_param("recordbytes", int(rng.choice([0, 0, 3000, 20000])))  # chunks of a few brick layers
        orc = O.OracleGrid(opt)
        n_xy = orc.dims[0] * orc.dims[1]
        sl = slice(None) if zr is None else slice(zr[0] * n_xy, zr[1] * n_xy)
        d = [dev.upload_sdf(s) for s in sdfs]
        for (a, b, livelist, extract, iso, do_upload, do_generic) in script:
            dev.set_param("livelist", livelist)
            if b - a == 1:
                ok = dev.CarveDevice(views[a], d[a])
            else:
                ok = dev.CarveBatchDevice(views[a:b], d[a:b])
            assert ok, vc.last_error()
            for k in range(a, b):
                orc.carve(views[k], sdfs[k])
            os_, ou = orc.download()
            ds, du = dev.download()
            if not same_state(ds, du, os_[sl], ou[sl]):
                bad += 1
                print("MISMATCH state seed", seed, "zr", zr, "views", a, b, int((du != ou[sl]).sum()))
                if os.environ.get("FUZZ_VERBOSE"):
                    eo, eu = os_[sl], ou[sl]
                    idx = np.nonzero((du != eu) | (ds.view(np.uint32) != eo.view(np.uint32)))[0]
                    print("  mode", uo.voxel_update, uo.update_outside, uo.use_truncation, uo.truncation_band, "dims", orc.dims,
                          "defer", dev.get_param("defer"), "script", script)
                    print("  first diffs", [(int(j), float(ds[j]), float(eo[j]), int(du[j]), int(eu[j])) for j in idx[:6]],
                          "n sdf diff", int((ds.view(np.uint32) != eo.view(np.uint32)).sum()))
                break
            if extract and zr is None and not np.isnan(os_).any():
                om = orc.marching_cubes(iso, True)
                m1 = dev.ExtractIsoSurface(iso, True)
                dev.set_param("mcskip", 0)
                m0 = dev.ExtractIsoSurface(iso, True)
                dev.set_param("mcskip", 2)
                if not (same_mesh(m1, om) and same_mesh(m0, om)):
                    bad += 1
                    print("MISMATCH mesh seed", seed, "views", a, b, "iso", iso, len(m1["vertices"]), len(m0["vertices"]), len(om["vertices"]))
                    break
            if do_upload:    # a state from outside (the same arrays on both sides)
                s2 = np.where(np.isnan(ds), np.float32(0), ds) + np.float32(0.125) * (np.arange(ds.size) % 3 == 0)
                s2 = s2.astype(np.float32)
                dev.upload(s2, du)
                full_s, full_u = os_.copy(), ou.copy()
                full_s[sl] = s2
                full_u[sl] = du
                orc.upload(full_s, full_u)
            if do_generic:   # the per-view kernel does not keep the brick minima
                dev.set_param("fused", 0)
                assert dev.CarveDevice(views[a], d[a])
                orc.carve(views[a], sdfs[a])
                dev.set_param("fused", 1)
        dev.close()
print("fuzz3 seeds %d..%d done, %d mismatches, %.0f s" % (lo, hi, bad, time.time() - t0))
