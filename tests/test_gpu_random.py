"""Randomised differential test: random grids (non-cubic), cameras (pinhole with fx != fy,
orthographic, inside / behind / far), ROIs, SDF images (smooth, noisy, with lowest() pixels) and
every option combination; HIP path (fused and per-view kernels, both tile sizes) vs the oracle,
bit for bit, state and marching-cubes mesh."""
import numpy as np
import pytest

import oracle_lib as O
from vacancy_amd import carver as vc
from vacancy_amd import synth
from vacancy_amd.capi import CarverOption, UpdateOption

pytestmark = pytest.mark.gpu


def _random_case(seed):
    rng = np.random.RandomState(1000 + seed)
    dims = rng.randint(5, 40, 3)
    res = float(rng.choice([0.5, 1.0, 1.7, 3.0]))
    centre = rng.uniform(-20, 20, 3)
    half = dims * res / 2.0
    bb_min = (centre - half).astype(np.float32)
    bb_max = (bb_min + np.float32(res) * dims + np.float32(res * 0.25)).astype(np.float32)
    uo = UpdateOption(voxel_update=int(rng.randint(0, 2)), sdf_interp=int(rng.randint(0, 2)),
                      update_outside=int(rng.randint(0, 2)),
                      voxel_max_update_num=int(rng.choice([1, 2, 3, 255, 1000])),
                      voxel_update_weight=float(rng.choice([1.0, 0.5, 2.25])),
                      use_truncation=bool(rng.randint(0, 2)), truncation_band=float(rng.choice([0.1, 0.35])))
    opt = CarverOption(bb_min=[float(x) for x in bb_min], bb_max=[float(x) for x in bb_max], resolution=res,
                       update_option=uo)
    nviews = int(rng.randint(1, 7))
    views, sdfs = [], []
    extent = float(np.linalg.norm(half))
    for _ in range(nviews):
        w, h = int(rng.randint(12, 220)), int(rng.randint(12, 160))
        kind = rng.randint(0, 10)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        if kind == 0:      # camera inside the grid
            pos = centre + rng.uniform(-0.3, 0.3, 3) * half
        elif kind == 1:    # very close
            pos = centre + d * extent * 1.05
        else:
            pos = centre + d * extent * rng.uniform(1.5, 6.0)
        up = (0.0, 1.0, 0.0) if abs(d[1]) < 0.9 else (1.0, 0.0, 0.0)
        target = centre + rng.uniform(-0.2, 0.2, 3) * half
        w2c = synth.affine_inverse(synth.lookat_c2w(pos, target, up)).astype(np.float32)
        ortho = rng.rand() < 0.15
        if ortho:
            w2c = w2c.copy()
            w2c[0, 3] += np.float32(w / 2)
            w2c[1, 3] += np.float32(h / 2)
        f = float(rng.uniform(0.6, 3.0)) * max(w, h)
        fx, fy = (f, f) if rng.rand() < 0.5 else (f, f * float(rng.uniform(0.8, 1.25)))
        if rng.rand() < 0.4:
            x0, y0 = int(rng.randint(0, w // 3)), int(rng.randint(0, h // 3))
            x1, y1 = int(rng.randint(2 * w // 3, w)), int(rng.randint(2 * h // 3, h))
            rmin, rmax = (x0, y0), (min(x1, w - 1), min(y1, h - 1))
        else:
            rmin, rmax = None, None
        views.append(vc.make_view(w2c, np.float32(fx), np.float32(fy), np.float32(w / 2 - 0.5 + rng.uniform(-3, 3)),
                                  np.float32(h / 2 - 0.5 + rng.uniform(-3, 3)), w, h, rmin, rmax, ortho))
        yy, xx = np.mgrid[0:h, 0:w]
        style = rng.randint(0, 4)
        if style == 0:
            img = np.hypot(xx - w / 2, yy - h / 2) / max(w, h) - rng.uniform(0.1, 0.4)
        elif style == 1:
            img = rng.uniform(-1.5, 1.0, (h, w))
        elif style == 2:
            mask = ((np.hypot(xx - w / 2, yy - h / 2) < min(w, h) * rng.uniform(0.15, 0.45)) * 255).astype(np.uint8)
            img = O.make_sdf(mask, rmin, rmax, True, bool(uo.use_truncation), uo.truncation_band)
        else:
            img = np.round(rng.uniform(-1, 1, (h, w)) * 3) / 3
        sdfs.append(np.ascontiguousarray(img, np.float32))
    return opt, views, sdfs, rng


@pytest.mark.parametrize("seed", range(36))
def test_random_scene(seed):
    opt, views, sdfs, rng = _random_case(seed)
    orc = O.OracleGrid(opt)
    for v, s in zip(views, sdfs):
        orc.carve(v, s)
    os_, ou = orc.download()
    iso = float(rng.choice([0.0, 0.05, -0.1]))
    om = orc.marching_cubes(iso, True)
    for fused, cull, tile in ((1, 1, 0), (1, 0, 2), (0, 0, 0)):
        dev = vc.VoxelCarver(opt)
        assert dev.Init(), vc.last_error()
        assert dev.dims == orc.dims
        dev.set_param("fused", fused)
        dev.set_param("cull", cull)
        dev.set_param("tile", tile)
        d = [dev.upload_sdf(s) for s in sdfs]
        assert dev.CarveBatchDevice(views, d), vc.last_error()
        ds, du = dev.download()
        assert np.array_equal(du, ou), (seed, fused, cull, tile, int((du != ou).sum()))
        assert np.array_equal(ds.view(np.uint32), os_.view(np.uint32)), \
            (seed, fused, cull, tile, int((ds.view(np.uint32) != os_.view(np.uint32)).sum()))
        dm = dev.ExtractIsoSurface(iso, True)
        assert np.array_equal(dm["keys"], om["keys"]) and np.array_equal(dm["faces"], om["faces"])
        assert np.array_equal(dm["vertices"].view(np.uint32), om["vertices"].view(np.uint32))
        dev.close()
    # the same scene as two z-slab contexts cut at a random layer (what two ranks hold): the slab
    # states tile the oracle's grid (the window maxima of a slab only cover its image band)
    nz = orc.dims[2]
    if nz >= 3:
        cut = int(rng.randint(2, nz))  # a non-first slab starts at z >= 2 (two halo slices below it)
        parts = []
        for z0, z1 in ((0, cut), (cut, nz)):
            dev = vc.VoxelCarver(opt, z_range=(z0, z1))
            assert dev.Init(), vc.last_error()
            d = [dev.upload_sdf(s) for s in sdfs]
            assert dev.CarveBatchDevice(views, d), vc.last_error()
            parts.append(dev.download())
            dev.close()
        ds = np.concatenate([p[0] for p in parts])
        du = np.concatenate([p[1] for p in parts])
        assert np.array_equal(du, ou), (seed, "slabs", cut)
        assert np.array_equal(ds.view(np.uint32), os_.view(np.uint32)), (seed, "slabs", cut)


@pytest.mark.parametrize("seed", range(12))
def test_random_silhouettes_device_sdf_and_batches(seed):
    """Random masks (blobs, noise, values other than 0/255, empty, full) with random ROIs: the device
    SDF builder (single and batched/streamed) equals the oracle's MakeSignedDistanceField bit for bit,
    and carving them through vcy_carve_batch_silhouettes equals the oracle's per-view loop."""
    rng = np.random.RandomState(500 + seed)
    n = int(rng.randint(12, 30))
    uo = UpdateOption(voxel_update=int(rng.randint(0, 2)), use_truncation=bool(rng.randint(0, 2)),
                      truncation_band=float(rng.choice([0.1, 0.4])))
    opt = synth.sphere_option(n, uo)
    opt.sdf_minmax_normalize = int(rng.randint(0, 2))
    nv = int(rng.randint(2, 40))
    w, h = int(rng.randint(8, 150)), int(rng.randint(8, 110))
    views, _ = synth.sphere_views(n, nv, w, h)
    masks = []
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(nv):
        kind = rng.randint(0, 6)
        if kind == 0:
            m = np.zeros((h, w), np.uint8)
        elif kind == 1:
            m = np.full((h, w), 255, np.uint8)
        elif kind == 2:
            m = ((rng.rand(h, w) < rng.uniform(0.05, 0.95)) * 255).astype(np.uint8)
        else:
            cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(2, max(w, h) / 2)
            m = ((np.hypot(xx - cx, yy - cy) < r) * 255).astype(np.uint8)
            if kind == 5:
                m[rng.rand(h, w) < 0.05] = rng.randint(1, 255)
        masks.append(m)
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    for m in masks[:4]:
        x0, y0 = int(rng.randint(0, w // 2)), int(rng.randint(0, h // 2))
        x1, y1 = int(rng.randint(x0, w)), int(rng.randint(y0, h))
        for rmin, rmax in ((None, None), ((x0, y0), (x1, y1))):
            d = dev.make_sdf_device(m, rmin, rmax, bool(opt.sdf_minmax_normalize), bool(uo.use_truncation),
                                    uo.truncation_band)
            got = dev.download_image(d, m.shape)
            dev.free_device(d)
            ref = O.make_sdf(m, rmin, rmax, bool(opt.sdf_minmax_normalize), bool(uo.use_truncation),
                             uo.truncation_band)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (seed, rmin, rmax)
    assert dev.CarveBatchSilhouettes(views, masks), vc.last_error()
    orc = O.OracleGrid(opt)
    for v, m in zip(views, masks):
        orc.carve(v, O.make_sdf(m, None, None, bool(opt.sdf_minmax_normalize), bool(uo.use_truncation),
                                uo.truncation_band))
    ds, du = dev.download()
    os_, ou = orc.download()
    nan_d, nan_o = np.isnan(ds), np.isnan(os_)
    assert np.array_equal(du, ou) and np.array_equal(nan_d, nan_o)
    assert np.array_equal(np.where(nan_d, 0, ds.view(np.uint32)), np.where(nan_o, 0, os_.view(np.uint32)))
