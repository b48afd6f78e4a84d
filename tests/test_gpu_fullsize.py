"""Parity at BASELINE.json's full sizes (configs[1], configs[2]): the grids are too large for the
CPU oracle, so the HIP path is checked through
  * a random SAMPLE of voxels carved by the oracle (same loop, bit-exact),
  * fused + view-dropping kernel == per-view generic kernel on the WHOLE grid (bit-exact; compared on
    the device by vcy_state_equal, also at 2048^3),
  * idempotence of kMax carving (a second pass over the same views changes nothing),
  * mesh invariants of the extracted surface (closed 2-manifold, Euler characteristic of a
    sphere-like hull, unique edge keys, faces in range) and slab-sharded == single-context."""
import numpy as np
import pytest

import oracle_lib as O
from vacancy_amd import carver as vc
from vacancy_amd import dist as vdist
from vacancy_amd import synth
from vacancy_amd.capi import UpdateOption

pytestmark = pytest.mark.gpu


def _sample_ids(n, nsample, seed):
    """Voxel indices for the oracle cross-check: uniform random, the grid corners, and a second half
    concentrated where the decisions of the fast path matter -- a shell around the carved surface
    (radius 0.35 n) and the planes where 8x8x8 bricks meet."""
    rng = np.random.RandomState(seed)
    half = nsample // 2
    ix = rng.randint(0, n, (half, 3))
    ix[:8] = [[a, b, c] for a in (0, n - 1) for b in (0, n - 1) for c in (0, n - 1)]
    d = rng.normal(size=(nsample - half, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = 0.35 * n + rng.uniform(-4.0, 4.0, (nsample - half, 1))
    shell = np.clip(np.floor(d * r + n / 2.0).astype(np.int64), 0, n - 1)
    edge = rng.rand(nsample - half) < 0.25  # snap a quarter of them onto brick faces
    shell[edge, 0] = np.clip((shell[edge, 0] // 8) * 8 + rng.randint(-1, 1, edge.sum()), 0, n - 1)
    return np.concatenate([ix, shell])


def _sample_check(dev, opt, views, sdfs, n, nsample=2_000_000, seed=5):
    """Oracle carve of a sample of voxels against the device state, read back through
    vcy_download_voxels (no whole-grid download)."""
    ix = _sample_ids(n, nsample, seed)
    ax = O.axis_positions(-n / 2.0, n / 2.0, 1.0, n)
    pos = np.stack([ax[ix[:, 0]], ax[ix[:, 1]], ax[ix[:, 2]]], 1)
    orc = O.OracleGrid(opt, positions=pos)
    for v, s in zip(views, sdfs):
        orc.carve(v, s)
    os_, ou = orc.download()
    lin = (ix[:, 2].astype(np.int64) * n + ix[:, 1]) * n + ix[:, 0]
    ds, du = dev.download_voxels(lin)
    assert np.array_equal(du, ou)
    assert np.array_equal(ds.view(np.uint32), os_.view(np.uint32))


_sample_check_queries = _sample_check


def _mesh_invariants(m, expect_closed=True):
    f = m["faces"].astype(np.int64)
    nv = len(m["vertices"])
    assert f.min() >= 0 and f.max() < nv
    d = m["keys"][:, 1] - m["keys"][:, 0]
    step = np.unique(d)
    assert len(step) <= 3 and (d > 0).all()  # +x, +y, +z neighbours only
    code = m["keys"][:, 0] * 4 + np.searchsorted(step, d)
    assert len(np.unique(code)) == nv  # every cut edge appears once
    assert np.isfinite(m["vertices"]).all()
    if expect_closed:
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        e.sort(axis=1)
        code = e[:, 0] * nv + e[:, 1]
        uniq, counts = np.unique(code, return_counts=True)
        assert (counts == 2).all(), "surface is not a closed 2-manifold"
        assert nv - len(uniq) + len(f) == 2  # Euler characteristic of a sphere-like hull


def _oracle_slices_check(dev, mesh, n, za, uo=None, nslices=16):
    """Marching cubes at full size against the ORACLE: the carved state of slices [za, za + 16) is read back
    (vcy_download_voxels), loaded into an oracle grid of n x n x 16 voxels whose voxel centres are those slices',
    and the oracle's slab extraction of its layers 2..15 (slices 0, 1 as the halo, exactly what one rank of a
    sharded run computes) is compared with what the device's WHOLE-GRID extraction holds for those cell layers:
    the same faces in the same order (as triples of global edge keys), the same vertex bits, and the device's
    vertex numbering increasing in order of first reference.  This is where the 64-bit offsets, the plane layout
    and cell indices beyond 2^32 are exercised under an oracle's eyes."""
    sl = n * n
    ids = np.arange(za * sl, (za + nslices) * sl, dtype=np.int64)
    ds, du = dev.download_voxels(ids)
    h = n / 2.0
    from vacancy_amd.capi import CarverOption
    sub = CarverOption(bb_min=(-h, -h, za - h), bb_max=(h, h, za + nslices - h), resolution=1.0,
                       update_option=uo or UpdateOption())
    orc = O.OracleGrid(sub)
    assert orc.dims == (n, n, nslices)
    ax = O.axis_positions(-h, h, 1.0, n)
    zpos = orc.positions()[::sl, 2]
    assert np.array_equal(zpos.view(np.uint32), ax[za:za + nslices].view(np.uint32))  # same voxel centres
    orc.upload(ds, du)
    del ds, du
    om = O.marching_cubes_slab(orc, 2, nslices, 0.0, True)
    orc.close()
    assert len(om["faces"]) > 0, "the slices chosen hold no surface"
    okeys = om["keys"] + np.int64(za) * sl          # local voxel ids -> global
    otri = okeys[om["faces"]].reshape(-1, 6)        # a face as the edge keys of its three vertices
    dkeys, dfaces = mesh["keys"], mesh["faces"]
    # where the oracle's first face sits in the device's face array (faces are in scan order: one block)
    k0 = otri[0]
    cand = np.nonzero((dkeys[dfaces[:, 0], 0] == k0[0]) & (dkeys[dfaces[:, 0], 1] == k0[1]))[0]
    start = None
    for c in cand:
        if np.array_equal(dkeys[dfaces[c]].reshape(6), k0):
            start = int(c)
            break
    assert start is not None, "the oracle's first face of layer %d is not in the device mesh" % (za + 2)
    nf = len(otri)
    block = dfaces[start:start + nf]
    assert len(block) == nf
    assert np.array_equal(dkeys[block].reshape(-1, 6), otri), "faces of layers %d..%d differ" % (za + 2, za + nslices - 1)
    assert np.array_equal(mesh["vertices"][block].view(np.uint32), om["vertices"][om["faces"]].view(np.uint32))
    # the face before / after the block belongs to another layer (the block is ALL faces of these layers)
    zmax = lambda f: int(dkeys[f].max() // sl)  # noqa: E731
    if start > 0:
        assert zmax(dfaces[start - 1]) <= za + 1
    if start + nf < len(dfaces):
        assert dkeys[dfaces[start + nf]].max() // sl >= za + nslices - 1
    # numbering: the vertices these layers create are numbered in order of first reference
    nfor = int(om["n_foreign"])
    own = om["faces"] >= nfor                       # oracle: indices below n_foreign belong to the slab below
    flat_o, flat_d = om["faces"].reshape(-1), block.reshape(-1)
    first = np.unique(flat_o[own.reshape(-1)], return_index=True)[1]
    own_dev = flat_d[own.reshape(-1)][np.sort(first)]
    assert (np.diff(own_dev) == 1).all(), "device vertex numbers of these layers are not consecutive in scan order"


def test_config1_512_tsdf():
    """configs[1]: 512^3, 16 sphere silhouettes at 640x480, TSDF fusion on."""
    n, nv, w, h = 512, 16, 640, 480
    uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    sdfs = [vc.make_sdf(m, use_truncation=True, band=0.1) for m in masks]
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    d = [dev.upload_sdf(s) for s in sdfs]
    assert dev.CarveBatchDevice(views, d), vc.last_error()
    _sample_check(dev, opt, views, sdfs, n)
    # per-view generic kernel on the whole grid
    ref = vc.VoxelCarver(opt)
    assert ref.Init()
    ref.set_param("fused", 0)
    assert ref.CarveBatchDevice(views, [ref.upload_sdf(s) for s in sdfs])
    assert dev.state_diff(ref) == 0
    ref.close()
    m = dev.ExtractIsoSurface(0.0, True)
    assert len(m["faces"]) > 100000
    _mesh_invariants(m, expect_closed=False)  # truncation leaves untouched voxels: open patches allowed


def test_config2_1024_default():
    """configs[2]: 1024^3, 32 views at 1280x720, default (kMax, bilinear)."""
    n, nv, w, h = 1024, 32, 1280, 720
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, nv, w, h)
    sdf0 = vc.make_sdf(masks[0])
    sdfs = [sdf0] * nv
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    d0 = dev.upload_sdf(sdf0)
    assert dev.CarveBatchDevice(views, [d0] * nv), vc.last_error()
    _sample_check(dev, opt, views, sdfs, n)
    # whole grid: per-view generic kernel, no fusion, no dropping -- compared on the device
    ref = vc.VoxelCarver(opt)
    assert ref.Init()
    ref.set_param("fused", 0)
    assert ref.CarveBatchDevice(views, [ref.upload_sdf(sdf0)] * nv)
    assert dev.state_diff(ref) == 0
    # the comparison itself must see a difference: one more view on one side only
    assert ref.CarveBatchDevice(views[:1], [ref.upload_sdf(sdf0 * np.float32(1.5))])
    assert ref.state_diff(dev) > 0
    ref.close()
    # idempotence of kMax on the fused path (view dropping against a non-fresh state, brick minima in use):
    # a second pass over the same views changes nothing
    ref = vc.VoxelCarver(opt)
    assert ref.Init()
    assert ref.CarveBatchDevice(views, [ref.upload_sdf(sdf0)] * nv)
    assert ref.CarveBatchDevice(views, [ref.upload_sdf(sdf0)] * nv)
    assert dev.state_diff(ref) == 0
    ref.close()
    m = dev.ExtractIsoSurface(0.0, True)
    _mesh_invariants(m)
    # every vertex lies between the two voxel centres of its edge key
    ax = O.axis_positions(-n / 2.0, n / 2.0, 1.0, n)
    k0, k1 = m["keys"][:, 0], m["keys"][:, 1]
    p0 = np.stack([ax[k0 % n], ax[(k0 // n) % n], ax[k0 // (n * n)]], 1)
    p1 = np.stack([ax[k1 % n], ax[(k1 // n) % n], ax[k1 // (n * n)]], 1)
    lo, hi = np.minimum(p0, p1), np.maximum(p0, p1)
    assert ((m["vertices"] >= lo) & (m["vertices"] <= hi)).all()
    # against the oracle, 16 slices at a time: inside the object, across z = nz / 2, and the top of the hull
    for za in (300, 504, 856):
        _oracle_slices_check(dev, m, n, za)
    # the same extraction without the brick minima (every brick read) is the same mesh
    dev.set_param("mcskip", 0)
    m0 = dev.ExtractIsoSurface(0.0, True)
    dev.set_param("mcskip", 1)
    assert np.array_equal(m0["faces"], m["faces"]) and np.array_equal(m0["keys"], m["keys"])
    assert np.array_equal(m0["vertices"].view(np.uint32), m["vertices"].view(np.uint32))
    del m0
    dev.close()
    # configs[3]: the same grid sharded by z-slab (2 contexts on this GPU) gives the same mesh
    parts, ranks = [], []
    for r in range(2):
        c = vc.VoxelCarver(opt, z_range=vdist.slab_range(n, r, 2))
        assert c.Init(), vc.last_error()
        assert c.CarveBatchDevice(views, [c.upload_sdf(sdf0)] * nv)
        ranks.append(c)
    info = vc.halo_allgather(ranks)  # the library's RCCL all-gather (one rank: both slabs are on this GPU)
    assert "op=ncclAllGather" in info and "ranks=1" in info
    for c in ranks:
        parts.append(c.ExtractIsoSurface(0.0, True))
        c.close()
    merged = vdist.merge_meshes(parts)
    assert np.array_equal(merged["faces"], m["faces"])
    assert np.array_equal(merged["keys"], m["keys"])
    assert np.array_equal(merged["vertices"].view(np.uint32), m["vertices"].view(np.uint32))


def test_config4_2048_64_views_streamed():
    """configs[4] on ONE GPU (the reference cannot represent this grid: 32-bit voxel ids): 2048^3,
    64 views at 1920x1080, silhouettes streamed (upload + device SDF overlapped with the fused carve)."""
    n, nv, w, h = 2048, 64, 1920, 1080
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, nv, w, h)
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    assert dev.dims == (n, n, n)
    assert dev.CarveBatchSilhouettes(views, masks), vc.last_error()
    sdf0 = O.make_sdf(masks[0])
    _sample_check_queries(dev, opt, views, [sdf0] * nv, n)
    # whole grid at 2048^3 (sub-half-pixel voxel footprints: where the `sure`-tile, short-division and
    # view-dropping decisions differ from the 1024^3 case): per-view generic kernel in a second context
    # (2 x 51.5 GB), compared on the device
    ref = vc.VoxelCarver(opt)
    assert ref.Init(), vc.last_error()
    ref.set_param("fused", 0)
    d0 = ref.upload_sdf(sdf0)
    assert ref.CarveBatchDevice(views, [d0] * nv), vc.last_error()
    assert dev.state_diff(ref) == 0
    ref.close()
    m = dev.ExtractIsoSurface(0.0, True)
    assert len(m["faces"]) > 15_000_000
    _mesh_invariants(m)
    # against the oracle: the slices around linear cell index 2^32 (z = 1024 at 2048^2 cells per layer)
    _oracle_slices_check(dev, m, n, 1016)
