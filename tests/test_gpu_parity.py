"""GPU parity tests: the HIP path (through the C-ABI, via the Python mirror of VoxelCarver)
against the CPU oracle on the same inputs.  Bit-exact: sdf bits, update_num, marching-cubes
vertex array, face array and edge keys (including their ORDER, which equals the reference's
serial scan)."""
import time
import numpy as np
import pytest

import bunny_data as B
import oracle_lib as O
from vacancy_amd import carver as vc
from vacancy_amd import synth
from vacancy_amd.capi import UpdateOption

pytestmark = pytest.mark.gpu


def assert_state_equal(dev, orc, ctx=""):
    ds, du = dev.download()
    os_, ou = orc.download()
    assert np.array_equal(du, ou), "%s update_num differs at %d voxels" % (ctx, int((du != ou).sum()))
    assert np.array_equal(ds.view(np.uint32), os_.view(np.uint32)), \
        "%s sdf bits differ at %d voxels" % (ctx, int((ds.view(np.uint32) != os_.view(np.uint32)).sum()))


def assert_mesh_equal(dm, om, ctx=""):
    assert dm["vertices"].shape == om["vertices"].shape, (ctx, dm["vertices"].shape, om["vertices"].shape)
    assert dm["faces"].shape == om["faces"].shape, (ctx, dm["faces"].shape, om["faces"].shape)
    assert np.array_equal(dm["keys"], om["keys"]), ctx + " edge keys / vertex order differ"
    assert np.array_equal(dm["faces"], om["faces"]), ctx + " faces differ"
    assert np.array_equal(dm["vertices"].view(np.uint32), om["vertices"].view(np.uint32)), \
        ctx + " vertex bits differ"


@pytest.mark.parametrize("mode", list(B.MODES))
def test_bunny_all_views(mode):
    """configs[0]: data/ bunny, 6 masks + tumpose, resolution 10 (54x53x42)."""
    uo = UpdateOption(**B.MODES[mode])
    opt = B.bunny_option(10.0, uo)
    views = B.bunny_views(lambda t, q: synth.affine_inverse(synth.pose_from_tum(t, q)))
    masks = B.load_masks()
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    orc = O.OracleGrid(opt)
    assert dev.dims == orc.dims
    assert np.array_equal(dev.positions().view(np.uint32), orc.positions().view(np.uint32))
    for i in range(6):
        ok, sdf = dev.CarveSilhouette(views[i], masks[i], return_sdf=True)
        assert ok, vc.last_error()
        osdf = O.make_sdf(masks[i], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
        assert np.array_equal(sdf.view(np.uint32), osdf.view(np.uint32))
        orc.carve(views[i], osdf)
        assert_state_equal(dev, orc, "%s view %d" % (mode, i))
        for interp in (True, False):
            assert_mesh_equal(dev.ExtractIsoSurface(0.0, interp), orc.marching_cubes(0.0, interp),
                              "%s view %d interp=%s" % (mode, i, interp))
    for inside_empty in (False, True):
        dv, ov = dev.ExtractVoxel(inside_empty), orc.extract_voxel(inside_empty)
        assert np.array_equal(dv["faces"], ov["faces"])
        assert np.array_equal(dv["vertices"].view(np.uint32), ov["vertices"].view(np.uint32))
    if mode == "default":
        m = dev.ExtractIsoSurface(0.0, True)
        assert (len(m["vertices"]), len(m["faces"])) == (8672, 17270)
        v = dev.ExtractVoxel(False)
        assert (len(v["vertices"]), len(v["faces"])) == (683400, 341700)


def test_bunny_fine_grid():
    opt = B.bunny_option(2.5)
    views = B.bunny_views(lambda t, q: synth.affine_inverse(synth.pose_from_tum(t, q)))
    masks = B.load_masks()
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    orc = O.OracleGrid(opt)
    for i in range(6):
        sdf = O.make_sdf(masks[i])
        assert dev.Carve(views[i], sdf)
        orc.carve(views[i], sdf)
    assert_state_equal(dev, orc, "res 2.5")
    dm, om = dev.ExtractIsoSurface(), orc.marching_cubes()
    assert (len(dm["vertices"]), len(dm["faces"])) == (144594, 288938)
    assert_mesh_equal(dm, om, "res 2.5")


SYN_MODES = {
    "default": dict(),
    "tsdf": dict(voxel_update=1, use_truncation=True, truncation_band=0.1),
    "tsdf_weight": dict(voxel_update=1, voxel_update_weight=0.37),
    "nn": dict(sdf_interp=0),
    "outside_max": dict(update_outside=1),
    "nn_outside_trunc": dict(sdf_interp=0, update_outside=1, use_truncation=True, truncation_band=0.3),
    "max_update_2": dict(voxel_max_update_num=2),
    "wa_max_update_300": dict(voxel_update=1, voxel_max_update_num=300),
    "max_update_70000": dict(voxel_max_update_num=70000),
}


@pytest.mark.parametrize("mode", list(SYN_MODES))
def test_synthetic_sphere_small(mode):
    n, nv, w, h = 48, 7, 160, 120
    uo = UpdateOption(**SYN_MODES[mode])
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    # shrink the ROI of some views so that voxels fall outside it
    views[2].roi_min[0], views[2].roi_min[1] = 40, 30
    views[2].roi_max[0], views[2].roi_max[1] = 120, 90
    views[5].roi_max[0] = 100
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    orc = O.OracleGrid(opt)
    for i in range(nv):
        rmin, rmax = tuple(views[i].roi_min), tuple(views[i].roi_max)
        sdf = O.make_sdf(masks[i], rmin, rmax, use_truncation=bool(uo.use_truncation),
                         band=uo.truncation_band)
        assert dev.Carve(views[i], sdf), vc.last_error()
        orc.carve(views[i], sdf)
        assert_state_equal(dev, orc, "%s view %d" % (mode, i))
    assert_mesh_equal(dev.ExtractIsoSurface(0.0, True), orc.marching_cubes(0.0, True), mode)
    assert_mesh_equal(dev.ExtractIsoSurface(0.25, True), orc.marching_cubes(0.25, True), mode + " iso .25")


def test_orthographic_camera():
    n = 40
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, 3, 96, 80)
    for v in views:
        v.is_ortho = 1
        # camera-space x,y are used as pixel coordinates: shift them into the image
        v.w2c[3] += 48.0
        v.w2c[7] += 40.0
    dev = vc.VoxelCarver(opt)
    assert dev.Init()
    orc = O.OracleGrid(opt)
    for i in range(3):
        sdf = O.make_sdf(masks[i])
        assert dev.Carve(views[i], sdf)
        orc.carve(views[i], sdf)
    assert_state_equal(dev, orc, "ortho")


@pytest.mark.parametrize("kw", [dict(), dict(voxel_update=1, use_truncation=True), dict(sdf_interp=0)])
def test_orthographic_views_in_the_select_free_loops(kw):
    """Orthographic cameras qualify for `sure` tiles too (no division; the only depth test is pc.z < 0): a grid
    of 96^3 unit voxels seen by 8 orthographic views of 200 x 190 pixels (one voxel = one pixel), in one fused
    launch, with view dropping on and off and with both tile kinds, against the oracle."""
    n, nv, w, h = 96, 8, 200, 190
    uo = UpdateOption(**kw)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    for v in views:
        v.is_ortho = 1
        v.w2c[3] += np.float32(w / 2)   # camera-space x, y are the pixel coordinates: centre the grid in the image
        v.w2c[7] += np.float32(h / 2)
    sdfs = [vc.make_sdf(m, use_truncation=bool(uo.use_truncation), band=uo.truncation_band) for m in masks]
    orc = O.OracleGrid(opt)
    for i in range(nv):
        orc.carve(views[i], sdfs[i])
    os_, ou = orc.download()
    assert int((ou > 0).sum()) > n ** 3 // 2
    for cull, tile in ((1, 0), (0, 1), (1, 2)):
        dev = vc.VoxelCarver(opt)
        assert dev.Init()
        dev.set_param("cull", cull)
        dev.set_param("tile", tile)
        devs = [dev.upload_sdf(s_) for s_ in sdfs]
        assert dev.CarveBatchDevice(views, devs), vc.last_error()
        ds, du = dev.download()
        assert np.array_equal(du, ou), (kw, cull, tile, int((du != ou).sum()))
        assert np.array_equal(ds.view(np.uint32), os_.view(np.uint32)), (kw, cull, tile)


@pytest.mark.parametrize("kw,ortho,lazy", [(dict(sdf_interp=0), False, 1), (dict(), True, 1),
                                            (dict(voxel_update=1, use_truncation=True, sdf_interp=0), True, 1),
                                            (dict(voxel_update=1, voxel_update_weight=0.5, use_truncation=True), False, 1),
                                            (dict(voxel_update=1, use_truncation=True, truncation_band=0.1), False, 0),
                                            (dict(update_outside=1), False, 0)])
@pytest.mark.parametrize("oneview", [1, 0])
def test_one_view_instance_in_every_flavour(kw, ortho, lazy, oneview):
    """The kernel instance compiled for ONE view ("oneview" 1: footprint record in registers, no view loop) and the general
    instance, one launch per view ("defer" 0), in the flavours the benchmark modes do not reach: nearest-neighbour taps
    and orthographic cameras (the GEN instances), general weights, two-byte counters from the start ("lazycount" 0),
    `update_outside = kMax`; state against the oracle after every view, mesh at the end.  Images of 160 x 120: raw tiles,
    footprint records, brick minima, live list and cooperative write-back all take part."""
    n, nv, w, h = 72, 9, 160, 120
    uo = UpdateOption(**kw)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    if ortho:
        for v in views:
            v.is_ortho = 1
            v.w2c[3] += np.float32(w / 2)
            v.w2c[7] += np.float32(h / 2)
    base = O.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
    rng = np.random.RandomState(3)
    noisy = (base + rng.uniform(-0.04, 0.04, base.shape)).astype(np.float32)
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    dev.set_param("lazycount", lazy)
    dev.set_param("defer", 0)
    dev.set_param("oneview", oneview)
    orc = O.OracleGrid(opt)
    for i in range(nv):
        img = noisy if i % 3 == 2 else base
        assert dev.Carve(views[i], img), vc.last_error()
        orc.carve(views[i], img)
        assert_state_equal(dev, orc, "%s ortho %s oneview %d view %d" % (kw, ortho, oneview, i))
    assert_mesh_equal(dev.ExtractIsoSurface(0.0, True), orc.marching_cubes(0.0, True), "%s oneview %d" % (kw, oneview))


def test_camera_inside_grid_and_behind():
    """voxels behind the camera (pc.z < 0), at pc.z == 0 and right at the camera."""
    n = 32
    opt = synth.sphere_option(n, UpdateOption(update_outside=1))
    w2c = np.array([[1, 0, 0, 0.5], [0, 1, 0, 0.5], [0, 0, 1, 0.5]], np.float32)  # camera at a voxel centre
    view = vc.make_view(w2c, np.float32(50), np.float32(50), np.float32(63.5), np.float32(47.5), 128, 96)
    rng = np.random.RandomState(7)
    sdf = rng.uniform(-1, 1, (96, 128)).astype(np.float32)
    dev = vc.VoxelCarver(opt)
    assert dev.Init()
    orc = O.OracleGrid(opt)
    assert dev.Carve(view, sdf)
    orc.carve(view, sdf)
    assert_state_equal(dev, orc, "camera inside")


def test_batch_equals_sequential():
    n, nv, w, h = 40, 6, 128, 96
    for kw in (dict(), dict(voxel_update=1, use_truncation=True)):
        uo = UpdateOption(**kw)
        opt = synth.sphere_option(n, uo)
        views, masks = synth.sphere_views(n, nv, w, h)
        sdfs = [vc.make_sdf(m, use_truncation=bool(uo.use_truncation), band=uo.truncation_band) for m in masks]
        a = vc.VoxelCarver(opt)
        b = vc.VoxelCarver(opt)
        assert a.Init() and b.Init()
        devs = [a.upload_sdf(s) for s in sdfs]
        for i in range(nv):
            assert b.Carve(views[i], sdfs[i])
        assert a.CarveBatchDevice(views, devs), vc.last_error()
        sa, ua = a.download()
        sb, ub = b.download()
        assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)) and np.array_equal(ua, ub)
        for d in devs:
            a.free_device(d)


@pytest.mark.parametrize("kw", [dict(voxel_update=1, use_truncation=True), dict(voxel_update=1),
                                dict(voxel_update=1, use_truncation=True, truncation_band=0.02), dict(),
                                dict(voxel_update=1, voxel_update_weight=0.37, use_truncation=True),
                                dict(voxel_update=1, voxel_update_weight=2.25), dict(sdf_interp=0),
                                dict(sdf_interp=0, voxel_update=1, use_truncation=True), dict(update_outside=1)])
def test_select_free_loops_at_benchmark_footprints(kw):
    """The loops the benchmark runs in -- raw 16 x 16 tiles, first-touch stores, the update chains, the
    truncation test compiled out where the tile's lower bound allows it, brick-wide weights of the average
    while all counts of a brick agree (unit and general weights), nearest-neighbour taps, update_outside = kMax
    with its bound -- on a scene with the benchmark's sub-pixel voxel footprints
    (160^3 in 200 x 150 images), in two launches (the second starts from a carved, non-fresh state) and
    after an upload (update_num no longer implied by sdf), with view dropping on and off and with the quad
    tile instead of the raw one: state bit-identical to the oracle every time."""
    n, nv, w, h = 160, 16, 200, 150
    uo = UpdateOption(**kw)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    sdfs = [vc.make_sdf(m, use_truncation=bool(uo.use_truncation), band=uo.truncation_band) for m in masks]
    sdfs[5] = np.ascontiguousarray(np.round(sdfs[5] * 8) / 8).astype(np.float32)  # plateaus: ties with the state
    orc = O.OracleGrid(opt)
    half = []
    for i in range(nv):
        orc.carve(views[i], sdfs[i])
        if i == nv // 2 - 1:
            half = orc.download()
    os_, ou = orc.download()
    for cull, tile, reupload, prologue in ((1, 0, False, 0), (0, 0, False, 0), (1, 2, False, 0), (1, 0, True, 0),
                                           (1, 0, False, 1), (0, 0, True, 1)):
        dev = vc.VoxelCarver(opt)
        assert dev.Init()
        dev.set_param("cull", cull)
        dev.set_param("tile", tile)
        dev.set_param("prologue", prologue)  # 1: footprints in the carve kernel's prologue instead of the pre-pass's records
        devs = [dev.upload_sdf(s_) for s_ in sdfs]
        assert dev.CarveBatchDevice(views[:nv // 2], devs[:nv // 2]), vc.last_error()
        if reupload:
            hs, hu = dev.download()
            assert np.array_equal(hu, half[1]) and np.array_equal(hs.view(np.uint32), half[0].view(np.uint32))
            dev.upload(hs, hu)
        assert dev.CarveBatchDevice(views[nv // 2:], devs[nv // 2:]), vc.last_error()
        ds, du = dev.download()
        for d in devs:
            dev.free_device(d)
        assert np.array_equal(du, ou), (kw, cull, tile, reupload, prologue, int((du != ou).sum()))
        assert np.array_equal(ds.view(np.uint32), os_.view(np.uint32)), (kw, cull, tile, reupload, prologue)


@pytest.mark.parametrize("kw", [dict(), dict(use_truncation=True, truncation_band=0.2),
                                dict(update_outside=1), dict(voxel_update=1, use_truncation=True)])
def test_view_dropping_is_exact(kw):
    """The fused kernel drops (wave brick, view) pairs that provably cannot change the brick.
    20 views in one fused call with dropping on == off == per-view kernel == oracle, on smooth
    silhouette SDFs and on adversarial ones (white noise, plateaus equal to the running max,
    NaN / inf / lowest() pixels)."""
    n, nv, w, h = 56, 20, 200, 150
    uo = UpdateOption(**kw)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    rng = np.random.RandomState(11)
    sdfs = []
    for i in range(nv):
        base = vc.make_sdf(masks[i], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
        kind = i % 5
        if kind == 1:
            base = rng.uniform(-1.5, 1.0, base.shape).astype(np.float32)
        elif kind == 2:
            base = np.round(base * 4) / 4  # plateaus: many exact ties with the running max
            base = base.astype(np.float32)
        elif kind == 3:
            base = base.copy()
            base[rng.rand(*base.shape) < 0.01] = np.nan
            base[rng.rand(*base.shape) < 0.01] = np.inf
            base[rng.rand(*base.shape) < 0.01] = np.finfo(np.float32).min
            base[rng.rand(*base.shape) < 0.01] = -np.inf
        sdfs.append(np.ascontiguousarray(base))
    views[4].roi_min[0], views[4].roi_max[1] = 30, 120
    orc = O.OracleGrid(opt)
    for i in range(nv):
        orc.carve(views[i], sdfs[i])
    os_, ou = orc.download()
    for fused, cull, tile, prologue in ((1, 1, 0, 0), (1, 1, 2, 0), (1, 0, 1, 0), (0, 0, 0, 0), (1, 1, 1, 1)):
        dev = vc.VoxelCarver(opt)
        assert dev.Init()
        dev.set_param("fused", fused)
        dev.set_param("cull", cull)
        dev.set_param("tile", tile)
        dev.set_param("prologue", prologue)
        devs = [dev.upload_sdf(s) for s in sdfs]
        assert dev.CarveBatchDevice(views, devs), vc.last_error()
        ds, du = dev.download()
        for d in devs:
            dev.free_device(d)
        assert np.array_equal(du, ou), (kw, fused, cull, tile, int((du != ou).sum()))
        # NaN voxels (only the NaN/inf-poisoned SDFs produce them) must be NaN on both sides;
        # their sign/payload is not defined by IEEE-754 and differs between x86 and gfx950
        # for generated NaNs (inf - inf).  Everything else is compared bit for bit.
        nan_d, nan_o = np.isnan(ds), np.isnan(os_)
        assert np.array_equal(nan_d, nan_o), (kw, fused, cull)
        bits_d = np.where(nan_d, 0, ds.view(np.uint32))
        bits_o = np.where(nan_o, 0, os_.view(np.uint32))
        assert np.array_equal(bits_d, bits_o), (kw, fused, cull, int((bits_d != bits_o).sum()))


def test_marching_cubes_random_state():
    """MC alone on arbitrary state: random sdf, invalid holes, untouched voxels, values within
    the 1e-5 snap band of the iso level, non-cubic dims."""
    rng = np.random.RandomState(3)
    opt = vc.CarverOption(bb_min=(0, 0, 0), bb_max=(37, 23, 29), resolution=1.0)
    dev = vc.VoxelCarver(opt)
    assert dev.Init()
    orc = O.OracleGrid(opt)
    assert dev.dims == orc.dims == (37, 23, 29)
    nvox = orc.n
    sdf = rng.uniform(-1, 1, nvox).astype(np.float32)
    sdf[rng.rand(nvox) < 0.05] = np.finfo(np.float32).min
    snap = rng.rand(nvox) < 0.05
    sdf[snap] = (rng.uniform(-2e-5, 2e-5, int(snap.sum()))).astype(np.float32)
    cnt = rng.randint(0, 4, nvox).astype(np.int32)
    dev.upload(sdf, cnt)
    orc.upload(sdf, cnt)
    for iso in (0.0, 0.3):
        for interp in (True, False):
            assert_mesh_equal(dev.ExtractIsoSurface(iso, interp), orc.marching_cubes(iso, interp),
                              "random iso=%s interp=%s" % (iso, interp))


@pytest.mark.parametrize("dims", [(64, 23, 29), (128, 70, 37), (256, 67, 12), (512, 40, 19), (1024, 70, 11),
                                  (2048, 35, 4)])
def test_marching_cubes_one_sweep_and_bit_plane_paths(dims):
    """Rows of 1, 2, 4 ... 32 whole 64-voxel words take the one-sweep cell search (mc_sweep_kernel: bit planes in
    LDS, several row groups and z chunks per grid at these sizes); "mcsweep" 0 sends the same state through the
    bit planes in memory.  Both against the oracle on uploaded state (update_num is read) and on carved state
    (it is not), for an iso level that is a float and one that is not."""
    rng = np.random.RandomState(dims[0] + dims[2])
    half = tuple(0.5 * d for d in dims)
    opt = vc.CarverOption(bb_min=tuple(-v for v in half), bb_max=half, resolution=1.0)
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    orc = O.OracleGrid(opt)
    assert dev.dims == orc.dims == dims
    # carved state: a few views of a ball around the origin that cuts through the box
    nv, w, h = 3, 160, 120
    views, masks = synth.sphere_views(max(dims), nv, w, h)
    for i in range(nv):
        sdf = vc.make_sdf(masks[i])
        assert dev.Carve(views[i], sdf)
        orc.carve(views[i], sdf)
    for sweep in (1, 0):
        dev.set_param("mcsweep", sweep)
        assert_mesh_equal(dev.ExtractIsoSurface(0.0, True), orc.marching_cubes(0.0, True), "carved sweep=%d" % sweep)
    # arbitrary state
    nvox = orc.n
    sdf = rng.uniform(-1, 1, nvox).astype(np.float32)
    smooth = rng.rand(nvox) < 0.7  # long runs of equal sign as well as noise
    x = np.arange(nvox) % dims[0]
    sdf[smooth] = (np.sin(x * 0.05) * 0.9 + 0.05).astype(np.float32)[smooth]
    sdf[rng.rand(nvox) < 0.02] = np.finfo(np.float32).min
    cnt = (rng.rand(nvox) < 0.97).astype(np.int32)
    dev.upload(sdf, cnt)
    orc.upload(sdf, cnt)
    for iso, interp in ((0.0, True), (0.3, False)):
        ref = orc.marching_cubes(iso, interp)
        assert len(ref["faces"]) > 0
        for sweep in (1, 0):
            dev.set_param("mcsweep", sweep)
            assert_mesh_equal(dev.ExtractIsoSurface(iso, interp), ref, "uploaded sweep=%d iso=%s" % (sweep, iso))


def test_mesh_without_edge_keys():
    """"meshkeys" 0: vcy_mesh.edge_keys stays NULL, vertices and faces are the same arrays."""
    n, nv = 48, 4
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, nv, 128, 96)
    dev = vc.VoxelCarver(opt)
    assert dev.Init()
    for i in range(nv):
        assert dev.Carve(views[i], vc.make_sdf(masks[i]))
    full = dev.ExtractIsoSurface(0.0, True)
    assert len(full["keys"]) == len(full["vertices"]) > 0
    dev.set_param("meshkeys", 0)
    assert dev.get_param("meshkeys") == 0
    bare = dev.ExtractIsoSurface(0.0, True)
    assert len(bare["keys"]) == 0
    assert np.array_equal(bare["faces"], full["faces"])
    assert np.array_equal(bare["vertices"].view(np.uint32), full["vertices"].view(np.uint32))
    dev.set_param("meshkeys", 1)
    assert_mesh_equal(dev.ExtractIsoSurface(0.0, True), full, "keys back on")


def test_extract_voxel_predicates_on_adversarial_state():
    """ExtractVoxel's keep predicates run on the device (extract_voxel.hip): both of them on uploaded state
    with zeros of either sign, denormals, products that underflow to -0 or to the smallest denormal,
    infinities, NaN, untouched voxels and invalid values, for every counter width and a grid whose rows are
    not multiples of 64; and on a fresh grid (nothing kept)."""
    rng = np.random.RandomState(11)
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, -1e-39, 1.1754944e-38, -1.1754944e-38, 1e-23, -1e-23,
                        2.0**-75, -(2.0**-75), 2.0**-74, -(2.0**-74), 2.0**-76, -(2.0**-76), 1.5 * 2.0**-75,
                        np.inf, -np.inf, np.nan, 1.0, -1.0, np.finfo(np.float32).min], dtype=np.float32)
    for max_update in (200, 255, 70000):  # u8, u16, u32 counters
        opt = vc.CarverOption(bb_min=(0, 0, 0), bb_max=(67, 9, 13), resolution=1.0,
                              update_option=UpdateOption(voxel_max_update_num=max_update))
        dev = vc.VoxelCarver(opt)
        assert dev.Init(), vc.last_error()
        orc = O.OracleGrid(opt)
        for inside_empty in (False, True):  # fresh: an empty mesh
            dv = dev.ExtractVoxel(inside_empty)
            assert len(dv["vertices"]) == 0 and len(dv["faces"]) == 0
        nvox = orc.n
        sdf = rng.uniform(-1, 1, nvox).astype(np.float32)
        pick = rng.rand(nvox) < 0.5
        sdf[pick] = special[rng.randint(0, len(special), int(pick.sum()))]
        cnt = rng.randint(0, 3, nvox).astype(np.int32)
        dev.upload(sdf, cnt)
        orc.upload(sdf, cnt)
        for inside_empty in (False, True):
            dv, ov = dev.ExtractVoxel(inside_empty), orc.extract_voxel(inside_empty)
            assert len(ov["faces"]) > 0
            assert np.array_equal(dv["faces"], ov["faces"]), (max_update, inside_empty)
            assert np.array_equal(dv["vertices"].view(np.uint32), ov["vertices"].view(np.uint32)), (max_update, inside_empty)
            # the same call writing into arrays the caller's callback returns (vcy_extract_voxel_into: the class API's path)
            iv = dev.ExtractVoxelInto(inside_empty)
            assert np.array_equal(iv["faces"], ov["faces"]), (max_update, inside_empty)
            assert np.array_equal(iv["vertices"].view(np.uint32), ov["vertices"].view(np.uint32)), (max_update, inside_empty)
    fresh = vc.VoxelCarver(opt)
    assert fresh.Init()
    assert len(fresh.ExtractVoxelInto(False)["vertices"]) == 0  # (nothing kept: the callback is never called)


def test_degenerate_grids_and_errors():
    # 1-voxel-thick grids have no cells: empty mesh, like the reference's loops from 1
    opt = vc.CarverOption(bb_min=(0, 0, 0), bb_max=(8, 8, 1), resolution=1.0)
    dev = vc.VoxelCarver(opt)
    assert dev.Init()
    m = dev.ExtractIsoSurface()
    assert len(m["vertices"]) == 0 and len(m["faces"]) == 0
    # invalid options -> Init() returns false (voxel_carver.cc:376-389, 278-287)
    for kw in (dict(voxel_max_update_num=0), dict(voxel_update_weight=0.0), dict(truncation_band=0.0)):
        assert not vc.VoxelCarver(B.bunny_option(10.0, UpdateOption(**kw))).Init()
    assert not vc.VoxelCarver(B.bunny_option(0.0)).Init()
    # Carve before Init (reference: UB; here: false)
    c = vc.VoxelCarver(B.bunny_option())
    assert not c.Carve(vc.make_view(np.eye(3, 4, dtype=np.float32), 1, 1, 0, 0, 4, 4), np.zeros((4, 4), np.float32))


def test_cpp_bunny_example(tmp_path):
    """configs[0] end to end through the C++ facade: examples/bunny.cc (the reference's
    examples.cc sequence) must print the reference's mesh sizes."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "vacancy_amd", "host"), "-s"], check=True)
    out = subprocess.run([os.path.join(root, "vacancy_amd", "host", "bunny"), B.BUNNY, str(tmp_path), "10", "3"],
                         check=True, capture_output=True, text=True).stdout
    # ShardedVoxelCarver (3 z-slabs, peer-copied halos, C++ merge) == single context, every view
    sh = [l.split() for l in out.splitlines() if l.startswith("SHARDED")]
    assert len(sh) == 6 and all(r[4] == "3" and r[6] == "1" for r in sh), sh
    assert [l for l in out.splitlines() if l.startswith("BOUNDS")] == ["BOUNDS 0 14 28 42"]
    # ShardedVoxelCarver::ExtractVoxel (device predicate per slab, one drifting cube on the host) == single context
    assert [l for l in out.splitlines() if l.startswith("VOXELSHARDED")] == ["VOXELSHARDED view %d identical 1" % i for i in range(6)]
    # the batch overload of ShardedVoxelCarver (one shared SDF producer for the three slabs) == single context
    assert [l for l in out.splitlines() if l.startswith("BATCHSHARDED")] == ["BATCHSHARDED slabs 3 verts 8672 identical 1"]
    # ... and with the cuts ShardedVoxelCarver::PlanPartition places for these six views (vcy_plan_z_slabs: whole
    # brick layers, here 42 slices = 6 layers into 3 slabs)
    out2 = subprocess.run([os.path.join(root, "vacancy_amd", "host", "bunny"), B.BUNNY, str(tmp_path), "10", "3", "planned"],
                          check=True, capture_output=True, text=True).stdout
    sh2 = [l.split() for l in out2.splitlines() if l.startswith("SHARDED")]
    assert len(sh2) == 6 and all(r[4] == "3" and r[6] == "1" for r in sh2), sh2
    b2 = [int(x) for x in [l for l in out2.splitlines() if l.startswith("BOUNDS")][0].split()[1:]]
    assert b2[0] == 0 and b2[-1] == 42 and len(b2) == 4 and all(z % 8 == 0 for z in b2[:-1]) and sorted(set(b2)) == b2, b2
    gold = json.load(open(os.path.join(root, "tests", "golden", "appendix_c.json")))
    rows = [l.split() for l in out.splitlines() if l.startswith("RESULT")]
    assert len(rows) == 6
    for i, r in enumerate(rows):
        exp = gold["modes"]["default"][i]
        assert (int(r[4]), int(r[6])) == (exp[5], exp[6]), (i, r)
    assert (int(rows[5][8]), int(rows[5][10])) == tuple(gold["final_nointerp"])
    assert int(rows[5][16]) == 683400  # ExtractVoxel, SURVEY Appendix C
    grid = [l.split() for l in out.splitlines() if l.startswith("GRID")]
    # batch overload + VoxelGrid read-back: touched / negative / sum(update_num) after the six views
    final = gold["modes"]["default"][5]
    assert [int(grid[0][2]), int(grid[0][4]), int(grid[0][6])] == [final[0], final[1], final[3]]
    vsum = [float(x) for x in rows[5][12:15]]
    assert np.allclose(vsum, gold["final_vertex_sums"], rtol=0, atol=1e-4)
    # ASCII PLY byte for byte what the reference's Mesh::WritePly (mesh.cc:583-631) writes for the
    # oracle's mesh: ostream default float formatting == "%g", "x y z \n", "3 a b c \n"
    views = B.bunny_views(lambda t, q: O.affine_inverse(O.pose_from_tum(t, q)))
    orc = O.OracleGrid(B.bunny_option(10.0))
    for i, m in enumerate(B.load_masks()):
        orc.carve(views[i], O.make_sdf(m))
    om = orc.marching_cubes(0.0, True)
    lines = ["ply", "format ascii 1.0", "element vertex %d" % len(om["vertices"]), "property float x",
             "property float y", "property float z", "element face %d" % len(om["faces"]),
             "property list uchar int vertex_indices", "end_header"]
    lines += ["%g %g %g " % (float(v[0]), float(v[1]), float(v[2])) for v in om["vertices"]]
    lines += ["3 %d %d %d " % (f[0], f[1], f[2]) for f in om["faces"]]
    expected = "\n".join(lines) + "\n"
    assert open(os.path.join(str(tmp_path), "surface_00005.ply")).read() == expected


def test_cpp_facade_refuses_camera_subclasses_it_cannot_project():
    """Camera::Project is virtual in the reference (camera.h:39-40, called per voxel at voxel_carver.cc:460); the
    device evaluates PinholeCamera and OrthoCamera.  A third subclass compiles against the facade (no extra pure
    virtual) and Carve() returns false for it -- single view and batch -- without touching the grid."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "vacancy_amd", "host"), "-s"], check=True)
    r = subprocess.run([os.path.join(root, "vacancy_amd", "host", "host_selftest"), B.BUNNY, "gpu"],
                       check=True, capture_output=True, text=True)
    row = [l.split() for l in r.stdout.splitlines() if l.startswith("CUSTOMCAM")][0]
    assert row[1:6] == ["1", "1", "0", "0", "1"], row
    assert int(row[6]) > 0  # the pinhole and orthographic views were applied
    assert "unsupported Camera subclass" in (r.stdout + r.stderr)
    assert (r.stdout + r.stderr).count("VoxelCarver::Carve main loop") >= 1  # the applied queue is logged as the reference's loop


@pytest.mark.parametrize("world,n", [(2, 44), (3, 44), (2, 64), (3, 64)])
def test_z_slab_sharding_on_one_gpu(world, n):
    """The multi-GPU path with every 'rank' as its own context on cuda:0: slab carve (no
    exchange), halo pack -> (host all-gather stand-in) -> unpack, per-slab extraction with the
    ghost layer, host merge == the single-context mesh == the oracle, array for array.  n = 64: rows of
    whole words, sent through the one-sweep cell search (ghost layer, halo slice whose update_num is read)."""
    from vacancy_amd import dist as vdist
    nv, w, h = 5, 128, 96
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, nv, w, h)
    sdfs = [vc.make_sdf(m) for m in masks]
    whole = vc.VoxelCarver(opt)
    assert whole.Init()
    orc = O.OracleGrid(opt)
    for i in range(nv):
        assert whole.Carve(views[i], sdfs[i])
        orc.carve(views[i], sdfs[i])
    ranks = []
    for r in range(world):
        c = vc.VoxelCarver(opt, z_range=vdist.slab_range(n, r, world))
        assert c.Init(), vc.last_error()
        for i in range(nv):
            assert c.Carve(views[i], sdfs[i])
        ranks.append(c)
    if n == 64:  # (the sweep is taken on request)
        for c in ranks + [whole]:
            c.set_param("mcsweep", 1)
    # slab states tile the whole grid
    ws, wu = whole.download()
    assert np.array_equal(np.concatenate([c.download()[0] for c in ranks]).view(np.uint32), ws.view(np.uint32))
    assert np.array_equal(np.concatenate([c.download()[1] for c in ranks]), wu)
    # extraction before the halo is installed must fail loudly on non-first slabs
    with pytest.raises(RuntimeError):
        ranks[1].ExtractIsoSurface()
    if world == 2:
        gathered = np.concatenate([c.halo_pack_host() for c in ranks])
        for r, c in enumerate(ranks):
            c.halo_unpack_host(gathered, r, world)
    else:
        vdist.exchange_halo(ranks, 0, 1)  # all slabs in this process: vcy_halo_install path
    for iso, interp in ((0.0, True), (0.1, False)):
        parts = [c.ExtractIsoSurface(iso, interp) for c in ranks]
        for r, (p, c) in enumerate(zip(parts, ranks)):
            ref = O.marching_cubes_slab(orc, c.z_range[0], c.z_range[1], iso, interp)
            assert p["n_foreign"] == ref["n_foreign"]
            assert_mesh_equal(p, ref, "slab %d/%d" % (r, world))
        merged = vdist.merge_meshes(parts)
        assert_mesh_equal(merged, whole.ExtractIsoSurface(iso, interp), "merged vs single context")
        assert_mesh_equal(merged, orc.marching_cubes(iso, interp), "merged vs oracle")


def test_native_rccl_halo_allgather_equals_host_exchange():
    """vcy_halo_allgather (ncclCommInitAll + ONE ncclAllGather, one communicator rank per device; here
    4 slabs on cuda:0 share one rank) installs the same halo slices as packing through the host."""
    from vacancy_amd import dist as vdist
    n, nv, w, h = 40, 4, 128, 96
    opt = synth.sphere_option(n, UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1))
    views, masks = synth.sphere_views(n, nv, w, h)
    world = 4

    def slabs():
        out = []
        for r in range(world):
            c = vc.VoxelCarver(opt, z_range=vdist.slab_range(n, r, world))
            assert c.Init(), vc.last_error()
            for v, m in zip(views, masks):
                assert c.CarveSilhouette(v, m)  # queued: the exchange must apply them first
            out.append(c)
        return out

    a, b = slabs(), slabs()
    info = vc.halo_allgather(a)
    assert "backend=rccl" in info and "op=ncclAllGather" in info and "ranks=1" in info
    assert "bytes_per_rank=%d" % (world * 2 * n * n * 6) in info
    gathered = np.concatenate([c.halo_pack_host() for c in b])
    for r, c in enumerate(b):
        c.halo_unpack_host(gathered, r, world)
    for ca, cb in zip(a, b):
        ma, mb = ca.ExtractIsoSurface(0.0, True), cb.ExtractIsoSurface(0.0, True)
        assert ma["n_foreign"] == mb["n_foreign"]
        assert_mesh_equal(ma, mb, "rccl vs host halo")
    # slabs that do not tile z in order are refused
    with pytest.raises(RuntimeError):
        vc.halo_allgather([a[0], a[2]])
    # a second exchange reuses the cached communicator
    calls = lambda text: int([kv for kv in text.split() if kv.startswith("calls=")][0][6:])
    assert calls(vc.halo_allgather(a)) == calls(info) + 1
    # vcy_halo_shutdown releases communicators, streams and staging; the next exchange builds them again and
    # installs the same halos (halo_exchange: the record bench.py prints, with what an all-gather moves)
    from vacancy_amd import capi
    capi.load().vcy_halo_shutdown()
    rec = vc.halo_exchange(a)
    assert rec["backend"].startswith("rccl") and rec["ranks"] == 1 and rec["slabs"] == world
    assert rec["bytes_received_per_rank"] == rec["bytes_per_rank"] * rec["ranks"]
    assert rec["bytes_needed_per_slab"] == 2 * n * n * 6
    for ca, cb in zip(a, b):
        assert_mesh_equal(ca.ExtractIsoSurface(0.0, True), cb.ExtractIsoSurface(0.0, True), "after shutdown")


def test_native_process_per_gpu_exchange_with_one_rank():
    """vcy_comm_create + vcy_halo_allgather_ranks: the halo all-gather of a C++ host that runs one process per GPU
    (ncclGetUniqueId -> rendezvous -> ncclCommInitRank; no torch).  With the one GPU of this box: ONE rank holding three
    slabs of the grid -- a real ncclCommInitRank communicator and a real ncclAllGather carrying the three packs -- and the
    slabs' meshes, merged by edge key, equal the single-context mesh.  (More than one rank needs more than one GPU.)"""
    import ctypes as C
    from vacancy_amd import dist as vdist
    lib = vc.capi.load()
    n, nv = 44, 5
    opt = synth.sphere_option(n, UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1))
    views, masks = synth.sphere_views(n, nv, 160, 120)
    slabs = []
    for r in range(3):
        c = vc.VoxelCarver(opt, z_range=vdist.slab_range(n, r, 3))
        assert c.Init(), vc.last_error()
        for i in range(nv):
            assert c.CarveSilhouette(views[i], masks[i])
        slabs.append(c)
    whole = vc.VoxelCarver(opt)
    assert whole.Init()
    for i in range(nv):
        assert whole.CarveSilhouette(views[i], masks[i])
    comm = C.c_void_p()
    assert lib.vcy_comm_create(0, 1, 0, None, 1000, C.byref(comm)) == 0, vc.last_error()
    arr = (C.c_void_p * 3)(*[c.ctx for c in slabs])
    assert lib.vcy_halo_allgather_ranks(comm, arr, 3) == 0, vc.last_error()
    text = lib.vcy_last_collective().decode()
    assert "op=ncclAllGather" in text and "ranks=1" in text, text
    meshes = [c.ExtractIsoSurface(0.0, True) for c in slabs]
    assert_mesh_equal(vdist.merge_meshes(meshes), whole.ExtractIsoSurface(0.0, True), "3 slabs, one rank")
    # a second exchange reuses the staging; a slab from another device set is refused
    assert lib.vcy_halo_allgather_ranks(comm, arr, 3) == 0
    assert lib.vcy_halo_allgather_ranks(comm, arr, 0) == vc.capi.VCY_ERR_INVALID_ARG
    lib.vcy_comm_destroy(comm)
    lib.vcy_comm_destroy(None)


def test_torch_shares_device_memory_with_the_library():
    """bench.py's multi-GPU halo exchange hands torch CUDA tensors to the C-ABI: the library and
    torch must sit on the same HIP runtime in one process."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    n = 24
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, 2, 64, 48)
    c = vc.VoxelCarver(opt)
    assert c.Init(), vc.last_error()
    for v, m in zip(views, masks):
        assert c.CarveSilhouette(v, m)
    nbytes = int(c._lib.vcy_halo_bytes(c.ctx))
    t = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert c._lib.vcy_halo_pack(c.ctx, C.c_void_p(t.data_ptr())) == 0, vc.last_error()
    c.sync()
    assert np.array_equal(t.cpu().numpy(), c.halo_pack_host())


def test_device_sdf_builder_equals_oracle():
    """MakeSignedDistanceField on the device (row f1): silhouettes, noise, all-255 / all-0 masks,
    sub-ROIs, odd sizes, every normalise / truncate combination -- bit-exact."""
    opt = synth.sphere_option(16)
    dev = vc.VoxelCarver(opt)
    assert dev.Init()
    rng = np.random.RandomState(2)
    masks = B.load_masks()[:2]
    noise = (rng.rand(97, 131) < 0.4).astype(np.uint8) * 255
    noise[rng.rand(97, 131) < 0.1] = 128
    sparse = np.full((333, 1283), 255, np.uint8)
    sparse[rng.randint(0, 333, 5), rng.randint(0, 1283, 5)] = 0
    cases = [(m, None, None) for m in masks] + [
        (masks[0], (10, 20), (300, 200)), (noise, None, None), (noise, (5, 7), (100, 60)),
        (np.full((20, 30), 255, np.uint8), None, None), (np.zeros((20, 30), np.uint8), None, None),
        (sparse, None, None), (sparse, (1, 1), (1281, 331)), (synth.sphere_views(64, 1, 1280, 720)[1][0], None, None)]
    for mask, rmin, rmax in cases:
        for norm in (True, False):
            for trunc in (False, True):
                d = dev.make_sdf_device(mask, rmin, rmax, norm, trunc, 0.1)
                got = dev.download_image(d, mask.shape)
                dev.free_device(d)
                ref = O.make_sdf(mask, rmin, rmax, norm, trunc, 0.1)
                assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (mask.shape, rmin, norm, trunc)


def test_streamed_silhouette_batch_equals_per_view_calls():
    """vcy_carve_batch_silhouettes (upload + device SDF on a second stream, fused carve in chunks of
    32 views; BASELINE config 5 shape) == one vcy_carve_silhouette per view, for 70 views (3 chunks)."""
    n, nv, w, h = 40, 70, 96, 72
    for kw in (dict(), dict(voxel_update=1, use_truncation=True)):
        opt = synth.sphere_option(n, UpdateOption(**kw))
        views, masks = synth.sphere_views(n, nv, w, h)
        rng = np.random.RandomState(4)
        masks = [np.where(rng.rand(h, w) < 0.02, 0, m).astype(np.uint8) for m in masks]  # distinct images
        a, b = vc.VoxelCarver(opt), vc.VoxelCarver(opt)
        assert a.Init() and b.Init()
        assert a.CarveBatchSilhouettes(views, masks), vc.last_error()
        for v, m in zip(views, masks):
            assert b.CarveSilhouette(v, m)
        sa, ua = a.download()
        sb, ub = b.download()
        assert np.array_equal(ua, ub) and np.array_equal(sa.view(np.uint32), sb.view(np.uint32))
        ids = rng.randint(0, n ** 3, 1000)
        qs, qu = a.download_voxels(ids)
        assert np.array_equal(qs.view(np.uint32), sa[ids].view(np.uint32)) and np.array_equal(qu, ua[ids])


@pytest.mark.parametrize("res", [10.0, 3.0])
def test_bunny_scale_big_tiles(res):
    """Voxels of >= 1 pixel (the reference's own demo scale): the fused kernel with the big LDS tile,
    the small one (mostly the generic in-kernel path) and the per-view kernel all equal the oracle."""
    opt = B.bunny_option(res)
    views = B.bunny_views(lambda t, q: synth.affine_inverse(synth.pose_from_tum(t, q)))
    masks = B.load_masks()
    sdfs = [O.make_sdf(m) for m in masks]
    orc = O.OracleGrid(opt)
    for v, s in zip(views, sdfs):
        orc.carve(v, s)
    os_, ou = orc.download()
    for tile in (0, 1, 2):
        dev = vc.VoxelCarver(opt)
        assert dev.Init()
        dev.set_param("tile", tile)
        devs = [dev.upload_sdf(s) for s in sdfs]
        assert dev.CarveBatchDevice(views, devs), vc.last_error()
        ds, du = dev.download()
        assert np.array_equal(du, ou) and np.array_equal(ds.view(np.uint32), os_.view(np.uint32)), (res, tile)


@pytest.mark.parametrize("dims", [(1, 1, 5), (2, 3, 1), (9, 1, 1), (1, 7, 2), (3, 3, 3), (65, 2, 9), (8, 8, 8), (33, 9, 17)])
def test_odd_grid_shapes(dims):
    """Grids thinner than a wave brick, single-voxel axes, dims straddling brick / word boundaries;
    a 1x1 and a 2x1 image; one camera looking away from the grid (nothing is carved)."""
    nx, ny, nz = dims
    opt = vc.CarverOption(bb_min=(-nx / 2.0, -ny / 2.0, -nz / 2.0), bb_max=(nx / 2.0, ny / 2.0, nz / 2.0),
                          resolution=1.0, update_option=UpdateOption(update_outside=int(nx % 2)))
    rng = np.random.RandomState(nx * 100 + ny * 10 + nz)
    views, sdfs = [], []
    for (w, h, away) in ((40, 30, False), (1, 1, False), (2, 1, False), (64, 48, True), (31, 17, False)):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        pos = d * (max(dims) * 2.0 + 3.0)
        target = pos * 2.0 if away else np.zeros(3)
        w2c = synth.affine_inverse(synth.lookat_c2w(pos, target, (0.0, 1.0, 0.0) if abs(d[1]) < 0.9 else (1.0, 0.0, 0.0)))
        f = np.float32(max(w, h) * 1.2)
        views.append(vc.make_view(w2c.astype(np.float32), f, f, np.float32(w / 2 - 0.5), np.float32(h / 2 - 0.5), w, h))
        sdfs.append(rng.uniform(-1, 1, (h, w)).astype(np.float32))
    orc = O.OracleGrid(opt)
    assert orc.dims == dims
    for v, s_ in zip(views, sdfs):
        orc.carve(v, s_)
    os_, ou = orc.download()
    for fused in (1, 0):
        dev = vc.VoxelCarver(opt)
        assert dev.Init(), vc.last_error()
        assert dev.dims == dims
        dev.set_param("fused", fused)
        assert dev.CarveBatchDevice(views, [dev.upload_sdf(s_) for s_ in sdfs]), vc.last_error()
        ds, du = dev.download()
        assert np.array_equal(du, ou) and np.array_equal(ds.view(np.uint32), os_.view(np.uint32)), (dims, fused)
        assert_mesh_equal(dev.ExtractIsoSurface(0.0, True), orc.marching_cubes(0.0, True), str(dims))
        dv, ov = dev.ExtractVoxel(True), orc.extract_voxel(True)
        assert np.array_equal(dv["faces"], ov["faces"]) and np.array_equal(dv["vertices"].view(np.uint32), ov["vertices"].view(np.uint32))


@pytest.mark.gpu
def test_measured_bandwidth_probe():
    """vcy_measure_bandwidth: the measured denominator bench.py prints next to the 8 TB/s figure."""
    rd, cp = vc.measure_bandwidth(0, 1 << 28, 2)
    assert 500.0 < rd < 16000.0 and 500.0 < cp < 16000.0
    lib = vc.capi.load()
    assert lib.vcy_measure_bandwidth(0, 16, 1, None, None) != 0  # too small / no outputs: rejected


def test_selftest_and_unit_weight_average_many_views():
    """The unit-weight weighted average uses a two-instruction reciprocal of update_num + 1:
    vcy_selftest checks it against IEEE division for every count; here 300 views push the counts
    of a small grid through 1..300 (u16 counter) against the oracle."""
    n, nv = 20, 300
    uo = UpdateOption(voxel_update=1, voxel_max_update_num=400)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, 64, 48)
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    assert dev.selftest(), vc.last_error()
    orc = O.OracleGrid(opt)
    sdf = O.make_sdf(masks[0])
    d = dev.upload_sdf(sdf)
    assert dev.CarveBatchDevice(vc.VoxelCarver.prepare_batch(views, [d] * nv)), vc.last_error()
    for i in range(nv):
        orc.carve(views[i], sdf)
    assert_state_equal(dev, orc, "unit-weight average, 300 views")
    assert int(dev.download()[1].max()) == nv
    dev.free_device(d)


def test_per_view_calls_are_applied_together_and_in_order():
    """vcy_carve / vcy_carve_device only queue the view (private copy of the image); the queue is
    carved by one fused launch when the state is needed or 32 views wait.  Same results as carving
    each view at once ("defer" 0) and as the oracle, including a projection-model switch, a caller
    that overwrites its image between calls, and a reset that discards queued views."""
    n, nv = 28, 37
    uo = UpdateOption(voxel_update=1, voxel_max_update_num=300)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, 96, 72)
    for i in (5, 6, 20):  # orthographic views in between: the queue is flushed at each switch
        views[i].is_ortho = 1
        views[i].w2c[3] += 48.0
        views[i].w2c[7] += 36.0
    sdfs = [vc.make_sdf(m) * np.float32(1.0 + 0.01 * i) for i, m in enumerate(masks)]
    orc = O.OracleGrid(opt)
    for v, s in zip(views, sdfs):
        orc.carve(v, s)
    states = []
    for defer in (1, 0):
        dev = vc.VoxelCarver(opt)
        assert dev.Init(), vc.last_error()
        dev.set_param("defer", defer)
        # queued views die with a reset
        assert dev.Carve(views[3], sdfs[7])
        dev.reset()
        scratch = dev.upload_sdf(sdfs[0])
        for i, (v, s) in enumerate(zip(views, sdfs)):
            if i % 3 == 0:   # device image owned by the caller, overwritten right after the call
                dev.memcpy_h2d(scratch, s)
                assert dev.CarveDevice(v, scratch), vc.last_error()
                dev.memcpy_h2d(scratch, np.full_like(s, 123.0))
            else:
                assert dev.Carve(v, s), vc.last_error()
        states.append(dev.download())
        assert_state_equal(dev, orc, "per-view calls, defer=%d" % defer)
        dev.free_device(scratch)
        dev.close()
    assert np.array_equal(states[0][0].view(np.uint32), states[1][0].view(np.uint32))
    assert np.array_equal(states[0][1], states[1][1])


def test_queued_views_with_slabs_and_halo_exchange():
    """Two z-slab contexts, per-view Carve() calls (queued), halo exchange and extraction with no read
    of the state in between: packing flushes the lower slab, extraction the upper one, and the halo
    installed in between stays valid."""
    from vacancy_amd import dist as vdist
    n, nv = 36, 9
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, nv, 112, 84)
    sdfs = [vc.make_sdf(m) for m in masks]
    orc = O.OracleGrid(opt)
    for v, s in zip(views, sdfs):
        orc.carve(v, s)
    ranks = []
    for r in range(2):
        c = vc.VoxelCarver(opt, z_range=vdist.slab_range(n, r, 2))
        assert c.Init(), vc.last_error()
        ranks.append(c)
    for v, s in zip(views, sdfs):
        for c in ranks:
            assert c.Carve(v, s), vc.last_error()
    gathered = np.concatenate([c.halo_pack_host() for c in ranks])
    for r, c in enumerate(ranks):
        c.halo_unpack_host(gathered, r, 2)
    parts = [c.ExtractIsoSurface(0.0, True) for c in ranks]
    assert_mesh_equal(vdist.merge_meshes(parts), orc.marching_cubes(0.0, True), "queued views, two slabs")
    # a halo installed BEFORE more views are queued is stale: extraction must refuse
    assert ranks[1].Carve(views[0], sdfs[0])
    with pytest.raises(RuntimeError):
        ranks[1].ExtractIsoSurface()
    for c in ranks:
        c.close()


@pytest.mark.parametrize("mode", ["default", "tsdf"])
def test_batches_of_up_to_64_views_in_one_launch(mode):
    """vcy_carve_batch_device fuses up to 64 views per launch (one prologue lane per view): 50 and 64 + 7
    views against the oracle's per-view loop."""
    n = 24
    uo = UpdateOption(**SYN_MODES[mode])
    opt = synth.sphere_option(n, uo)
    for nv in (50, 71):
        views, masks = synth.sphere_views(n, nv, 80, 60)
        sdfs = [O.make_sdf(m, use_truncation=bool(uo.use_truncation), band=uo.truncation_band) * np.float32(1 + 0.003 * i)
                for i, m in enumerate(masks)]
        dev = vc.VoxelCarver(opt)
        assert dev.Init(), vc.last_error()
        orc = O.OracleGrid(opt)
        d = [dev.upload_sdf(s) for s in sdfs]
        assert dev.CarveBatchDevice(vc.VoxelCarver.prepare_batch(views, d)), vc.last_error()
        for v, s in zip(views, sdfs):
            orc.carve(v, s)
        assert_state_equal(dev, orc, "%s %d views" % (mode, nv))
        for p in d:
            dev.free_device(p)
        dev.close()


def test_short_division_is_only_used_when_verified():
    """fx / z in the fused kernel: the 4- / 6-instruction sequences are selected per focal length after an
    exhaustive device check; results equal the full sequence ("shortdiv" 0) and the oracle for focal
    lengths that pass either check and for ones that fall back."""
    n, nv = 32, 5
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, nv, 96, 72)
    sdfs = [O.make_sdf(m) for m in masks]
    levels = set()
    rng = np.random.RandomState(11)
    for trial in range(12):
        f = np.float32(rng.uniform(60, 140))
        for v in views:
            v.fx = v.fy = float(f)
        orc = O.OracleGrid(opt)
        for v, s in zip(views, sdfs):
            orc.carve(v, s)
        for short in (1, 0):
            dev = vc.VoxelCarver(opt)
            assert dev.Init(), vc.last_error()
            dev.set_param("shortdiv", short)
            d = [dev.upload_sdf(s) for s in sdfs]
            assert dev.CarveBatchDevice(views, d), vc.last_error()
            lvl = dev.get_param("div_level")
            assert (lvl == 0) if not short else lvl in (0, 1, 2)
            if short:
                levels.add(lvl)
            assert_state_equal(dev, orc, "f=%r shortdiv=%d level=%d" % (float(f), short, lvl))
            dev.close()
    assert 2 in levels  # the common case


def test_failed_flush_is_reported_once_to_the_carve_loop():
    """vacancy_hip.h, vcy_carve: a failure while applying queued views is returned by the call that applies
    them; only when that call was NOT a carve entry point (a download, an extraction) does the next carve
    call return it once more.  A carve entry point that already returned the failure must not make the
    following, valid view fail as well.  ("inject_carve_failure": test hook, the next application fails.)"""
    n, nv, w, h = 40, 40, 96, 80
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, nv, w, h)
    sdf = O.make_sdf(masks[0])
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    # (a) the flush happens inside a download: reported there AND by the next carve call, then cleared
    dev.set_param("inject_carve_failure", 1)
    assert dev.Carve(views[0], sdf)              # queued
    with pytest.raises(RuntimeError, match="injected"):
        dev.download()
    assert not dev.Carve(views[1], sdf)          # the loop's `if (!Carve())` sees it ...
    assert "earlier queued view failed" in vc.last_error()
    assert dev.Carve(views[1], sdf), vc.last_error()   # ... once
    dev.download()
    # (b) the flush happens inside a carve entry point (32 views waiting): that call returns the failure,
    # the next valid view is accepted and queued
    dev.reset()
    dev.set_param("inject_carve_failure", 1)
    for i in range(31):
        assert dev.Carve(views[i], sdf), vc.last_error()
    assert not dev.Carve(views[31], sdf)
    assert "injected" in vc.last_error()
    assert dev.Carve(views[32], sdf), vc.last_error()
    # (c) the same through the batch entry point with views still queued
    dev.reset()
    assert dev.Carve(views[0], sdf)
    dev.set_param("inject_carve_failure", 1)
    d = dev.upload_sdf(sdf)
    assert not dev.CarveBatchDevice(views[1:3], [d, d])
    assert dev.Carve(views[3], sdf), vc.last_error()
    dev.free_device(d)
    # and the state after all this is what the oracle gets from the views that were applied
    orc = O.OracleGrid(opt)
    orc.carve(views[3], sdf)
    assert_state_equal(dev, orc, "after injected failures")


_TSDF = dict(voxel_update=1, use_truncation=True, truncation_band=0.1)
_TRUNC = dict(use_truncation=True, truncation_band=0.1)


# (image 160 x 120: voxels of 0.72 px, the raw 16 x 16 tiles with footprint records, live list and cooperative
# write-back; 200 x 150: 0.9 px, the big tiles, which bound their footprints in the kernel and know none of those)
# rowkernel -1: launches of few views take the few-view flavour of the fused kernel (a wave walks the four bricks of a
# row segment, state through LDS-direct loads, whole-row-segment stores; nx = 72: the last segment of a row has ONE brick
# inside the grid); 0: the workgroup-per-block kernel with (coopstore) its cooperative write-back, as round 5 ran it.
@pytest.mark.parametrize("kw,livelist,recordbytes,coopstore,img,rowkernel",
                         [(dict(), 1, 0, -1, (160, 120), -1), (dict(), 0, 0, -1, (160, 120), -1), (dict(), 1, 2000, -1, (160, 120), -1),
                          (_TRUNC, 1, 0, -1, (160, 120), -1), (_TSDF, 1, 0, -1, (160, 120), -1), (_TSDF, 0, 2000, -1, (160, 120), -1),
                          (_TRUNC, 0, 2000, 1, (160, 120), -1),
                          (dict(), 1, 0, -1, (160, 120), 0), (_TSDF, 1, 0, -1, (160, 120), 0), (_TSDF, 1, 0, 0, (160, 120), 0),
                          (_TSDF, 0, 2000, 1, (160, 120), 0), (dict(), 1, 0, 1, (160, 120), 0), (_TRUNC, 0, 2000, 1, (160, 120), 0),
                          (dict(), 1, 2000, -1, (160, 120), 0), (_TRUNC, 1, 2000, -1, (160, 120), 0),
                          (dict(), 1, 0, -1, (200, 150), -1), (_TSDF, 1, 0, -1, (200, 150), -1)])
def test_single_view_launches_with_brick_minima(kw, livelist, recordbytes, coopstore, img, rowkernel):
    """The reference's call pattern (examples.cc:117-149): carve ONE view, extract, carve the next ... With
    `defer` 0 every call is a launch of its own; from the second on a wave whose view provably changes nothing
    (bound against the brick minimum the previous launch left, or below the truncation limit) returns without
    reading the state -- or, with the live list, is never started -- and marching cubes skips bricks whose minimum lies above the iso level.  State and mesh
    equal the oracle's after every view, on smooth and adversarial images; writes that bypass the fused kernel
    (vcy_upload, the per-view kernel) switch the minima off until the next fused launch has rebuilt them."""
    n, nv, (w, h) = 72, 14, img
    uo = UpdateOption(**kw)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    rng = np.random.RandomState(5)
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    dev.set_param("defer", 0)
    dev.set_param("livelist", livelist)  # 1: only the workgroups with a live (brick, view) pair are started
    dev.set_param("recordbytes", recordbytes)  # 2000: every launch in chunks of three brick layers (as 2048^3 x 64 is)
    # write-back of a workgroup's bricks through LDS in whole row segments: -1 = the library's rule (weighted average,
    # few views, carved grid), 0 never, 1 wherever the layout allows (nx = 72: the last workgroup of a row has one wave
    # inside the grid, the other three leave before the barrier)
    dev.set_param("coopstore", coopstore)
    dev.set_param("rowkernel", rowkernel)
    # single-view launches take the kernel instance compiled for ONE view (footprint record in registers, flags from the
    # kept brick minimum); the unlisted cases that also cut their launches into chunks run the general instance
    # ("oneview" 0), the listed ones the one-view instance with a list -- and its records -- per chunk
    dev.set_param("oneview", 0 if recordbytes and not livelist else 1)
    # the state of a brick requested next to its footprint record, before the early-return test ("eagerstate"): by the
    # library's rule (listed launches, or the last list held most workgroups), and forced on for the unlisted launches,
    # where most waves then return early with the state in flight
    dev.set_param("eagerstate", 1 if livelist == 0 else -1)
    # the live list of a one-view launch with the footprint records and per-wave live bits in its entries (the default),
    # and with workgroup ids only in the cases that force the cooperative write-back
    dev.set_param("listrecords", 0 if coopstore == 1 else 1)
    orc = O.OracleGrid(opt)
    base = O.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
    for i in range(nv):
        sdf = base
        if i % 5 == 3:  # plateaus: ties with the running maximum
            sdf = (np.round(base * 8) / 8).astype(np.float32)
        if i % 5 == 4:  # noise + a few non-finite pixels
            sdf = (base + rng.uniform(-0.05, 0.05, base.shape)).astype(np.float32)
            sdf[rng.rand(*sdf.shape) < 0.002] = np.nan
            sdf[rng.rand(*sdf.shape) < 0.002] = np.finfo(np.float32).min
        assert dev.Carve(views[i], sdf), vc.last_error()
        orc.carve(views[i], sdf)
        assert dev.get_param("brick_min_valid") == (1 if i <= 9 else 0)
        ds, du = dev.download()
        os_, ou = orc.download()
        assert np.array_equal(du, ou), (kw, i, int((du != ou).sum()))
        nan_d, nan_o = np.isnan(ds), np.isnan(os_)
        assert np.array_equal(nan_d, nan_o), (kw, i)
        assert np.array_equal(np.where(nan_d, 0, ds.view(np.uint32)), np.where(nan_o, 0, os_.view(np.uint32))), (kw, i)
        if i % 3 == 0 and not nan_o.any():
            for iso in (0.0, 0.25, -0.125, 0.1):
                m1 = dev.ExtractIsoSurface(iso, True)
                dev.set_param("mcskip", 0)
                m0 = dev.ExtractIsoSurface(iso, True)
                dev.set_param("mcskip", 2)
                om = orc.marching_cubes(iso, True)
                assert_mesh_equal(m1, om, "%s view %d iso %g (bricks skipped)" % (kw, i, iso))
                assert_mesh_equal(m0, om, "%s view %d iso %g (every brick read)" % (kw, i, iso))
        if i == 4:  # the per-view kernel does not keep the minima; the next fused launch restarts them from lowest()
            dev.set_param("fused", 0)
            assert dev.Carve(views[0], base), vc.last_error()
            orc.carve(views[0], base)
            assert dev.get_param("brick_min_valid") == 0
            dev.set_param("fused", 1)
        if i == 9:  # a state from outside ("update_num == 0 implies sdf == lowest()" is gone): no minima from here on
            s2 = np.where(rng.rand(ds.size) < 0.3, ds + np.float32(0.5), ds).astype(np.float32)
            s2 = np.where(np.isnan(s2), np.float32(0.0), s2)
            dev.upload(s2, du)
            orc.upload(s2, du)
            assert dev.get_param("brick_min_valid") == 0


@pytest.mark.parametrize("kw", [dict(voxel_update=1, use_truncation=True, truncation_band=0.1),
                                dict(use_truncation=True, truncation_band=0.1)])
def test_brick_minima_after_writes_that_bypass_the_fused_kernel(kw):
    """A fused launch does not rewrite every brick minimum: a wave whose views all lie below the truncation limit
    returns before it has read anything, a workgroup left off the live list never starts.  After the per-view kernel
    ("fused" 0) has changed the state the old minima are stale -- and on a slab whose first write was the per-view
    kernel the array has never been written at all -- so the next fused launch must start every entry from lowest()
    ("may hold an untouched voxel": never drops a view, never lets marching cubes skip the brick).  The sequence of
    the round-3 advisor finding: fused launches, a per-view launch that pulls values below the iso level, a fused
    truncation launch in which EVERY wave returns early, then extraction and further carving against the oracle."""
    n, nv, w, h = 72, 8, 200, 150
    uo = UpdateOption(**kw)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    base = O.make_sdf(masks[0], use_truncation=True, band=uo.truncation_band)
    low = np.maximum(base - np.float32(0.9), np.float32(-1.0)).astype(np.float32)  # valid samples far below the iso level
    nothing = np.full_like(base, -2.0)                                             # every sample < -1: no voxel changes
    for first_fused in (True, False):
        dev = vc.VoxelCarver(opt)
        assert dev.Init(), vc.last_error()
        dev.set_param("defer", 0)
        orc = O.OracleGrid(opt)
        steps = []
        if first_fused:
            steps += [(1, views[i], base) for i in range(3)]      # minima valid
        steps += [(0, views[3], low), (0, views[4], base)]        # per-view kernel: the minima know nothing of these
        steps += [(1, views[5], nothing)]                         # fused: every wave returns early / no workgroup is live
        for fused, view, img in steps:
            dev.set_param("fused", fused)
            assert dev.Carve(view, img), vc.last_error()
            orc.carve(view, img)
        assert_state_equal(dev, orc, "%s first_fused=%s" % (kw, first_fused))
        for iso in (0.0, -0.25, 0.3):
            assert_mesh_equal(dev.ExtractIsoSurface(iso, True), orc.marching_cubes(iso, True),
                              "%s first_fused=%s iso %g" % (kw, first_fused, iso))
        # ... and the minima left behind must not drop views that still change something (kMax: `ub <= smin`)
        for i in (6, 7, 0):
            assert dev.Carve(views[i], base), vc.last_error()
            orc.carve(views[i], base)
        assert_state_equal(dev, orc, "%s first_fused=%s, later views" % (kw, first_fused))
        assert_mesh_equal(dev.ExtractIsoSurface(0.0, True), orc.marching_cubes(0.0, True), "%s later" % (kw,))
        dev.close()


def test_slab_planner_estimate_and_planned_cuts():
    """vcy_plan_z_slabs: (1) its per-layer estimate of the (brick, view) pairs a carve will process follows what the
    kernel really processes ("paircount"); with view dropping off every pair is processed and every layer costs the
    same; (2) the cuts are whole brick layers, cover the grid, and slabs carved with them merge into the whole grid's
    mesh array for array; (3) the same inputs give the same cuts again."""
    n, nv, w, h = 128, 12, 320, 240
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, nv, w, h)
    sdf = O.make_sdf(masks[0])
    whole = vc.VoxelCarver(opt)
    assert whole.Init(), vc.last_error()
    d = whole.upload_sdf(sdf)
    batch = vc.VoxelCarver.prepare_batch(views, [d] * nv)
    whole.set_param("paircount", 1)
    assert whole.CarveBatchDevice(batch), vc.last_error()
    proc, total, per_layer = whole.last_carve_pairs()
    assert total == (n // 8) ** 3 * nv and 0 < proc < total and per_layer.sum() == proc and len(per_layer) == n // 8
    planner = vc.VoxelCarver(opt, z_range=(0, 8))
    assert planner.Init(), vc.last_error()
    bounds, cost = planner.plan_z_slabs(batch, None, 4, stride=1, brick_cost=1e-9)   # (the pairs alone)
    assert len(cost) == n // 8
    assert np.corrcoef(cost, per_layer)[0, 1] > 0.97, (cost, per_layer)
    assert 0.9 * proc < cost.sum() < 1.4 * proc          # an over-estimate by construction (lower bounds of the minima)
    assert bounds[0] == 0 and bounds[-1] == n and all(b % 8 == 0 for b in bounds) and sorted(set(bounds)) == bounds
    assert planner.plan_z_slabs(batch, None, 4, stride=1, brick_cost=1e-9)[0] == bounds
    b2, c2 = planner.plan_z_slabs(batch, None, 4)        # default stride and brick cost
    parts = [c2[b2[s] // 8:b2[s + 1] // 8].sum() for s in range(4)]
    eq = [c2[s * 4:(s + 1) * 4].sum() for s in range(4)]
    assert max(parts) <= max(eq) + 1e-9                  # never worse than equal thickness, by its own model
    # (the object is in the middle: the planned outer slabs are thicker than the inner ones)
    assert b2[1] - b2[0] >= b2[2] - b2[1]
    planner.set_param("cull", 0)
    _, flat = planner.plan_z_slabs(batch, None, 4, stride=1, brick_cost=1e-9)
    assert np.all(flat == flat[0]) and abs(flat[0] - (n // 8) ** 2 * nv) < 1e-3
    with pytest.raises(RuntimeError):
        planner.plan_z_slabs(batch, None, n // 8 + 1)
    planner.close()
    # slabs with the planned cuts: state and merged mesh equal the whole grid's
    from vacancy_amd import dist as vdist
    slabs = []
    for s in range(4):
        c = vc.VoxelCarver(opt, z_range=(b2[s], b2[s + 1]))
        assert c.Init(), vc.last_error()
        assert c.CarveBatchDevice(batch), vc.last_error()
        slabs.append(c)
    ws, wu = whole.download()
    assert np.array_equal(np.concatenate([c.download()[0] for c in slabs]).view(np.uint32), ws.view(np.uint32))
    assert np.array_equal(np.concatenate([c.download()[1] for c in slabs]), wu)
    vc.halo_exchange(slabs)
    merged = vdist.merge_meshes([c.ExtractIsoSurface(0.0, True) for c in slabs])
    ref = whole.ExtractIsoSurface(0.0, True)
    assert np.array_equal(merged["vertices"].view(np.uint32), ref["vertices"].view(np.uint32))
    assert np.array_equal(merged["faces"], ref["faces"]) and np.array_equal(merged["keys"], ref["keys"])
    # with view dropping off the kernel processes every pair
    whole.set_param("cull", 0)
    whole.reset()
    assert whole.CarveBatchDevice(batch)
    proc0, total0, _ = whole.last_carve_pairs()
    assert proc0 == total0 == total
    for c in slabs:
        c.close()
    whole.free_device(d)
    whole.close()


def test_carve_log_records_queued_steps_without_synchronising():
    """"carvetimer": one record per chunk of every fused launch, read after the fact (vcy_carve_log); setting the
    parameter clears the log; vcy_last_carve_ms is the last launch's sum."""
    n, nv, w, h = 64, 6, 160, 120
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, nv, w, h)
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    d = dev.upload_sdf(O.make_sdf(masks[0]))
    batch = vc.VoxelCarver.prepare_batch(views, [d] * nv)
    dev.set_param("carvetimer", 1)
    for _ in range(5):
        dev.reset()
        assert dev.CarveBatchDevice(batch)
    log = dev.carve_log(clear=False)
    assert len(log) == 5 and all(r[3] == 1 for r in log) and log[0][0] == 0.0
    assert all(b[0] >= a[0] + a[1] + a[2] - 1e-3 for a, b in zip(log, log[1:]))   # launches follow each other
    pre, ker = dev.last_carve_ms()
    assert abs(pre - log[-1][1]) < 1e-6 and abs(ker - log[-1][2]) < 1e-6 and ker > 0
    dev.set_param("recordbytes", 600)   # several chunks per launch
    dev.set_param("carvetimer", 1)
    dev.reset()
    assert dev.CarveBatchDevice(batch)
    log = dev.carve_log()
    assert len(log) > 1 and log[0][3] == 1 and all(r[3] == 0 for r in log[1:])
    assert dev.carve_log() == []
    dev.free_device(d)
    dev.close()


@pytest.mark.parametrize("kw,coopstore,rowkernel", [(_TSDF, -1, 0), (_TSDF, 1, 0), (_TRUNC, 1, 0), (dict(), 1, 0),
                                                    (dict(voxel_update=1, voxel_update_weight=0.5), -1, 0),
                                                    (_TSDF, -1, -1), (_TRUNC, -1, -1), (dict(), -1, -1), (dict(), -1, 3),
                                                    (dict(voxel_update=1, voxel_update_weight=0.5), -1, -1),
                                                    (dict(sdf_interp=0), -1, -1)])
def test_cooperative_write_back_with_groups_of_views(kw, coopstore, rowkernel):
    """Launches of 1, 2, 3, 5 and 8 views over a carved grid with the cooperative write-back (the four waves of a
    workgroup exchange their bricks through LDS and store whole row segments): by the library's rule (weighted
    average, up to 8 views) and forced on in the other modes.  A sphere deep enough inside the grid that whole
    workgroups -- and single waves of a workgroup -- drop every view and leave before the barrier; the rows those
    waves would have stored belong to their neighbours' bricks (the first version lost them).  nx = 96: every
    workgroup has its four waves inside the grid; state against the oracle after every launch, mesh at the end."""
    n, w, h = 96, 160, 120
    groups = [1, 2, 3, 5, 8, 1]
    nv = sum(groups)
    uo = UpdateOption(**kw)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    base = O.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
    rng = np.random.RandomState(11)
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    dev.set_param("defer", 0)
    dev.set_param("coopstore", coopstore)
    # (rowkernel 0: the workgroup-per-block kernel the docstring describes; -1: the few-view flavour takes every one of
    # these launches -- a wave walks the four bricks of a row segment, pairs of several views per brick, bricks whose
    # every view is dropped neither read nor stored; 3: launches of up to three views only)
    dev.set_param("rowkernel", rowkernel)
    dev.set_param("eagerstate", 1 if coopstore == 1 else -1)  # (1: the launches of ONE view request the state early whether listed or not)
    orc = O.OracleGrid(opt)
    d_base = dev.upload_sdf(base)
    noisy = (base + rng.uniform(-0.03, 0.03, base.shape)).astype(np.float32)
    d_noisy = dev.upload_sdf(noisy)
    first = 0
    for gi, g in enumerate(groups):
        imgs = [(noisy, d_noisy) if (first + j) % 4 == 3 else (base, d_base) for j in range(g)]
        assert dev.CarveBatchDevice(views[first:first + g], [p for _, p in imgs]), vc.last_error()
        for j in range(g):
            orc.carve(views[first + j], imgs[j][0])
        assert_state_equal(dev, orc, "%s coopstore %d rowkernel %d group %d (%d views)" % (kw, coopstore, rowkernel, gi, g))
        first += g
    assert_mesh_equal(dev.ExtractIsoSurface(0.0, True), orc.marching_cubes(0.0, True), "%s coopstore %d" % (kw, coopstore))
    dev.free_device(d_base)
    dev.free_device(d_noisy)


@pytest.mark.parametrize("mode", ["wa_max_update_300", "default", "max_update_70000", "tsdf"])
def test_counters_widen_lazily_across_the_256th_view(mode):
    """update_num of a voxel cannot exceed the number of views applied since the fill, so the counter array is u8 (5
    B/voxel with the sdf) until the 256th view and is widened in one pass then -- also between two chunks of one batch
    call -- to whatever voxel_max_update_num needs in the end (u16 for the default 255, u32 for 70000, where the fused
    kernel now serves the first 65535 views).  State and meshes equal the oracle's before and after; a reset goes back
    to one byte, an upload of counts above 255 widens at once, "lazycount" 0 is round 4's layout."""
    n, nv, w, h = 40, 300, 128, 96
    kw = dict(SYN_MODES[mode])
    if mode in ("default", "tsdf"):
        kw["voxel_max_update_num"] = 255  # (the reference's default: counts up to 256 -> u16 in the end)
    uo = UpdateOption(**kw)
    final = 4 if mode == "max_update_70000" else 2
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    sdfs = [vc.make_sdf(m, use_truncation=bool(uo.use_truncation), band=uo.truncation_band) for m in masks[:8]]
    orc = O.OracleGrid(opt)
    dev = vc.VoxelCarver(opt)
    assert dev.Init(), vc.last_error()
    assert (dev.get_param("count_bytes"), dev.get_param("count_bytes_final")) == (1, final)
    devs = [dev.upload_sdf(s_) for s_ in sdfs]
    img = lambda i: i % len(sdfs)
    state_at = {}
    for i in range(nv):
        orc.carve(views[i], sdfs[img(i)])
        if i + 1 in (190, 260):
            state_at[i + 1] = orc.download()
    assert dev.CarveBatchDevice(views[:190], [devs[img(i)] for i in range(190)]), vc.last_error()
    assert dev.get_param("count_bytes") == 1
    ds, du = dev.download()
    assert np.array_equal(du, state_at[190][1]) and np.array_equal(ds.view(np.uint32), state_at[190][0].view(np.uint32))
    m190 = dev.ExtractIsoSurface(0.0, True)
    # 190 + 64 = 254 views still fit one byte; the last chunk of this call (46 views) does not
    assert dev.CarveBatchDevice(views[190:], [devs[img(i)] for i in range(190, nv)]), vc.last_error()
    assert dev.get_param("count_bytes") == 2
    assert_state_equal(dev, orc, mode + " after 300 views")
    assert_mesh_equal(dev.ExtractIsoSurface(0.0, True), orc.marching_cubes(0.0, True), mode)
    if mode == "wa_max_update_300":
        assert int(dev.download()[1].max()) == nv
    # the per-view kernel and single-view fused launches cross the boundary view by view
    for fused in (0, 1):
        one = vc.VoxelCarver(opt)
        assert one.Init()
        one.set_param("fused", fused)
        one.set_param("defer", 0)
        one.upload(*state_at[190])
        assert one.get_param("count_bytes") == 1
        d1 = [one.upload_sdf(s_) for s_ in sdfs]
        # (the upload leaves views_carved at the largest count it saw, not at 190: carve until the width must change)
        for i in range(190, 260):
            assert one.CarveDevice(views[i], d1[img(i)]), vc.last_error()
        s1, u1 = one.download()
        assert np.array_equal(u1, state_at[260][1]), (mode, fused)
        assert np.array_equal(s1.view(np.uint32), state_at[260][0].view(np.uint32)), (mode, fused)
    # reset: one byte again, and the same 190 views give the same mesh
    dev.reset()
    assert dev.get_param("count_bytes") == 1
    assert dev.CarveBatchDevice(views[:190], [devs[img(i)] for i in range(190)]), vc.last_error()
    assert_mesh_equal(dev.ExtractIsoSurface(0.0, True), m190, "after reset")
    # an upload of counts beyond 255 widens first
    os_, ou = orc.download()
    if int(ou.max()) > 255:
        dev.reset()
        dev.upload(os_, ou)
        assert dev.get_param("count_bytes") == 2
        assert_state_equal(dev, orc, "upload of wide counts")
    # "lazycount" 0: the final width from the start
    old = vc.VoxelCarver(opt)
    assert old.Init()
    old.set_param("lazycount", 0)
    assert old.get_param("count_bytes") == final
    if final <= 2:
        d2 = [old.upload_sdf(s_) for s_ in sdfs]
        assert old.CarveBatchDevice(views, [d2[img(i)] for i in range(nv)]), vc.last_error()
        assert_state_equal(old, orc, "lazycount 0")


def test_halo_exchange_between_slabs_of_different_counter_width():
    """Halo packs carry update_num at its final width whatever the sending slab currently stores (vcy_halo_bytes does
    not change when a slab is widened), and the receiver converts to its own width: a u16 slab below a u8 slab and the
    other way round, through the host packs, vcy_halo_copy_from and the native all-gather -- merged mesh == oracle."""
    from vacancy_amd import dist as vdist
    n, nv, w, h = 44, 6, 128, 96
    opt = synth.sphere_option(n, UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.2))
    views, masks = synth.sphere_views(n, nv, w, h)
    sdfs = [vc.make_sdf(m, use_truncation=True, band=0.2) for m in masks]
    orc = O.OracleGrid(opt)
    for i in range(nv):
        orc.carve(views[i], sdfs[i])
    want = orc.marching_cubes(0.0, True)
    lib = vc.capi.load()
    for wide_rank, how in ((0, "host"), (1, "host"), (0, "copy"), (1, "copy"), (1, "rccl")):
        ranks = []
        for r in range(2):
            c = vc.VoxelCarver(opt, z_range=vdist.slab_range(n, r, 2))
            assert c.Init(), vc.last_error()
            if r == wide_rank:
                c.set_param("lazycount", 0)
            for i in range(nv):
                assert c.Carve(views[i], sdfs[i])
            ranks.append(c)
        assert [c.get_param("count_bytes") for c in ranks] == ([2, 1] if wide_rank == 0 else [1, 2])
        assert int(lib.vcy_halo_bytes(ranks[0].ctx)) == int(lib.vcy_halo_bytes(ranks[1].ctx)) == 2 * n * n * 6
        if how == "host":
            gathered = np.concatenate([c.halo_pack_host() for c in ranks])
            for r, c in enumerate(ranks):
                c.halo_unpack_host(gathered, r, 2)
        elif how == "copy":
            assert lib.vcy_halo_copy_from(ranks[0].ctx, None) == 0
            assert lib.vcy_halo_copy_from(ranks[1].ctx, ranks[0].ctx) == 0, vc.last_error()
            # (neither slab's counters are re-allocated for the exchange: the two slices are converted through a staging
            # buffer of the receiving context -- the neighbour's array may be in use by its own driver thread)
            assert [c.get_param("count_bytes") for c in ranks] == ([2, 1] if wide_rank == 0 else [1, 2])
        else:
            vc.halo_allgather(ranks)
        merged = vdist.merge_meshes([c.ExtractIsoSurface(0.0, True) for c in ranks])
        assert_mesh_equal(merged, want, "wide rank %d via %s" % (wide_rank, how))


@pytest.mark.parametrize("split,kw", [(0, dict()), (1, dict()),
                                      (1, dict(voxel_update=1, use_truncation=True, truncation_band=0.1))])
def test_sharded_silhouette_producer_equals_per_slab_producer(split, kw, monkeypatch):
    """vcy_carve_batch_silhouettes_sharded: the slabs of one grid share the producer of SDF images -- rank r of R builds
    the views r, r + R, ... of every chunk of 32, the images are gathered, every slab carves the chunk from the
    gathered copy while the next chunk is produced.  On one GPU: split = 0, the three slabs share the device's images
    (R = 1); split = 1 (test hook VCY_TEST_SPLIT_PRODUCERS), every slab is a producer rank of its own and the gather
    runs as device copies -- the share / slot / gather layout of a 3-GPU run.  70 views = chunks of 32, 32 and 6 (the
    last one leaves rank 2's second slot empty), one view with a sub-ROI: every slab's state == the per-slab form's
    (every slab building every SDF) == the oracle's."""
    from vacancy_amd import dist as vdist
    n, nv, w, h = 48, 70, 160, 120
    uo = UpdateOption(**kw)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    rng = np.random.RandomState(5)
    for i in range(0, nv, 7):  # distinct silhouettes: a wrong slot would show
        masks[i] = (rng.rand(h, w) < 0.5).astype(np.uint8) * 255
    views[3].roi_min[0], views[3].roi_min[1], views[3].roi_max[0], views[3].roi_max[1] = 20, 10, 130, 100
    monkeypatch.setenv("VCY_TEST_SPLIT_PRODUCERS", str(split))

    def slabs():
        out = []
        for r in range(3):
            c = vc.VoxelCarver(opt, z_range=vdist.slab_range(n, r, 3))
            assert c.Init(), vc.last_error()
            out.append(c)
        return out

    a, b = slabs(), slabs()
    for c in a:
        assert c.CarveBatchSilhouettes(views, masks), vc.last_error()
    assert vc.carve_batch_silhouettes_sharded(b, views, masks)
    for ca, cb in zip(a, b):
        assert ca.state_diff(cb) == 0
        prod, carve, wall = cb.last_stream_ms()
        assert prod > 0 and carve > 0 and wall > 0
    orc = O.OracleGrid(opt)
    for i in range(nv):
        rmin, rmax = tuple(views[i].roi_min), tuple(views[i].roi_max)
        orc.carve(views[i], O.make_sdf(masks[i], rmin, rmax, use_truncation=bool(uo.use_truncation), band=uo.truncation_band))
    os_, ou = orc.download()
    assert np.array_equal(np.concatenate([c.download()[1] for c in b]), ou)
    assert np.array_equal(np.concatenate([c.download()[0] for c in b]).view(np.uint32), os_.view(np.uint32))
    # a second call on the same contexts reuses the cached producer buffers; a failing argument check leaves the state alone
    for c in a + b:
        c.reset()
    assert vc.carve_batch_silhouettes_sharded(b, views[:5], masks[:5])
    for c in a:
        assert c.CarveBatchSilhouettes(views[:5], masks[:5])
    assert all(ca.state_diff(cb) == 0 for ca, cb in zip(a, b))
    with pytest.raises(RuntimeError):
        bad = vc.make_view(np.eye(3, 4, dtype=np.float32), 10.0, 10.0, 5.0, 5.0, w, h)
        bad.roi_max[0] = w  # outside the image
        vc.carve_batch_silhouettes_sharded(b, [bad], masks[:1])
    vc.capi.load().vcy_halo_shutdown()  # releases the producer groups as well


def test_one_rank_of_the_process_per_gpu_producer_equals_the_batch_call():
    """vacancy_amd.dist.carve_silhouettes_sharded with world == 1 (what `bench.py --variants streamed` runs on one GPU,
    and the degenerate case of the one-process-per-GPU job): the rank builds every SDF image in place in the set its
    slabs carve from -- no exchange, no copy on another stream -- over several chunks, so that both image sets are
    reused while an earlier chunk's carve may still be queued.  State == CarveBatchSilhouettes == the oracle."""
    import torch
    from vacancy_amd import dist as vdist
    n, nv, w, h = 40, 75, 128, 96
    uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, w, h)
    rng = np.random.RandomState(11)
    for i in range(0, nv, 4):
        masks[i] = (rng.rand(h, w) < 0.4).astype(np.uint8) * 255

    def slabs():
        out = []
        for r in range(2):
            c = vc.VoxelCarver(opt, z_range=vdist.slab_range(n, r, 2))
            assert c.Init(), vc.last_error()
            out.append(c)
        return out

    a, b = slabs(), slabs()
    for c in a:
        assert c.CarveBatchSilhouettes(views, masks), vc.last_error()
    # a pending kernel on torch's own stream must not matter to the carvers' streams
    junk = torch.empty(1 << 24, device="cuda").normal_()
    info = vdist.carve_silhouettes_sharded(b, 0, 1, views, masks, chunk=16)
    assert info["views_built_by_this_rank"] == nv and junk.numel() > 0
    for ca, cb in zip(a, b):
        assert ca.state_diff(cb) == 0
    orc = O.OracleGrid(opt)
    for i in range(nv):
        orc.carve(views[i], O.make_sdf(masks[i], use_truncation=True, band=0.1))
    os_, ou = orc.download()
    assert np.array_equal(np.concatenate([c.download()[1] for c in b]), ou)
    assert np.array_equal(np.concatenate([c.download()[0] for c in b]).view(np.uint32), os_.view(np.uint32))


def test_make_sdf_batch_into_caller_owned_images():
    """vcy_make_sdf_batch_device: a rank's share of the SDF images of a one-process-per-GPU job, built into images the
    caller owns (one allocation, as the all-gather's send buffer is) -- bit-equal to the oracle's transform."""
    import ctypes as C
    opt = synth.sphere_option(16, UpdateOption(use_truncation=True, truncation_band=0.25))
    dev = vc.VoxelCarver(opt)
    assert dev.Init()
    n, w, h = 40, 96, 72
    views, masks = synth.sphere_views(16, n, w, h)
    rng = np.random.RandomState(9)
    for i in range(n):
        if i % 3:
            masks[i] = (rng.rand(h, w) < 0.3 + 0.01 * i).astype(np.uint8) * 255
    lib = vc.capi.load()
    buf = C.c_void_p()
    stride = w * h * 4
    assert lib.vcy_device_alloc(dev.ctx, n * stride, C.byref(buf)) == 0
    assert dev.make_sdf_batch_into(views, masks, [buf.value + i * stride for i in range(n)]), vc.last_error()
    for i in range(n):
        got = dev.download_image(C.c_void_p(buf.value + i * stride), (h, w))
        want = O.make_sdf(masks[i], use_truncation=True, band=0.25)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), i
    dev.free_device(buf)


@pytest.mark.parametrize("world", [2, 3])
def test_extract_voxel_over_z_slabs(world):
    """ExtractVoxel on a grid cut into z-slabs (row f2 over slabs): the keep predicate and the compaction run per slab on
    the device (with inside_empty the -z neighbour of a slab's first slice is its halo slice), the kept ids of all slabs
    are walked in z order by ONE drifting cube (vcy_voxel_cubes, host) -- the reference translates one cube mesh from
    voxel to voxel, extract_voxel.cc:290-311.  Bunny, all six views, both predicates: == the single context == the oracle,
    through the Python ShardedVoxelCarver and through the bare C-ABI calls."""
    from vacancy_amd import dist as vdist
    from vacancy_amd.sharded import ShardedVoxelCarver
    opt = B.bunny_option(10.0)
    views = B.bunny_views(lambda t, q: synth.affine_inverse(synth.pose_from_tum(t, q)))
    masks = B.load_masks()
    whole = vc.VoxelCarver(opt)
    assert whole.Init()
    sh = ShardedVoxelCarver(opt, [0] * world, 1)
    assert sh.Init()
    orc = O.OracleGrid(opt)
    for i in range(6):
        sdf = O.make_sdf(masks[i])
        assert whole.Carve(views[i], sdf)
        for c in sh.slabs:
            assert c.Carve(views[i], sdf)
        orc.carve(views[i], sdf)
        if i in (0, 3, 5):
            for inside_empty in (False, True):
                want = orc.extract_voxel(inside_empty)
                one = whole.ExtractVoxel(inside_empty)
                many = sh.ExtractVoxel(inside_empty)
                for got in (one, many):
                    assert np.array_equal(got["faces"], want["faces"]), (i, inside_empty)
                    assert np.array_equal(got["vertices"].view(np.uint32), want["vertices"].view(np.uint32)), (i, inside_empty)
    assert len(sh.ExtractVoxel(False)["vertices"]) == 683400
    # a slab above another one cannot decide the on-surface test of its first slice without its halo
    nz = whole.dims[2]
    lone = vc.VoxelCarver(opt, z_range=vdist.slab_range(nz, 1, world))
    assert lone.Init()
    assert lone.Carve(views[0], O.make_sdf(masks[0]))
    assert len(lone.extract_voxel_ids(False)) >= 0
    with pytest.raises(RuntimeError):
        lone.extract_voxel_ids(True)
    with pytest.raises(RuntimeError):
        lone.ExtractVoxel(False)  # (the single-context call still refuses a slab)


@pytest.mark.gpu
def test_clock_probe_runs_beside_other_work():
    """vcy_clock_probe_*: one wave samples the shader clock counter against the 100 MHz reference while other kernels run
    (bench.py prices the carve kernel's VALU issue cycles at the clock of the timed run with it)."""
    probe = vc.ClockProbe(0)
    rd, cp = vc.measure_bandwidth(0, 1 << 28, 2)   # work on other streams while the probe wave is resident
    assert rd > 0 and cp > 0
    r = probe.stop()
    assert r["samples"] >= 2 and r["covered_ms"] > 0.0
    assert 0.3e9 < r["mean_hz"] < 3.5e9, r          # MI355X: 2.4 GHz peak shader clock
    assert 0.3e9 < r["settled_hz"] < 3.5e9, r       # (the second half of the span)
    assert r["min_hz"] <= r["mean_hz"] * 1.01 and r["max_hz"] >= r["mean_hz"] * 0.99
    assert probe.stop() == r                         # idempotent
    # a probe that fills its buffer ends by itself
    short = vc.ClockProbe(0, max_samples=4)
    time.sleep(0.01)
    assert short.stop()["samples"] == 4


def test_extractions_on_eight_streams_beside_a_long_carve():
    """The chained scan of an extraction (scan_chained_kernel) draws its chunk numbers as tickets, so its forward progress
    does not depend on the order in which workgroups of a grid are started -- which is exactly what other contexts'
    streams perturb.  Eight contexts (a stream each, grids large enough for several chunks per scan) extract at the same
    time from eight host threads, round after round at changing iso levels, while a ninth context keeps the device
    busy with long carve launches on its own stream; every mesh equals the oracle's."""
    import threading
    n, nv, w, h = 264, 6, 320, 240
    opt = synth.sphere_option(n)
    views, masks = synth.sphere_views(n, nv, w, h)
    orc = O.OracleGrid(opt)
    sdfs = [O.make_sdf(m) for m in masks]
    for i in range(nv):
        orc.carve(views[i], sdfs[i])
    isos = [0.0, 0.013, -0.02, 0.05]
    want = [orc.marching_cubes(iso, True) for iso in isos]
    ctxs = []
    for _ in range(8):
        c = vc.VoxelCarver(opt)
        assert c.Init(), vc.last_error()
        c.set_param("mcskip", 0)   # the dense pass: the scans cover every word block (several 1024-block chunks)
        for i in range(nv):
            assert c.Carve(views[i], sdfs[i])
        c.sync()
        ctxs.append(c)
    # the ninth context: 512^3, 32 views without view dropping, launched again and again until the others are done
    big = vc.VoxelCarver(synth.sphere_option(512))
    assert big.Init(), vc.last_error()
    big.set_param("cull", 0)
    bviews, bmasks = synth.sphere_views(512, 32, 640, 480)
    bimgs = [big.upload_sdf(O.make_sdf(m)) for m in bmasks[:1]] * 32
    batch = vc.VoxelCarver.prepare_batch(bviews, bimgs)
    stop = threading.Event()
    errors = []

    def keep_busy():
        try:
            while not stop.is_set():
                for _ in range(4):
                    assert big.CarveBatchDevice(batch), vc.last_error()
                big.sync()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def extract(c, out):
        try:
            for rnd in range(3):
                for k, iso in enumerate(isos):
                    out.append((k, c.ExtractIsoSurface(iso, True)))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    busy = threading.Thread(target=keep_busy)
    busy.start()
    outs = [[] for _ in ctxs]
    ths = [threading.Thread(target=extract, args=(c, o)) for c, o in zip(ctxs, outs)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    stop.set()
    busy.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in ths) and not busy.is_alive(), "an extraction did not finish: the scan hung"
    for ci, o in enumerate(outs):
        assert len(o) == 3 * len(isos)
        for k, m in o:
            assert_mesh_equal(m, want[k], "context %d iso %g" % (ci, isos[k]))
    # (several chunks per scan: the word blocks of this grid exceed one 1024-block chunk)
    assert (n - 1) * (n - 1) * ((n + 63) // 64) > 256 * 1024
