"""CPU-only tests: C-ABI library loads and exports every declared symbol, host-side SDF builder
and camera arithmetic of the product agree bit-for-bit with the oracle, MC table integrity."""
import json
import os
import re

import numpy as np
import pytest

import bunny_data as B
import oracle_lib as O
from vacancy_amd import capi, carver, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "appendix_c.json")))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    header = open(os.path.join(ROOT, "include", "vacancy_hip.h")).read()
    declared = set(re.findall(r"\b(vcy_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "libvacancy_hip.so does not export " + name
    # and the ctypes mirror binds all of them
    assert declared == set(lib._vcy_symbols), declared ^ set(lib._vcy_symbols)


def test_struct_sizes_match_header_layout():
    import ctypes as C
    assert C.sizeof(capi.UpdateOption) == 28
    assert C.sizeof(capi.CarverOption) == 60
    assert C.sizeof(capi.View) == 48 + 16 + 4 + 16 + 8
    assert C.sizeof(capi.Mesh) == 48


def test_no_device_is_a_loud_failure_not_a_fallback():
    import ctypes as C
    lib = capi.load()
    n = C.c_int(0)
    lib.vcy_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    c = carver.VoxelCarver(B.bunny_option())
    assert not c.Init()
    assert "no HIP device" in carver.last_error()


def test_host_sdf_builder_equals_oracle():
    masks = B.load_masks()
    rng = np.random.RandomState(0)
    noise = (rng.rand(97, 131) < 0.4).astype(np.uint8) * 255
    noise[rng.rand(97, 131) < 0.1] = 128  # not-255 values are "outside" (voxel_carver.cc:109)
    cases = [(m, None, None) for m in masks] + [(masks[0], (10, 20), (300, 200)), (noise, None, None),
                                               (noise, (5, 7), (100, 60)),
                                               (np.full((20, 30), 255, np.uint8), None, None),
                                               (np.zeros((20, 30), np.uint8), None, None)]
    for mask, rmin, rmax in cases:
        assert np.array_equal(carver.distance_transform_l1(mask, rmin, rmax).view(np.uint32),
                              O.distance_transform_l1(mask, rmin, rmax).view(np.uint32))
        for norm in (True, False):
            for trunc in (False, True):
                a = carver.make_sdf(mask, rmin, rmax, norm, trunc, 0.1)
                b = O.make_sdf(mask, rmin, rmax, norm, trunc, 0.1)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (mask.shape, rmin, norm, trunc)


def test_host_pose_arithmetic_equals_oracle():
    for _, t, q in B.load_tum():
        assert np.array_equal(synth.affine_inverse(synth.pose_from_tum(t, q)),
                              O.affine_inverse(O.pose_from_tum(t, q)))
    rng = np.random.RandomState(1)
    for _ in range(20):
        p = rng.uniform(-500, 500, 3)
        assert np.array_equal(synth.lookat_c2w(p, (0, 0, 0), (0, 1, 0)), O.lookat_c2w(p, (0, 0, 0), (0, 1, 0)))
    for h, fov in ((480, 60.0), (720, 60.0), (1080, 45.0)):
        assert synth.focal_from_fov_y(h, fov) == np.float32(O.focal_from_fov_y(h, fov))


def test_mc_case_table_integrity():
    txt = open(os.path.join(ROOT, "include", "vacancy_mc_cases.inc")).read()
    strs = re.findall(r'"([0-9a-b]*)"', txt)
    assert len(strs) == 256
    tri = [[int(ch, 16) for ch in s] for s in strs]
    gold = GOLD["mc_tables"]
    assert sum(len(t) // 3 for t in tri) == gold["n_triangles"]
    assert max(len(t) // 3 for t in tri) == gold["max_tris_per_cube"]
    assert sum(sum(t) - (16 - len(t)) for t in tri) == gold["tri_sum"]  # -1 terminators included
    ec = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
    edge_sum = 0
    for c in range(256):
        cut = sum(1 << k for k, (a, b) in enumerate(ec) if ((c >> a) ^ (c >> b)) & 1)
        used = 0
        for e in tri[c]:
            used |= 1 << e
        assert cut == used, c  # edge table == edges used by the case's triangles
        edge_sum += cut
    assert edge_sum == gold["edge_sum"]


def _build_host():
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "vacancy_amd", "host"), "-s"], check=True)


def test_cpp_facade_host_utilities():
    """The C++ facade (include/vacancy/*.h + vacancy_amd/host) builds; its PNG decoder reads the
    reference's masks exactly and its camera arithmetic reproduces the reference's w2c."""
    import subprocess
    _build_host()
    out = subprocess.run([os.path.join(ROOT, "vacancy_amd", "host", "host_selftest"), B.BUNNY],
                         check=True, capture_output=True, text=True).stdout.splitlines()
    masks = B.load_masks()
    rows = [l.split() for l in out if l.startswith("MASK")]
    assert len(rows) == 6
    for i, r in enumerate(rows):
        assert (int(r[2]), int(r[3])) == (320, 240)
        assert int(r[4]) == int(masks[i].astype(np.int64).sum())
    w2c = np.array([float(x) for x in [l for l in out if l.startswith("W2C")][0].split()[1:]], np.float32)
    ref = GOLD["w2c_f32"]["2"]
    got = w2c.reshape(3, 4)
    assert np.array_equal(got[:, :3].reshape(-1), np.array(ref["R"], np.float32))
    assert np.array_equal(got[:, 3], np.array(ref["t"], np.float32))
    grid = [l for l in out if l.startswith("GRID")][0].split()
    assert [int(x) for x in grid[1:4]] == GOLD["dims"]["10"]          # VoxelGrid::Init, voxel_carver.cc:276-345
    assert grid[4] == GOLD["fnv1a64"]["res10_pos"] and grid[5] == "1"  # voxel centres, ids, reset, error returns
    assert [l for l in out if l.startswith("C2W")][0].split()[1] == "1"  # common.h:51-75 forms agree
    focal = [np.float32(x) for x in [l for l in out if l.startswith("FOCAL")][0].split()[1:]]
    assert focal[0] == synth.focal_from_fov_y(720, 60.0) and focal[1] == np.float32(639.5)


def test_c_abi_argument_validation_without_a_gpu():
    """Errors the reference reports as `return false` + LOGE come back as negative status codes with
    the same message, before any device is touched."""
    import ctypes as C
    lib = capi.load()
    ctx = C.c_void_p()

    def create(opt):
        rc = lib.vcy_create(C.byref(opt), 0, 0, -1, C.byref(ctx))
        return rc, lib.vcy_last_error().decode()

    rc, msg = create(B.bunny_option(10.0, capi.UpdateOption(voxel_max_update_num=0)))
    assert rc == -1 and "voxel_max_update_num must be positive" in msg      # voxel_carver.cc:376-379
    rc, msg = create(B.bunny_option(10.0, capi.UpdateOption(voxel_update_weight=0.0)))
    assert rc == -1 and "voxel_update_weight must be positive" in msg        # :380-384
    rc, msg = create(B.bunny_option(10.0, capi.UpdateOption(truncation_band=-1.0)))
    assert rc == -1 and "truncation_band must be positive" in msg            # :385-389
    rc, msg = create(B.bunny_option(0.0))
    assert rc == -1 and "resolution must be positive" in msg                 # :278-281
    bad = B.bunny_option(10.0)
    bad.bb_max[1] = bad.bb_min[1]
    rc, msg = create(bad)
    assert rc == -1 and "input bounding box is invalid" in msg               # :282-286
    rc, msg = create(B.bunny_option(10.0, capi.UpdateOption(voxel_update=7)))
    assert rc == -1
    # grid sizing without a context: n = (int)((bb_max - bb_min) / resolution)  (:292-296)
    mn, mx = B.bunny_bb()
    dims = (C.c_int32 * 3)()
    assert lib.vcy_compute_dims((C.c_float * 3)(*mn), (C.c_float * 3)(*mx), 10.0, dims) == 0
    assert list(dims) == GOLD["dims"]["10"]
    assert lib.vcy_compute_dims((C.c_float * 3)(*mn), (C.c_float * 3)(*mx), 2.5, dims) == 0
    assert list(dims) == GOLD["dims"]["2.5"]
    # host SDF entry points validate the ROI
    mask = np.zeros((8, 8), np.uint8)
    out = np.zeros((8, 8), np.float32)
    rmin, rmax = (C.c_int32 * 2)(0, 0), (C.c_int32 * 2)(8, 7)
    assert lib.vcy_make_sdf(mask.ctypes.data, 8, 8, rmin, rmax, 1, 0, 0.1, out.ctypes.data) == -1
    # null context
    assert lib.vcy_sync(None) == -2 and lib.vcy_reset(None) == -2
    assert lib.vcy_halo_bytes(None) == 0


def test_public_headers_compile_in_their_eigen_form():
    """The facade headers pick Eigen when <Eigen/Geometry> is on the include path (include/vacancy/common.h) --
    the branch every real user of the reference takes, and one this image cannot build: Eigen is an un-vendored
    submodule.  tests/eigen_decl/ holds DECLARATIONS of the Eigen names those headers use (no bodies: it pins no
    arithmetic and nothing can run against it); compiling the headers against it with -fsyntax-only catches
    syntax rot in that branch (it found a missing <cmath> in camera.h)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    r = subprocess.run([gxx, "-std=c++14", "-fsyntax-only", "-Wall", "-I", os.path.join(root, "tests", "eigen_decl"),
                        "-I", os.path.join(root, "include"),
                        os.path.join(root, "tests", "eigen_decl", "use_public_headers.cc")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]


def test_run_loops_of_the_fused_kernel_fetch_their_tables_through_scalar_loads():
    """A guard against a compiler trap that cost round 3 15 %: a wave-uniform value first computed inside a divergent
    branch reaches later uses through a phi the compiler must treat as divergent -- it then lives in a VGPR, the
    address of the per-view x tables with it, and the 24 scalar loads of every run loop become vector loads followed
    by vmcnt(0) waits.  Compiles the bench instantiations of carve_fused_kernel to assembly (no GPU needed) and checks
    every run-loop block (>= 8 ds_read2_b32): at most one vector load (the generic fallback's), none in the fast loops."""
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc), "hipcc is part of the image"
    src = os.path.join(root, "vacancy_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "fused.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                            "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
                            "-fno-slp-vectorize", "-DVCY_DEV_BENCH_KERNELS_ONLY", "-I", os.path.join(root, "include"),
                            "-I", src, "-S", "--cuda-device-only", "-o", out, os.path.join(src, "carve_fused_u8.hip")],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        txt = open(out).read()
    kernels = list(re.finditer(r"^(_ZN3vcy12_GLOBAL__N_118carve_fused_kernelI\w+):", txt, re.M))
    assert len(kernels) >= 4
    fast_blocks = 0
    for m in kernels:
        body = txt[m.end():txt.index(".Lfunc_end", m.end())]
        block = []
        for line in body.split("\n") + [".LBB_end:"]:
            if re.match(r"^\.LBB\w+:", line):
                ins = [x.strip() for x in block if x.startswith("\t") and not x.strip().startswith((";", "."))]
                if sum(i.startswith("ds_read2_b32") for i in ins) >= 8:
                    gload = sum(i.startswith("global_load") for i in ins)
                    assert gload <= 1, (m.group(1)[:80], gload)
                    if gload == 0:
                        fast_blocks += 1
                        assert sum(i.startswith("s_load") for i in ins) >= 3, m.group(1)[:80]
                block = []
            else:
                block.append(line)
    assert fast_blocks >= 8


def test_partition_layers_is_optimal_against_brute_force():
    """vcy_partition_layers (what vcy_plan_z_slabs cuts its estimate with; host arithmetic, no GPU): the contiguous
    partition with the smallest largest part -- checked against every partition for small cases -- whole brick layers,
    every slab at least one layer, the last bound = nz; and its argument checks."""
    import ctypes as C
    import itertools
    lib = capi.load()
    rng = np.random.RandomState(7)

    def cut(cost, parts, nz=None):
        cost = np.ascontiguousarray(cost, np.float64)
        nz = len(cost) * 8 if nz is None else nz
        b = np.zeros(parts + 1, np.int32)
        rc = lib.vcy_partition_layers(cost.ctypes.data_as(C.c_void_p), len(cost), parts, nz, b.ctypes.data_as(C.c_void_p))
        return rc, [int(x) for x in b]

    for trial in range(60):
        L = int(rng.randint(1, 11))
        parts = int(rng.randint(1, L + 1))
        cost = rng.rand(L) * (1 + 5 * (rng.rand(L) < 0.3))
        nz = L * 8 - int(rng.randint(0, 8))
        tail_single = L > 1 and nz - (L - 1) * 8 == 1  # a one-slice last layer cannot be a slab of its own
        rc, b = cut(cost, parts, nz)
        if tail_single and parts > L - 1:
            assert rc == capi.VCY_ERR_INVALID_ARG
            continue
        assert rc == 0
        assert b[0] == 0 and b[-1] == nz and all(x % 8 == 0 for x in b[:-1])
        assert all(b1 - b0 >= (2 if parts > 1 else 1) for b0, b1 in zip(b, b[1:])), (b, nz)  # (halo: >= 2 slices)
        got = max(cost[b[s] // 8:(b[s + 1] + 7) // 8].sum() for s in range(parts))
        best = min(max(cost[i:j].sum() for i, j in zip((0,) + cuts, cuts + (L,)))
                   for cuts in itertools.combinations(range(1, L), parts - 1)
                   if not (tail_single and cuts and cuts[-1] == L - 1))
        assert got <= best * (1 + 1e-9), (cost, parts, b, got, best)
    # equal costs, parts dividing the layers: equal slabs; a heavy middle: thin slabs there
    assert cut(np.ones(16), 4)[1] == [0, 32, 64, 96, 128]
    b = cut([1, 1, 1, 1, 6, 6, 1, 1, 1, 1], 4)[1]
    assert b == [0, 32, 40, 48, 80]
    for bad in ((np.ones(4), 5, None), (np.ones(4), 0, None), (np.array([1.0, np.nan]), 1, None), (np.ones(4), 2, 40),
                (np.ones(4), 2, 24)):
        assert cut(*bad)[0] == capi.VCY_ERR_INVALID_ARG


def test_voxel_cubes_without_a_gpu_equals_the_oracle():
    """vcy_voxel_cubes -- the serial half of ExtractVoxel that ShardedVoxelCarver runs on the concatenated id lists of its
    slabs (ONE cube mesh drifting from kept voxel to kept voxel, extract_voxel.cc:290-311) -- is host arithmetic: the
    oracle's carved bunny, its keep predicate evaluated in numpy, cubes from the library == the oracle's ExtractVoxel."""
    from vacancy_amd import carver
    opt = B.bunny_option(10.0)
    views = B.bunny_views(lambda t, q: O.affine_inverse(O.pose_from_tum(t, q)))
    g = O.OracleGrid(opt)
    for i, m in enumerate(B.load_masks()):
        g.carve(views[i], O.make_sdf(m))
    sdf, cnt = g.download()
    ids = np.nonzero(~((sdf > 0) | (cnt < 1)))[0].astype(np.int64)   # extract_voxel.cc:283-286
    got, want = carver.voxel_cubes(opt, ids), g.extract_voxel(False)
    assert len(want["vertices"]) == 683400 == len(got["vertices"])
    assert np.array_equal(got["faces"], want["faces"])
    assert np.array_equal(got["vertices"].view(np.uint32), want["vertices"].view(np.uint32))
    empty = carver.voxel_cubes(opt, np.zeros(0, np.int64))
    assert len(empty["vertices"]) == 0 and len(empty["faces"]) == 0
    with pytest.raises(RuntimeError):
        carver.voxel_cubes(opt, np.array([g.n], np.int64))  # outside the grid
    # a list long enough for the threaded fill (>= 65536 kept voxels: the chain is walked serially, the 24 vertices and 12
    # triangles of every voxel are written by host threads), ascending as a scan leaves it and in an order no scan
    # produces (the row of a voxel is then found by division): every voxel of the resolution-5 grid kept
    opt5 = B.bunny_option(5.0)
    g5 = O.OracleGrid(opt5)
    rng = np.random.RandomState(7)
    g5.upload(-rng.rand(g5.n).astype(np.float32), np.ones(g5.n, np.int32))
    want5 = g5.extract_voxel(False)
    got5 = carver.voxel_cubes(opt5, np.arange(g5.n, dtype=np.int64))
    assert g5.n >= 65536 and len(got5["vertices"]) == 24 * g5.n
    assert np.array_equal(got5["faces"], want5["faces"])
    assert np.array_equal(got5["vertices"].view(np.uint32), want5["vertices"].view(np.uint32))
    perm = rng.permutation(g5.n)[:70000].astype(np.int64)
    one_by_one = carver.voxel_cubes(opt5, perm)
    # (the same chain on one thread: a list below the threading threshold cannot be compared, so compare with numpy's
    # restatement of the six running values)
    h = np.float32(5.0) / np.float32(2)
    pos = g5.positions()[perm]
    lo, hi = np.full(3, -h, np.float32), np.full(3, h, np.float32)
    for t in (0, 1, 69999):
        lo_t, hi_t = lo.copy(), hi.copy()
        if t:
            lo_t, hi_t = np.full(3, -h, np.float32), np.full(3, h, np.float32)
            for q in range(t):
                lo_t = (lo_t + pos[q]) + -pos[q]
                hi_t = (hi_t + pos[q]) + -pos[q]
        v = one_by_one["vertices"][24 * t:24 * t + 24]
        assert np.array_equal(np.unique(v[:, 0]), np.unique(np.array([lo_t[0] + pos[t][0], hi_t[0] + pos[t][0]], np.float32)))
        assert np.array_equal(np.unique(v[:, 2]), np.unique(np.array([lo_t[2] + pos[t][2], hi_t[2] + pos[t][2]], np.float32)))


def test_voxel_cubes_into_the_callers_arrays():
    """vcy_voxel_cubes_into: the callback is called once with the mesh's sizes (never for an empty mesh) and the arrays it
    returns hold what vcy_voxel_cubes returns -- 16-byte aligned ones through the streaming-store fill, misaligned ones
    through the scalar loop."""
    import ctypes as C
    from vacancy_amd import capi, carver
    import bunny_data as B
    opt = B.bunny_option(2.3)
    rng = np.random.RandomState(5)
    ids = np.sort(rng.choice(232 * 233 * 184 // 8, 90000, replace=False)).astype(np.int64)
    want = carver.voxel_cubes(opt, ids)
    got = carver.voxel_cubes_into(opt, ids)
    assert got["calls"] == 1
    assert np.array_equal(got["vertices"].view(np.uint32), want["vertices"].view(np.uint32))
    assert np.array_equal(got["faces"], want["faces"])
    empty = carver.voxel_cubes_into(opt, np.zeros(0, np.int64))
    assert empty["calls"] == 0 and len(empty["vertices"]) == 0
    # arrays 4 bytes off a 16-byte boundary, and a callback that refuses
    lib = capi.load()
    hold = {}

    def misaligned(user, nv, nf, pv, pf):
        hold["v"] = np.empty(3 * nv + 4, np.float32)
        hold["f"] = np.empty(3 * nf + 4, np.int32)
        ov = (4 - (hold["v"].ctypes.data % 16) // 4 + 1) % 4 or 1
        of = (4 - (hold["f"].ctypes.data % 16) // 4 + 1) % 4 or 1
        hold["ov"], hold["of"], hold["nv"], hold["nf"] = ov, of, nv, nf
        assert (hold["v"].ctypes.data + 4 * ov) % 16 != 0
        pv[0] = C.cast(hold["v"].ctypes.data + 4 * ov, C.POINTER(C.c_float))
        pf[0] = C.cast(hold["f"].ctypes.data + 4 * of, C.POINTER(C.c_int32))
        return 0

    cb = capi.MeshArraysFn(misaligned)
    assert lib.vcy_voxel_cubes_into(C.byref(opt), len(ids), ids.ctypes.data_as(C.c_void_p), cb, None) == 0, carver.last_error()
    v = hold["v"][hold["ov"]:hold["ov"] + 3 * hold["nv"]].reshape(-1, 3)
    f = hold["f"][hold["of"]:hold["of"] + 3 * hold["nf"]].reshape(-1, 3)
    assert np.array_equal(v.view(np.uint32), want["vertices"].view(np.uint32)) and np.array_equal(f, want["faces"])
    refuse = capi.MeshArraysFn(lambda user, nv, nf, pv, pf: 1)
    assert lib.vcy_voxel_cubes_into(C.byref(opt), len(ids), ids.ctypes.data_as(C.c_void_p), refuse, None) != 0


def test_voxel_cubes_speculative_chunks_are_checked_and_redone():
    """vcy_voxel_cubes runs the serial cube chain in chunks on host threads, each from a GUESSED incoming state, and checks
    afterwards that every guess equals what the predecessor really left.  (1) a resolution whose half is not a dyadic
    number, so that the cube really drifts from voxel to voxel; (2) in a subprocess with the test hook that makes every
    guess wrong by one ulp: the check must catch it, the chunks are done again, and the mesh still equals the oracle's."""
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import bunny_data as B, oracle_lib as O
from vacancy_amd import carver
opt = B.bunny_option(2.3)
g = O.OracleGrid(opt)
rng = np.random.RandomState(3)
sdf = rng.uniform(-1, 1, g.n).astype(np.float32)
g.upload(sdf, np.ones(g.n, np.int32))
ids = np.nonzero(sdf <= 0)[0].astype(np.int64)
assert len(ids) > 200000
want = g.extract_voxel(False)
got = carver.voxel_cubes(opt, ids)
assert np.array_equal(got["faces"], want["faces"])
assert np.array_equal(got["vertices"].view(np.uint32), want["vertices"].view(np.uint32))
h = np.float32(2.3) / np.float32(2)
v = got["vertices"].reshape(-1, 24, 3)
width = v[:, :, 0].max(1) - v[:, :, 0].min(1)
print("DRIFT", int((width != width[0]).sum()))
""" % (ROOT, os.path.join(ROOT, "tests"))
    for hook in (False, True):
        env = dict(os.environ, VCY_XV_TIMING="1")
        if hook:
            env["VCY_TEST_XV_BAD_GUESS"] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        redone = [int(x) for x in re.findall(r"check \+ (\d+) chunks done again", r.stderr)]
        assert redone, r.stderr[-500:]
        assert (max(redone) > 0) == hook, (hook, redone)
        assert int(re.search(r"DRIFT (\d+)", r.stdout).group(1)) > 0   # the cube's width is not constant: it does drift


def _png_decode_filter0(path):
    """An independent decoder for what WritePng8 emits (8-bit, non-interlaced, every row filter 0): chunk walk with CRC
    check, zlib inflate.  Returns (width, height, channels, pixels)."""
    import struct
    import zlib
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr, seen_end = 8, b"", None, False
    while pos < len(raw):
        (ln,), typ = struct.unpack(">I", raw[pos:pos + 4]), raw[pos + 4:pos + 8]
        body = raw[pos + 8:pos + 8 + ln]
        (crc,) = struct.unpack(">I", raw[pos + 8 + ln:pos + 12 + ln])
        assert crc == (zlib.crc32(typ + body) & 0xffffffff), typ
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            seen_end = True
        pos += 12 + ln
    assert seen_end and pos == len(raw)
    w, h, depth, ctype, comp, flt, inter = ihdr
    assert (depth, comp, flt, inter) == (8, 0, 0, 0)
    ch = {0: 1, 4: 2, 2: 3, 6: 4}[ctype]
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * ch + 1)
    assert not rows[:, 0].any()
    return w, h, ch, rows[:, 1:].reshape(h, w, ch)


def test_host_outputs_are_read_back_byte_for_byte(tmp_path):
    """Rows f3 / f4 of SURVEY section 8: what the facade WRITES.  `Mesh::WritePlyBinary` (ours; the reference has only the
    ASCII writer, mesh.cc:583-631) is parsed back -- header, 12-byte vertices, 13-byte face records -- to the mesh it was
    given and to what the ASCII writer says about the same mesh; `SignedDistance2Color` (voxel_carver.cc:239-267) is
    compared with a numpy restatement on the SDF of every bunny mask; `Image::WritePng` (image.h:103-118; zlib here, stb
    in the reference) is decoded by a decoder of the test's own, by Pillow when it is there, and by the facade's Load."""
    import subprocess
    _build_host()
    out = str(tmp_path)
    masks = B.load_masks()
    sdfs = []
    for i, m in enumerate(masks):
        # (view 3: not normalised, so that values beyond both ends of the colour ramp occur)
        sdfs.append(O.make_sdf(m, None, None, i != 3, False, 0.1) * (np.float32(0.01) if i == 3 else np.float32(1)))
        sdfs[-1].astype(np.float32).tofile(os.path.join(out, "sdf_%d.f32" % i))
    lines = subprocess.run([os.path.join(ROOT, "vacancy_amd", "host", "host_selftest"), B.BUNNY, "io", out],
                           check=True, capture_output=True, text=True).stdout.splitlines()
    assert [l for l in lines if l.startswith("PLY")][0].split() == ["PLY", "1", "5000", "9000"]
    # ---- the mesh the selftest builds, restated
    i5 = np.arange(5000)
    V = np.stack([np.float32(0.1) * i5.astype(np.float32) - np.float32(250.0),
                  np.float32(1.0) / (np.float32(1.0) + i5.astype(np.float32)),
                  (i5 % 7).astype(np.float32) * np.float32(-1234.5678)], 1).astype(np.float32)
    i9 = np.arange(9000)
    F = np.stack([i9 % 5000, (i9 * 7 + 1) % 5000, (i9 * 13 + 2) % 5000], 1).astype(np.int32)
    # ---- binary PLY
    raw = open(os.path.join(out, "mesh_binary.ply"), "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    assert raw[:end].decode() == ("ply\nformat binary_little_endian 1.0\nelement vertex 5000\nproperty float x\n"
                                  "property float y\nproperty float z\nelement face 9000\n"
                                  "property list uchar int vertex_indices\nend_header\n")
    assert len(raw) == end + 12 * 5000 + 13 * 9000
    v = np.frombuffer(raw, "<f4", 3 * 5000, end).reshape(-1, 3)
    rec = np.frombuffer(raw, np.uint8, 13 * 9000, end + 12 * 5000).reshape(-1, 13)
    assert (rec[:, 0] == 3).all()
    f = np.ascontiguousarray(rec[:, 1:]).view("<i4").reshape(-1, 3)
    assert np.array_equal(v.view(np.uint32), V.view(np.uint32)) and np.array_equal(f, F)
    empty = open(os.path.join(out, "mesh_empty.ply"), "rb").read()
    assert empty.endswith(b"end_header\n") and b"element vertex 0\n" in empty and b"element face 0\n" in empty
    # ---- the ASCII file of the same mesh: the reference's layout, %g digits (ostream default, mesh.cc:613-628)
    txt = open(os.path.join(out, "mesh_ascii.ply")).read().split("\n")
    assert txt[:9] == ["ply", "format ascii 1.0", "element vertex 5000", "property float x", "property float y",
                       "property float z", "element face 9000", "property list uchar int vertex_indices", "end_header"]
    assert txt[9] == "%g %g %g " % tuple(float(x) for x in V[0]) and txt[9 + 4999] == "%g %g %g " % tuple(float(x) for x in V[4999])
    assert txt[9 + 5000] == "3 0 1 2 " and txt[9 + 5000 + 8999] == "3 %d %d %d " % tuple(F[8999]) and txt[-1] == ""
    va = np.array([[float(t) for t in l.split()] for l in txt[9:9 + 5000]])
    assert np.allclose(va, V, rtol=1e-5, atol=0)  # six significant digits
    # ---- SignedDistance2Color against numpy, PNG against three decoders
    rows = {int(l.split()[1]): l.split()[2:] for l in lines if l.startswith("PNGRT")}
    assert len(rows) == 6 and all(r == ["1", "1", "1"] for r in rows.values()), rows
    assert [l for l in lines if l.startswith("PNGEMPTY")][0].split()[1] == "0"   # empty image: WritePng returns false
    try:
        from PIL import Image as PILImage
    except ImportError:
        PILImage = None
    for i, sdf in enumerate(sdfs):
        lo, hi = (np.float32(-0.25), np.float32(0.125)) if i == 3 else (np.float32(-1.0), np.float32(1.0))
        d = sdf.astype(np.float32)
        kp = np.minimum(np.maximum((hi - d) / hi, np.float32(0)), np.float32(1))          # voxel_carver.cc:251-252
        kn = np.minimum(np.maximum((d - lo) / (-lo), np.float32(0)), np.float32(1))       # :258-259
        cp = (np.float32(255) * kp).astype(np.uint8)                                      # truncation, as static_cast does
        cn = (np.float32(255) * kn).astype(np.uint8)
        pos = d > 0
        want = np.stack([np.where(pos, 255, cn), np.where(pos, cp, cn), np.where(pos, cp, 255)], 2).astype(np.uint8)
        got = np.fromfile(os.path.join(out, "vis_%d.rgb" % i), np.uint8).reshape(want.shape)
        assert np.array_equal(got, want), i
        if i == 3:  # both clamps fired, and the ramp in between is there
            assert (cp[pos] == 0).any() and (cn[~pos] == 0).any() and len(np.unique(want)) > 20
        w, h, ch, px = _png_decode_filter0(os.path.join(out, "vis_%d.png" % i))
        assert (w, h, ch) == (want.shape[1], want.shape[0], 3) and np.array_equal(px, want)
        w, h, ch, px = _png_decode_filter0(os.path.join(out, "mask_%d.png" % i))
        assert ch == 1 and np.array_equal(px[:, :, 0], masks[i])
        if PILImage is not None:
            assert np.array_equal(np.asarray(PILImage.open(os.path.join(out, "vis_%d.png" % i))), want)
            assert np.array_equal(np.asarray(PILImage.open(os.path.join(out, "mask_%d.png" % i))), masks[i])
