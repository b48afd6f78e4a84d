"""The one arithmetic assumption on the path that nothing the reference ships can pin: how Eigen sums the three
products of a row of `Affine3f * Vector3f` (voxel_carver.cc:453; Eigen is an un-vendored, unpinned submodule).
This test pins NOTHING about the reference.  It keeps tests/golden/association_exposure.json honest: the numbers
DESIGN.md section 2 quotes for what the OTHER summation orders would change (a test-only switch of the oracle,
orc_set_association; the product has none) are re-measured here for the small scenes."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import make_association_exposure as X  # noqa: E402
import oracle_lib as O  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "association_exposure.json")))


def test_recorded_exposure_is_what_the_oracle_measures_now():
    for name, opt, views, sdfs in X.scenes(full=False):
        assert X.exposure(opt, views, sdfs) == GOLD["scenes"][name], name


def test_the_switch_is_off_by_default_and_after_use():
    """Every other test relies on the default order: the switch must never leak."""
    name, opt, views, sdfs = X.scenes(full=False)[2]  # the 48^3 sphere: general rotations, the orders differ
    s0 = X.run(opt, views, sdfs, 0)[0]
    s1 = X.run(opt, views, sdfs, 1)[0]
    assert (s0.view(np.uint32) != s1.view(np.uint32)).any()
    g = O.OracleGrid(opt)  # no switch touched
    for v, s in zip(views, sdfs):
        g.carve(v, s)
    assert np.array_equal(g.download()[0].view(np.uint32), s0.view(np.uint32))


def test_what_design_md_says_about_it():
    sc = GOLD["scenes"]
    # data/ bunny: the six cameras are axis-aligned (one non-zero entry per row of R), two of the three products
    # are +-0 and every order gives the same sum -- the bunny cannot tell the orders apart, for better or worse
    for name in ("bunny_res10_default", "bunny_res10_tsdf", "bunny_res5_default", "bunny_res2.5_default"):
        a = sc[name]["alternatives"]["1"]
        assert a["sdf_bits_differ"] == 0 and a["mesh"] == sc[name]["mesh"]
    # general rotations (the synthetic sphere): a few per cent of the voxels change in the last bits, no voxel changes
    # sign or update_num, the mesh keeps its counts and moves by less than the north star's 1e-4
    a = sc["sphere48_default"]["alternatives"]["1"]
    assert 0 < a["sdf_bits_differ"] < 0.1 * sc["sphere48_default"]["voxels"]
    assert a["update_num_differ"] == 0 and a["sign_differs"] == 0 and a["mesh"] == sc["sphere48_default"]["mesh"]
    assert a["max_abs_vertex_difference"] < 1e-4
