"""Pins the CPU oracle (oracle/vacancy_oracle.cc) against the known answers the survey
recorded from the reference's sources on data/ bunny (copied as
tests/golden/appendix_c.json).  The reference has no tests or golden files of its own."""
import json
import os

import numpy as np
import pytest

import bunny_data as B
import oracle_lib as O
from hashing import fnv1a64
from vacancy_amd.capi import UpdateOption

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_c.json")))
LOWEST = np.finfo(np.float32).min


def oracle_w2c(t, q):
    return O.affine_inverse(O.pose_from_tum(t, q))


@pytest.fixture(scope="module")
def views():
    return B.bunny_views(oracle_w2c)


@pytest.fixture(scope="module")
def masks():
    return B.load_masks()


def test_voxel_record_is_40_bytes(oracle):
    assert oracle.orc_sizeof_voxel() == 40


def test_w2c_matches_reference(views):
    for i, ref in GOLD["w2c_f32"].items():
        m = np.array(list(views[int(i)].w2c), np.float32).reshape(3, 4)
        np.testing.assert_allclose(m[:, 3], np.array(ref["t"], np.float32), rtol=0, atol=0)
        if "R" in ref:
            np.testing.assert_array_equal(m[:, :3].reshape(-1), np.array(ref["R"], np.float32))


def test_sdf_image_stats_and_hash(masks):
    for i, (mn, mx, total) in enumerate(GOLD["sdf_stats_default"]):
        sdf = O.make_sdf(masks[i])
        assert sdf.min() == np.float32(mn)
        assert sdf.max() == np.float32(mx)
        assert abs(float(sdf.astype(np.float64).sum()) - total) < 1e-3 * max(1.0, abs(total)) * 1e-2
    assert fnv1a64(O.make_sdf(masks[0])) == GOLD["fnv1a64"]["sdf_view0"]


@pytest.mark.parametrize("mode", list(B.MODES))
def test_per_view_known_answers(mode, views, masks):
    uo = UpdateOption(**B.MODES[mode])
    g = O.OracleGrid(B.bunny_option(10.0, uo))
    assert list(g.dims) == GOLD["dims"]["10"]
    for i, row in enumerate(GOLD["modes"][mode]):
        sdf = O.make_sdf(masks[i], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
        g.carve(views[i], sdf)
        s, u = g.download()
        upd = u >= 1
        mesh = g.marching_cubes(0.0, True)
        got = [int(upd.sum()), int((s[upd] < 0).sum()), float(s[upd].astype(np.float64).sum()),
               int(u.sum()), int((sdf == LOWEST).sum()), len(mesh["vertices"]), len(mesh["faces"])]
        for k in (0, 1, 3, 4, 5, 6):
            assert got[k] == row[k], (mode, i, k, got, row)
        # the table prints 9 significant digits
        assert abs(got[2] - row[2]) <= 5e-9 * abs(row[2]) + 1e-6, (mode, i, got[2], row[2])
    if mode == "default":
        np.testing.assert_allclose(mesh["vertices"].astype(np.float64).sum(0),
                                   GOLD["final_vertex_sums"], rtol=0, atol=1e-5)
        m2 = g.marching_cubes(0.0, False)
        assert [len(m2["vertices"]), len(m2["faces"])] == GOLD["final_nointerp"]
        ev = g.extract_voxel(False)  # ExtractVoxel: 28 475 cubes x 24 / 12 (Appendix C)
        assert (len(ev["vertices"]), len(ev["faces"])) == (683400, 341700)
        assert fnv1a64(g.positions()) == GOLD["fnv1a64"]["res10_pos"]
        assert fnv1a64(s) == GOLD["fnv1a64"]["res10_sdf"]


def test_finer_grid_counts_and_hashes(views, masks):
    g = O.OracleGrid(B.bunny_option(5.0))
    for i in range(6):
        g.carve(views[i], O.make_sdf(masks[i]))
    s, _ = g.download()
    mesh = g.marching_cubes()
    assert [len(mesh["vertices"]), len(mesh["faces"])] == GOLD["mesh_counts"]["5"]
    assert fnv1a64(g.positions()) == GOLD["fnv1a64"]["res5_pos"]
    assert fnv1a64(s) == GOLD["fnv1a64"]["res5_sdf"]


def test_option_validation_matches_reference_init():
    # VoxelCarver::Init / VoxelGrid::Init return false (voxel_carver.cc:278-287,376-389)
    bad = [dict(voxel_max_update_num=0), dict(voxel_update_weight=0.0), dict(truncation_band=0.0)]
    for kw in bad:
        with pytest.raises(ValueError):
            O.OracleGrid(B.bunny_option(10.0, UpdateOption(**kw)))
    with pytest.raises(ValueError):
        O.OracleGrid(B.bunny_option(0.0))
    opt = B.bunny_option(10.0)
    opt.bb_max[0] = opt.bb_min[0]
    with pytest.raises(ValueError):
        O.OracleGrid(opt)
