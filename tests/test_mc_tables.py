"""Row a8 pinned to the reference itself: tests/golden/mc_tables.bin was dumped from the reference's
own src/vacancy/marching_cubes_lut.cc, compiled unmodified into oracle/_ref (oracle/ref.mk,
tests/golden/make_mc_tables.py).  The product's case data (include/vacancy_mc_cases.inc, which both
the HIP kernels and the oracle unpack) and the oracle's in-memory tables must equal it entry for
entry; where oracle/_ref exists (the build container) the fixture is re-checked against it live."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden():
    t = np.fromfile(os.path.join(ROOT, "tests", "golden", "mc_tables.bin"), "<i4")
    assert t.shape == (256 + 256 * 16,)
    return t[:256].copy(), t[256:].reshape(256, 16).copy()


def test_case_data_file_equals_reference_tables():
    edge, tri = _golden()
    txt = open(os.path.join(ROOT, "include", "vacancy_mc_cases.inc")).read()
    strs = re.findall(r'"([0-9a-b]*)"', txt)
    assert len(strs) == 256
    ec = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
    for c, s in enumerate(strs):
        row = [int(ch, 16) for ch in s] + [-1] * (16 - len(s))
        assert row == list(tri[c]), c
        # the kernels derive kEdgeTable[c] as "edges whose two corners differ" (mc_kernels.hip cut_edges)
        cut = sum(1 << k for k, (a, b) in enumerate(ec) if ((c >> a) ^ (c >> b)) & 1)
        assert cut == edge[c], c
        # ... and the oracle as "edges the case's triangles use"
        used = 0
        for e in row:
            if e >= 0:
                used |= 1 << e
        assert used == edge[c], c


def test_oracle_tables_equal_reference_tables(oracle):
    edge, tri = _golden()
    e = np.zeros(256, np.int32)
    t = np.zeros((256, 16), np.int32)
    oracle.orc_mc_tables(e.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p))
    assert np.array_equal(e, edge) and np.array_equal(t, tri)


def test_fixture_equals_live_reference_build_when_present():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_mc_lut.so")
    if not os.path.exists(so):
        import pytest
        pytest.skip("oracle/_ref not built here (the reference sources are absent on this box)")
    lib = C.CDLL(so)
    edge, tri = _golden()
    for c in range(256):
        assert lib.ref_edge_table(c) == edge[c]
        for k in range(16):
            assert lib.ref_tri_table(c, k) == tri[c][k]
