"""Writes tests/golden/association_exposure.json: how much of the oracle's output depends on the ONE arithmetic
assumption nothing the reference ships can pin -- the order in which Eigen sums the three products of a row of
`Affine3f * Vector3f` (voxel_carver.cc:453).  The oracle (and the HIP kernels) use t + (c0 + (c1 + c2)); this script
carves the same scenes with the other two orders, t + ((c0 + c1) + c2) and ((t + c0) + c1) + c2 (a TEST-ONLY switch of
the oracle, orc_set_association), and records how many voxels end with other sdf bits / update_num and how the mesh
counts move.  It pins nothing; it turns "unpinned" into numbers.

Run from the repo root:  python tests/golden/make_association_exposure.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import bunny_data as B  # noqa: E402
import oracle_lib as O  # noqa: E402
from vacancy_amd import synth  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402

ASSOC = {0: "t + (c0 + (c1 + c2))  [oracle and device]", 1: "t + ((c0 + c1) + c2)", 2: "((t + c0) + c1) + c2"}


def run(option, views, sdfs, assoc):
    lib = O.load()
    lib.orc_set_association(assoc)
    try:
        g = O.OracleGrid(option)
        for v, s in zip(views, sdfs):
            g.carve(v, s)
        s, u = g.download()
        m = g.marching_cubes(0.0, True)
        out = (s, u, m["vertices"], len(m["faces"]))
        g.close()
        return out
    finally:
        lib.orc_set_association(0)


def exposure(option, views, sdfs):
    s0, u0, v0, f0 = run(option, views, sdfs, 0)
    rec = {"voxels": int(s0.size), "mesh": [int(len(v0)), int(f0)], "alternatives": {}}
    for a in (1, 2):
        s, u, v, f = run(option, views, sdfs, a)
        diff = s.view(np.uint32) != s0.view(np.uint32)
        d = np.abs(s[diff].astype(np.float64) - s0[diff].astype(np.float64))
        alt = {"order": ASSOC[a], "sdf_bits_differ": int(diff.sum()), "update_num_differ": int((u != u0).sum()),
               "sign_differs": int(((s < 0) != (s0 < 0)).sum()),
               "max_abs_sdf_difference": float(d.max()) if d.size else 0.0,
               "mesh": [int(len(v)), int(f)]}
        if len(v) == len(v0):
            alt["max_abs_vertex_difference"] = float(np.abs(v.astype(np.float64) - v0.astype(np.float64)).max())
        rec["alternatives"][str(a)] = alt
    return rec


def scenes(full=True):
    """(name, option, views, sdfs): data/ bunny at the resolutions of SURVEY Appendix C, the 48^3 sphere of the GPU tests."""
    views = B.bunny_views(lambda t, q: O.affine_inverse(O.pose_from_tum(t, q)))
    masks = B.load_masks()
    sdfs = [O.make_sdf(m) for m in masks]
    out = [("bunny_res10_default", B.bunny_option(10.0), views, sdfs)]
    tsdf = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1)
    out.append(("bunny_res10_tsdf", B.bunny_option(10.0, tsdf), views,
                [O.make_sdf(m, use_truncation=True, band=0.1) for m in masks]))
    sv, sm = synth.sphere_views(48, 8, 160, 120)
    out.append(("sphere48_default", synth.sphere_option(48), sv, [O.make_sdf(m) for m in sm]))
    if full:
        out.append(("bunny_res5_default", B.bunny_option(5.0), views, sdfs))
        out.append(("bunny_res2.5_default", B.bunny_option(2.5), views, sdfs))
    return out


def main():
    rec = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "orders": ASSOC, "scenes": {}}
    for name, opt, views, sdfs in scenes(True):
        rec["scenes"][name] = exposure(opt, views, sdfs)
        print(name, json.dumps(rec["scenes"][name]))
    with open(os.path.join(HERE, "association_exposure.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
