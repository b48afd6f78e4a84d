"""data/ bunny fixture of the reference (tests/golden/bunny: 6 masks + tumpose.txt) and the
constants examples.cc hard-codes (reference examples.cc:87-115)."""
import os

import numpy as np

from vacancy_amd.capi import CarverOption, UpdateOption, make_view

HERE = os.path.dirname(os.path.abspath(__file__))
BUNNY = os.path.join(HERE, "golden", "bunny")

BB_MIN = (-250.000000 - 20.0, -344.586151 - 20.0, -129.982697 - 20.0)
BB_MAX = (250.000000 + 20.0, 150.542343 + 20.0, 257.329224 + 20.0)
WIDTH, HEIGHT = 320, 240
PRINCIPAL = (159.3, 127.65)
FOCAL = (258.65, 258.25)


def bunny_bb():
    """bb +- 20 computed in float32 like examples.cc:91-99 (option.bb_min[0] -= bb_offset)."""
    mn = np.array([-250.000000, -344.586151, -129.982697], np.float32) - np.float32(20.0)
    mx = np.array([250.000000, 150.542343, 257.329224], np.float32) + np.float32(20.0)
    return mn, mx


def load_masks():
    npz = os.path.join(BUNNY, "masks.npz")
    return [m for m in np.load(npz)["masks"]]


def load_tum():
    rows = []
    for line in open(os.path.join(BUNNY, "tumpose.txt")):
        p = line.split(" ")
        if len(p) != 8:
            continue
        # std::atof on each field (examples.cc:36-48)
        rows.append((int(p[0]), [float(x) for x in p[1:4]], [float(x) for x in p[4:8]]))
    return rows


def bunny_option(resolution=10.0, update_option=None):
    mn, mx = bunny_bb()
    return CarverOption(bb_min=[float(x) for x in mn], bb_max=[float(x) for x in mx],
                        resolution=resolution, update_option=update_option or UpdateOption())


def bunny_views(w2c_from_pose):
    """w2c_from_pose(t, q) -> 3x4 float64 w2c.  Returns the 6 vcy_view structs."""
    views = []
    for _, t, q in load_tum():
        w2c = np.asarray(w2c_from_pose(t, q), np.float64).astype(np.float32)
        views.append(make_view(w2c, np.float32(FOCAL[0]), np.float32(FOCAL[1]),
                               np.float32(PRINCIPAL[0]), np.float32(PRINCIPAL[1]), WIDTH, HEIGHT))
    return views


MODES = {
    "default": dict(),
    "tsdf": dict(voxel_update=1, use_truncation=True, truncation_band=0.1),
    "nn_outside": dict(sdf_interp=0, update_outside=1),
    "max_trunc": dict(use_truncation=True, truncation_band=0.1),
}
