"""ctypes wrapper of oracle/libvacancy_oracle.so -- test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from vacancy_amd.capi import CarverOption, Mesh, UpdateOption, View  # POD structs only

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libvacancy_oracle.so")

_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def load():
    global _lib
    if _lib is not None:
        return _lib
    build()  # incremental make: never test against a stale oracle
    lib = C.CDLL(LIB)
    P, vp = C.POINTER, C.c_void_p
    lib.orc_grid_create.restype = vp
    lib.orc_grid_create.argtypes = [P(CarverOption)]
    lib.orc_set_num_threads.argtypes = [C.c_int]
    lib.orc_set_association.argtypes = [C.c_int]
    lib.orc_mc_tables.argtypes = [vp, vp]
    lib.orc_grid_destroy.argtypes = [vp]
    lib.orc_grid_from_positions.restype = vp
    lib.orc_grid_from_positions.argtypes = [P(UpdateOption), vp, C.c_int]
    lib.orc_axis_positions.argtypes = [C.c_float, C.c_float, C.c_float, vp, P(C.c_int)]
    lib.orc_grid_dims.argtypes = [vp, P(C.c_int32)]
    lib.orc_grid_download.argtypes = [vp, vp, vp]
    lib.orc_grid_upload.argtypes = [vp, vp, vp]
    lib.orc_grid_positions.argtypes = [vp, vp]
    lib.orc_carve.restype = C.c_double
    lib.orc_carve.argtypes = [vp, P(View), vp]
    lib.orc_distance_transform_l1.argtypes = [vp, C.c_int, C.c_int, P(C.c_int32), P(C.c_int32), vp]
    lib.orc_make_sdf.argtypes = [vp, C.c_int, C.c_int, P(C.c_int32), P(C.c_int32), C.c_int, C.c_int,
                                 C.c_float, vp]
    lib.orc_marching_cubes.restype = C.c_double
    lib.orc_marching_cubes.argtypes = [vp, C.c_double, C.c_int, P(Mesh)]
    lib.orc_marching_cubes_slab.restype = C.c_double
    lib.orc_marching_cubes_slab.argtypes = [vp, C.c_double, C.c_int, C.c_int, C.c_int, P(Mesh)]
    lib.orc_extract_voxel.restype = C.c_double
    lib.orc_extract_voxel.argtypes = [vp, C.c_int, P(Mesh)]
    lib.orc_mesh_free.argtypes = [P(Mesh)]
    lib.orc_pose_from_tum.argtypes = [vp, vp, vp]
    lib.orc_affine_inverse.argtypes = [vp, vp]
    lib.orc_lookat_c2w.argtypes = [vp, vp, vp, vp]
    lib.orc_affine_to_float.argtypes = [vp, vp]
    lib.orc_focal_from_fov_y.restype = C.c_float
    lib.orc_focal_from_fov_y.argtypes = [C.c_int, C.c_float]
    # The OpenMP loop over z with one thread per VISIBLE core on a box whose cgroup grants fewer (256 visible, 16
    # granted on the GPU boxes) spends its time in contended barriers: 0.12 s per carve of a 40^3 grid, 36 s for a
    # 300-view test.  Results do not depend on the thread count.
    usable = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            usable = max(1, min(usable, int(round(int(quota) / float(period)))))
    except Exception:
        pass
    lib.orc_set_num_threads(int(os.environ.get("VCY_ORACLE_THREADS", usable)))
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def axis_positions(bb_min, bb_max, resolution, n_max):
    out = np.empty(n_max, np.float32)
    n = C.c_int(0)
    load().orc_axis_positions(bb_min, bb_max, resolution, _p(out), C.byref(n))
    return out[: n.value].copy()


class OracleGrid:
    def __init__(self, option, positions=None):
        self.lib = load()
        if positions is not None:
            self._pos = np.ascontiguousarray(positions, np.float32)
            self.h = self.lib.orc_grid_from_positions(C.byref(option.update_option), _p(self._pos),
                                                      len(self._pos))
        else:
            self.h = self.lib.orc_grid_create(C.byref(option))
        if not self.h:
            raise ValueError("oracle: invalid option (reference Init() returns false)")
        d = (C.c_int32 * 3)()
        self.lib.orc_grid_dims(self.h, d)
        self.dims = tuple(d)
        self.n = self.dims[0] * self.dims[1] * self.dims[2]

    def close(self):
        if self.h:
            self.lib.orc_grid_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def carve(self, view, sdf):
        sdf = np.ascontiguousarray(sdf, dtype=np.float32)
        return self.lib.orc_carve(self.h, C.byref(view), _p(sdf))

    def download(self):
        s = np.empty(self.n, np.float32)
        u = np.empty(self.n, np.int32)
        self.lib.orc_grid_download(self.h, _p(s), _p(u))
        return s, u

    def upload(self, sdf, update_num):
        s = np.ascontiguousarray(sdf, np.float32)
        u = np.ascontiguousarray(update_num, np.int32)
        self.lib.orc_grid_upload(self.h, _p(s), _p(u))

    def positions(self):
        p = np.empty((self.n, 3), np.float32)
        self.lib.orc_grid_positions(self.h, _p(p))
        return p

    def extract_voxel(self, inside_empty=False):
        m = Mesh()
        self.lib.orc_extract_voxel(self.h, int(inside_empty), C.byref(m))
        nv, nf = m.n_vertices, m.n_faces
        out = {"vertices": np.ctypeslib.as_array(m.vertices, shape=(max(nv, 1) * 3,))[: nv * 3].reshape(nv, 3).copy(),
               "faces": np.ctypeslib.as_array(m.faces, shape=(max(nf, 1) * 3,))[: nf * 3].reshape(nf, 3).copy()}
        self.lib.orc_mesh_free(C.byref(m))
        return out

    def marching_cubes(self, iso=0.0, linear_interp=True):
        m = Mesh()
        ms = self.lib.orc_marching_cubes(self.h, iso, int(linear_interp), C.byref(m))
        out = mesh_to_numpy(m)
        self.lib.orc_mesh_free(C.byref(m))
        out["ms"] = ms
        return out


def marching_cubes_slab(grid, z0, z1, iso=0.0, linear_interp=True):
    m = Mesh()
    grid.lib.orc_marching_cubes_slab(grid.h, iso, int(linear_interp), z0, z1, C.byref(m))
    out = mesh_to_numpy(m)
    out["n_foreign"] = int(m.n_foreign_vertices)
    grid.lib.orc_mesh_free(C.byref(m))
    return out


def mesh_to_numpy(m):
    nv, nf = m.n_vertices, m.n_faces
    v = np.ctypeslib.as_array(m.vertices, shape=(max(nv, 1) * 3,))[: nv * 3].reshape(nv, 3).copy()
    f = np.ctypeslib.as_array(m.faces, shape=(max(nf, 1) * 3,))[: nf * 3].reshape(nf, 3).copy()
    k = np.ctypeslib.as_array(m.edge_keys, shape=(max(nv, 1) * 2,))[: nv * 2].reshape(nv, 2).copy()
    return {"vertices": v, "faces": f, "keys": k}


def make_sdf(mask, roi_min=None, roi_max=None, normalize=True, use_truncation=False, band=0.1):
    lib = load()
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    rmin = (C.c_int32 * 2)(*(roi_min or (0, 0)))
    rmax = (C.c_int32 * 2)(*(roi_max or (w - 1, h - 1)))
    out = np.empty((h, w), np.float32)
    lib.orc_make_sdf(_p(mask), w, h, rmin, rmax, int(normalize), int(use_truncation), band, _p(out))
    return out


def distance_transform_l1(mask, roi_min=None, roi_max=None):
    lib = load()
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    rmin = (C.c_int32 * 2)(*(roi_min or (0, 0)))
    rmax = (C.c_int32 * 2)(*(roi_max or (w - 1, h - 1)))
    out = np.empty((h, w), np.float32)
    lib.orc_distance_transform_l1(_p(mask), w, h, rmin, rmax, _p(out))
    return out


def pose_from_tum(t, q):
    lib = load()
    t = np.asarray(t, np.float64)
    q = np.asarray(q, np.float64)
    out = np.empty(12, np.float64)
    lib.orc_pose_from_tum(_p(t), _p(q), _p(out))
    return out.reshape(3, 4)


def affine_inverse(c2w):
    lib = load()
    a = np.ascontiguousarray(c2w, np.float64).reshape(12)
    out = np.empty(12, np.float64)
    lib.orc_affine_inverse(_p(a), _p(out))
    return out.reshape(3, 4)


def lookat_c2w(position, target, up):
    lib = load()
    a, b, c = (np.asarray(v, np.float64) for v in (position, target, up))
    out = np.empty(12, np.float64)
    lib.orc_lookat_c2w(_p(a), _p(b), _p(c), _p(out))
    return out.reshape(3, 4)


def focal_from_fov_y(height, fov_y_deg):
    return float(load().orc_focal_from_fov_y(height, fov_y_deg))
