"""Python host-side mirror of vacancy::VoxelCarver over the C-ABI (include/vacancy_hip.h).

Method names and argument meaning follow the reference class
(include/vacancy/voxel_carver.h:95-118 in unclearness/vacancy); errors that the reference
reports as `return false` + LOGE surface here as `False` returns with the message in
`last_error()`.  Every call goes to libvacancy_hip.so; there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import CarverOption, Mesh, UpdateOption, View, make_view  # noqa: F401


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _mesh_array(ptr, count, width, dtype):
    """Copy of `count` rows of `width` from a library-owned mesh array (NULL when the mesh is empty)."""
    if count == 0 or not ptr:
        return np.zeros((0, width), dtype)
    return np.ctypeslib.as_array(ptr, shape=(count * width,)).reshape(count, width).copy()


def last_error():
    return capi.load().vcy_last_error().decode()


class VoxelCarver:
    def __init__(self, option=None, device_id=0, z_range=None):
        self._lib = capi.load()
        self._ctx = C.c_void_p()
        self._device = device_id
        self._z_range = z_range
        self.option = option
        self.dims = None

    # -- VoxelCarver::set_option / Init (voxel_carver.cc:373-392)
    def set_option(self, option):
        self.option = option

    def Init(self):
        self.close()
        z0, z1 = self._z_range if self._z_range else (0, -1)
        rc = self._lib.vcy_create(C.byref(self.option), self._device, z0, z1, C.byref(self._ctx))
        if rc != 0:
            self._ctx = C.c_void_p()
            return False
        d = (C.c_int32 * 3)()
        self._lib.vcy_grid_dims(self._ctx, d)
        self.dims = tuple(d)
        zr = (C.c_int32 * 2)()
        self._lib.vcy_slab_range(self._ctx, zr)
        self.z_range = tuple(zr)
        return True

    def close(self):
        if self._ctx:
            self._lib.vcy_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ctx(self):
        return self._ctx

    @property
    def slab_voxels(self):
        return self.dims[0] * self.dims[1] * (self.z_range[1] - self.z_range[0])

    # -- Carve(camera, roi_min, roi_max, sdf)  (voxel_carver.cc:415-496)
    def Carve(self, view, sdf):
        if not self._ctx:
            return False
        sdf = np.ascontiguousarray(sdf, dtype=np.float32)
        assert sdf.shape == (view.height, view.width)
        return self._lib.vcy_carve(self._ctx, C.byref(view), _p(sdf)) == 0

    # -- Carve(camera, silhouette, roi_min, roi_max, &sdf)  (voxel_carver.cc:394-413)
    def CarveSilhouette(self, view, silhouette, return_sdf=False):
        if not self._ctx:
            return False
        mask = np.ascontiguousarray(silhouette, dtype=np.uint8)
        sdf = np.empty(mask.shape, np.float32) if return_sdf else None
        rc = self._lib.vcy_carve_silhouette(self._ctx, C.byref(view), _p(mask),
                                            _p(sdf) if return_sdf else None)
        return (rc == 0, sdf) if return_sdf else rc == 0

    # -- Carve(vector<Camera>, vector<Image1b>) (voxel_carver.cc:516-528), streamed + fused
    def CarveBatchSilhouettes(self, views, silhouettes):
        n = len(views)
        arr = (View * n)(*views)
        masks = [np.ascontiguousarray(m, np.uint8) for m in silhouettes]
        ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
        return self._lib.vcy_carve_batch_silhouettes(self._ctx, n, arr, ptrs) == 0

    def make_sdf_batch_into(self, views, silhouettes, out_ptrs):
        """vcy_make_sdf_batch_device: SDF images of `silhouettes` (host) into caller-owned device images."""
        n = len(views)
        arr = (View * n)(*views)
        masks = [np.ascontiguousarray(m, np.uint8) for m in silhouettes]
        mptr = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
        optr = (C.c_void_p * n)(*[p.value if isinstance(p, C.c_void_p) else int(p) for p in out_ptrs])
        return self._lib.vcy_make_sdf_batch_device(self._ctx, n, arr, mptr, optr) == 0

    def last_stream_ms(self):
        """(producer ms, carve ms, wall ms) of the last CarveBatchSilhouettes (vcy_last_stream_ms)."""
        a, b, w = C.c_float(), C.c_float(), C.c_float()
        assert self._lib.vcy_last_stream_ms(self._ctx, C.byref(a), C.byref(b), C.byref(w)) == 0, last_error()
        return a.value, b.value, w.value

    def download_voxels(self, ids):
        ids = np.ascontiguousarray(ids, np.int64)
        s = np.empty(len(ids), np.float32)
        u = np.empty(len(ids), np.int32)
        rc = self._lib.vcy_download_voxels(self._ctx, len(ids), _p(ids), _p(s), _p(u))
        if rc != 0:
            raise RuntimeError(last_error())
        return s, u

    # -- device-resident SDF images (bench / streaming)
    def upload_sdf(self, sdf):
        sdf = np.ascontiguousarray(sdf, dtype=np.float32)
        out = C.c_void_p()
        rc = self._lib.vcy_sdf_upload(self._ctx, _p(sdf), sdf.shape[1], sdf.shape[0], C.byref(out))
        if rc != 0:
            raise RuntimeError(last_error())
        return out

    def make_sdf_device(self, mask, roi_min=None, roi_max=None, normalize=True, use_truncation=False, band=0.1):
        """MakeSignedDistanceField on the device; returns a device pointer (free_device it)."""
        mask = np.ascontiguousarray(mask, np.uint8)
        h, w = mask.shape
        rmin = (C.c_int32 * 2)(*(roi_min or (0, 0)))
        rmax = (C.c_int32 * 2)(*(roi_max or (w - 1, h - 1)))
        out = C.c_void_p()
        rc = self._lib.vcy_make_sdf_device(self._ctx, _p(mask), w, h, rmin, rmax, int(normalize),
                                           int(use_truncation), band, C.byref(out))
        if rc != 0:
            raise RuntimeError(last_error())
        return out

    def download_image(self, ptr, shape):
        out = np.empty(shape, np.float32)
        assert self._lib.vcy_memcpy_d2h(self._ctx, _p(out), ptr, out.nbytes) == 0, last_error()
        return out

    def memcpy_h2d(self, ptr, array):
        array = np.ascontiguousarray(array)
        assert self._lib.vcy_memcpy_h2d(self._ctx, ptr, _p(array), array.nbytes) == 0, last_error()

    def free_device(self, ptr):
        self._lib.vcy_device_free(self._ctx, ptr)

    def CarveDevice(self, view, sdf_dev):
        return self._lib.vcy_carve_device(self._ctx, C.byref(view), sdf_dev) == 0

    # -- Carve(vector<Camera>, vector<...>) loop (voxel_carver.cc:516-528), fused on device
    @staticmethod
    def prepare_batch(views, sdf_devs):
        """ctypes arrays for CarveBatchDevice, built once when the same batch is carved repeatedly."""
        n = len(views)
        arr = (View * n)(*views)
        ptrs = (C.c_void_p * n)(*[p.value if isinstance(p, C.c_void_p) else p for p in sdf_devs])
        return n, arr, ptrs

    def CarveBatchDevice(self, views, sdf_devs=None):
        n, arr, ptrs = views if sdf_devs is None else self.prepare_batch(views, sdf_devs)
        return self._lib.vcy_carve_batch_device(self._ctx, n, arr, ptrs) == 0

    # -- ExtractIsoSurface(mesh, iso_level, linear_interp)  (voxel_carver.cc:540-543)
    def ExtractIsoSurface(self, iso_level=0.0, linear_interp=True):
        m = Mesh()
        rc = self._lib.vcy_extract_iso(self._ctx, iso_level, int(linear_interp), C.byref(m))
        if rc != 0:
            self._lib.vcy_mesh_free(C.byref(m))
            raise RuntimeError(last_error())
        nv, nf = m.n_vertices, m.n_faces
        out = {
            "vertices": _mesh_array(m.vertices, nv, 3, np.float32),
            "faces": _mesh_array(m.faces, nf, 3, np.int32),
            "keys": _mesh_array(m.edge_keys, nv, 2, np.int64),
            "n_foreign": int(m.n_foreign_vertices),
        }
        self._lib.vcy_mesh_free(C.byref(m))
        ms = C.c_float()
        self._lib.vcy_last_extract_ms(self._ctx, C.byref(ms))
        out["device_ms"] = ms.value
        self._lib.vcy_last_extract_wall_ms(self._ctx, C.byref(ms))
        out["wall_ms"] = ms.value  # vcy_extract_iso entry -> mesh arrays in host memory
        return out

    # -- ExtractVoxel(mesh, inside_empty)  (voxel_carver.cc:530-538)
    def ExtractVoxel(self, inside_empty=False, arrays=True):
        """arrays=False: only the sizes (bench.py times the library call, not numpy's copy of an 800 MB mesh)."""
        m = Mesh()
        rc = self._lib.vcy_extract_voxel(self._ctx, int(inside_empty), C.byref(m))
        if rc != 0:
            self._lib.vcy_mesh_free(C.byref(m))
            raise RuntimeError(last_error())
        nv, nf = m.n_vertices, m.n_faces
        if not arrays:
            self._lib.vcy_mesh_free(C.byref(m))
            return {"n_vertices": int(nv), "n_faces": int(nf)}
        out = {
            "vertices": _mesh_array(m.vertices, nv, 3, np.float32),
            "faces": _mesh_array(m.faces, nf, 3, np.int32),
        }
        self._lib.vcy_mesh_free(C.byref(m))
        return out

    def ExtractVoxelInto(self, inside_empty=False):
        """vcy_extract_voxel_into: the voxel mesh written by the library into numpy arrays this call allocates once the
        sizes are known (what the C++ class API does with its Mesh's vectors)."""
        got = {"vertices": np.zeros((0, 3), np.float32), "faces": np.zeros((0, 3), np.int32)}

        def provide(user, nv, nf, pv, pf):
            got["vertices"] = np.empty((nv, 3), np.float32)
            got["faces"] = np.empty((nf, 3), np.int32)
            pv[0] = got["vertices"].ctypes.data_as(C.POINTER(C.c_float))
            pf[0] = got["faces"].ctypes.data_as(C.POINTER(C.c_int32))
            return 0

        cb = capi.MeshArraysFn(provide)
        rc = self._lib.vcy_extract_voxel_into(self._ctx, int(inside_empty), cb, None)
        if rc != 0:
            raise RuntimeError(last_error())
        return got

    def extract_voxel_ids(self, inside_empty=False):
        """vcy_extract_voxel_ids: GLOBAL ids of the voxels of this slab that ExtractVoxel keeps, in scan order."""
        p, n = C.POINTER(C.c_int64)(), C.c_int64(0)
        rc = self._lib.vcy_extract_voxel_ids(self._ctx, int(inside_empty), C.byref(p), C.byref(n))
        if rc != 0:
            raise RuntimeError(last_error())
        if n.value == 0:
            return np.zeros(0, np.int64)
        ids = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        self._lib.vcy_ids_free(p)
        return ids

    # -- state access
    def download(self):
        n = self.slab_voxels
        s = np.empty(n, np.float32)
        u = np.empty(n, np.int32)
        rc = self._lib.vcy_download(self._ctx, _p(s), _p(u))
        if rc != 0:
            raise RuntimeError(last_error())
        return s, u

    def upload(self, sdf, update_num):
        s = np.ascontiguousarray(sdf, np.float32)
        u = np.ascontiguousarray(update_num, np.int32)
        rc = self._lib.vcy_upload(self._ctx, _p(s), _p(u))
        if rc != 0:
            raise RuntimeError(last_error())

    def positions(self):
        p = np.empty((self.slab_voxels, 3), np.float32)
        rc = self._lib.vcy_download_positions(self._ctx, _p(p))
        if rc != 0:
            raise RuntimeError(last_error())
        return p

    # -- halo staging through the host (gloo path of vacancy_amd.dist)
    def halo_pack_host(self):
        import ctypes as C_
        nbytes = int(self._lib.vcy_halo_bytes(self._ctx))
        dev = C_.c_void_p()
        host = np.empty(nbytes, np.uint8)
        rc = self._lib.vcy_device_alloc(self._ctx, nbytes, C_.byref(dev))
        assert rc == 0, last_error()
        assert self._lib.vcy_halo_pack(self._ctx, dev) == 0, last_error()
        assert self._lib.vcy_memcpy_d2h(self._ctx, _p(host), dev, nbytes) == 0, last_error()
        self._lib.vcy_device_free(self._ctx, dev)
        return host

    def halo_unpack_host(self, gathered, rank, world):
        import ctypes as C_
        gathered = np.ascontiguousarray(gathered, np.uint8)
        dev = C_.c_void_p()
        assert self._lib.vcy_device_alloc(self._ctx, gathered.nbytes, C_.byref(dev)) == 0, last_error()
        assert self._lib.vcy_memcpy_h2d(self._ctx, dev, _p(gathered), gathered.nbytes) == 0, last_error()
        assert self._lib.vcy_halo_unpack(self._ctx, dev, rank, world) == 0, last_error()
        self._lib.vcy_device_free(self._ctx, dev)

    def halo_install_host(self, pack):
        import ctypes as C_
        pack = np.ascontiguousarray(pack, np.uint8)
        dev = C_.c_void_p()
        assert self._lib.vcy_device_alloc(self._ctx, pack.nbytes, C_.byref(dev)) == 0, last_error()
        assert self._lib.vcy_memcpy_h2d(self._ctx, dev, _p(pack), pack.nbytes) == 0, last_error()
        assert self._lib.vcy_halo_install(self._ctx, dev) == 0, last_error()
        self._lib.vcy_device_free(self._ctx, dev)

    def state_diff(self, other):
        """Number of voxels whose (sdf bits, update_num) differ from `other` (same slab, same device);
        compared on the device (vcy_state_equal), nothing is downloaded."""
        n = C.c_int64(-1)
        if self._lib.vcy_state_equal(self._ctx, other._ctx, C.byref(n)) != 0:
            raise RuntimeError(last_error())
        return int(n.value)

    def use_stream_of(self, other):
        """Launch on another context's stream (several slabs of one GPU in sequence)."""
        st = C.c_void_p()
        assert self._lib.vcy_get_stream(other._ctx, C.byref(st)) == 0, last_error()
        assert self._lib.vcy_set_stream(self._ctx, st) == 0, last_error()

    def set_param(self, name, value):
        assert self._lib.vcy_set_param(self._ctx, name.encode(), int(value)) == 0, last_error()

    def get_param(self, name):
        v = C.c_int()
        assert self._lib.vcy_get_param(self._ctx, name.encode(), C.byref(v)) == 0, last_error()
        return v.value

    def reset(self):
        """Back to the state right after Init(): sdf = lowest(), update_num = 0."""
        assert self._lib.vcy_reset(self._ctx) == 0, last_error()

    def sync(self):
        self._lib.vcy_sync(self._ctx)

    def selftest(self):
        """vcy_selftest: device-side identities behind the fast paths; True when all hold."""
        return self._lib.vcy_selftest(self._ctx) == 0

    def last_carve_ms(self):
        """(pre-pass ms, carve kernel ms) of the last fused launch; needs set_param("carvetimer", 1)."""
        a, b = C.c_float(), C.c_float()
        assert self._lib.vcy_last_carve_ms(self._ctx, C.byref(a), C.byref(b)) == 0, last_error()
        return a.value, b.value

    def last_carve_pairs(self):
        """(processed, total, per-layer array) of the last fused launch; needs set_param("paircount", 1)."""
        a, b, n = C.c_int64(), C.c_int64(), C.c_int()
        per = np.zeros(4096, np.int64)
        rc = self._lib.vcy_last_carve_pairs(self._ctx, C.byref(a), C.byref(b), _p(per), len(per), C.byref(n))
        if rc != 0:
            raise RuntimeError(last_error())
        return int(a.value), int(b.value), per[: n.value].copy()

    def plan_z_slabs(self, views, sdf_devs, n_slabs, stride=0, brick_cost=0.0):
        """vcy_plan_z_slabs: (z_bounds [n_slabs + 1], layer_cost [brick layers of the grid]) for a fused carve of these
        views; `views, sdf_devs` as for CarveBatchDevice (or a prepared batch as `views`)."""
        n, arr, ptrs = views if sdf_devs is None else self.prepare_batch(views, sdf_devs)
        bounds = np.zeros(n_slabs + 1, np.int32)
        cost = np.zeros(8192, np.float64)
        nl = C.c_int()
        rc = self._lib.vcy_plan_z_slabs(self._ctx, n, arr, ptrs, int(n_slabs), int(stride), float(brick_cost),
                                        _p(bounds), _p(cost), len(cost), C.byref(nl))
        if rc != 0:
            raise RuntimeError(last_error())
        return [int(z) for z in bounds], cost[: nl.value].copy()

    def carve_log(self, clear=True, max_records=8192):
        """[(begin_ms, prepass_ms, kernel_ms, first_chunk)] of every chunk of every fused launch since "carvetimer" was
        set / the log was cleared (vcy_carve_log); waits for those launches, nothing synchronised in between."""
        b = np.empty(max_records, np.float32)
        p = np.empty(max_records, np.float32)
        k = np.empty(max_records, np.float32)
        f = np.empty(max_records, np.int32)
        n = C.c_int(0)
        rc = self._lib.vcy_carve_log(self._ctx, max_records, _p(b), _p(p), _p(k), _p(f), C.byref(n), int(clear))
        assert rc == 0, last_error()
        return [(float(b[i]), float(p[i]), float(k[i]), int(f[i])) for i in range(n.value)]

    def timer_begin(self):
        self._lib.vcy_timer_begin(self._ctx)

    def timer_end(self):
        ms = C.c_float()
        self._lib.vcy_timer_end(self._ctx, C.byref(ms))
        return ms.value


def carve_batch_silhouettes_sharded(carvers, views, silhouettes):
    """vcy_carve_batch_silhouettes_sharded: the slabs of one grid held by THIS process carve `views` from silhouettes
    in host memory; the devices share the producer (device r builds the SDFs of views r, r + R, ... of every chunk, one
    RCCL all-gather per chunk hands every device all of them)."""
    lib = capi.load()
    n = len(views)
    arr = (View * n)(*views)
    masks = [np.ascontiguousarray(m, np.uint8) for m in silhouettes]
    ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
    ctxs = (C.c_void_p * len(carvers))(*[c.ctx for c in carvers])
    rc = lib.vcy_carve_batch_silhouettes_sharded(ctxs, len(carvers), n, arr, ptrs)
    if rc != 0:
        e = RuntimeError(last_error())
        e.rc = rc
        raise e
    return True


def voxel_cubes(option, ids):
    """vcy_voxel_cubes: the serial half of ExtractVoxel (extract_voxel.cc:290-311) for kept voxel ids in scan order --
    host arithmetic, no GPU."""
    lib = capi.load()
    ids = np.ascontiguousarray(ids, np.int64)
    m = Mesh()
    rc = lib.vcy_voxel_cubes(C.byref(option), len(ids), _p(ids), C.byref(m))
    if rc != 0:
        lib.vcy_mesh_free(C.byref(m))
        raise RuntimeError(last_error())
    out = {"vertices": _mesh_array(m.vertices, m.n_vertices, 3, np.float32),
           "faces": _mesh_array(m.faces, m.n_faces, 3, np.int32)}
    lib.vcy_mesh_free(C.byref(m))
    return out


def voxel_cubes_into(option, ids):
    """vcy_voxel_cubes_into: as voxel_cubes, into numpy arrays allocated by the callback."""
    lib = capi.load()
    ids = np.ascontiguousarray(ids, np.int64)
    got = {"vertices": np.zeros((0, 3), np.float32), "faces": np.zeros((0, 3), np.int32), "calls": 0}

    def provide(user, nv, nf, pv, pf):
        got["calls"] += 1
        got["vertices"] = np.empty((nv, 3), np.float32)
        got["faces"] = np.empty((nf, 3), np.int32)
        pv[0] = got["vertices"].ctypes.data_as(C.POINTER(C.c_float))
        pf[0] = got["faces"].ctypes.data_as(C.POINTER(C.c_int32))
        return 0

    cb = capi.MeshArraysFn(provide)
    rc = lib.vcy_voxel_cubes_into(C.byref(option), len(ids), _p(ids), cb, None)
    if rc != 0:
        raise RuntimeError(last_error())
    return got


def halo_allgather(carvers):
    """All z-slabs of one grid held by THIS process, in z order: one native RCCL all-gather
    (vcy_halo_allgather) installs every slab's two halo slices."""
    lib = capi.load()
    arr = (C.c_void_p * len(carvers))(*[c.ctx for c in carvers])
    rc = lib.vcy_halo_allgather(arr, len(carvers))
    if rc != 0:
        e = RuntimeError(last_error())
        e.rc = rc
        raise e
    return lib.vcy_last_collective().decode()


def halo_exchange(carvers):
    """halo_allgather, or -- only when the library reports that librccl cannot be loaded at all
    (VCY_ERR_UNSUPPORTED) -- the explicit alternative: peer-to-peer copies slab by slab
    (vcy_halo_copy_from).  Returns what was done as a dict for the benchmark record."""
    lib = capi.load()
    try:
        text = halo_allgather(carvers)
    except RuntimeError as e:
        if getattr(e, "rc", 0) != capi.VCY_ERR_UNSUPPORTED:
            raise
        nbytes = 0
        for below, c in zip([None] + list(carvers[:-1]), carvers):
            if lib.vcy_halo_copy_from(c.ctx, below.ctx if below is not None else None) != 0:
                raise RuntimeError(last_error())
            nbytes += int(lib.vcy_halo_bytes(c.ctx)) if below is not None else 0
        return {"backend": "peer copies (vcy_halo_copy_from): librccl could not be loaded", "op": "hipMemcpyPeerAsync",
                "ranks": len({c._device for c in carvers}), "bytes_total": nbytes, "slabs": len(carvers),
                "note": str(e)[:160]}
    info = dict(kv.split("=", 1) for kv in text.split() if "=" in kv)
    ranks, per = int(info.get("ranks", 0)), int(info.get("bytes_per_rank", 0))
    # an all-gather hands EVERY rank every pack (ranks * bytes_per_rank received per rank) although a slab only needs
    # the pack of the slab below it: what the north star asks for, and small next to the state (10 MiB per slab at 1024^2)
    return {"backend": "rccl (native, vcy_halo_allgather)", "op": info.get("op"), "ranks": ranks,
            "bytes_per_rank": per, "bytes_received_per_rank": per * ranks,
            "bytes_needed_per_slab": int(capi.load().vcy_halo_bytes(carvers[0].ctx)),
            "rccl_version": int(info.get("version", 0)), "slabs": len(carvers)}


class ClockProbe:
    """Shader clock of the device while other work runs on it (vcy_clock_probe_*): one wave on a stream of its own
    samples the clock counter against the 100 MHz reference until stop().  `with ClockProbe(dev) as p: ...; p.result`."""

    def __init__(self, device_id=0, max_samples=1 << 16):
        self._lib = capi.load()
        self._p = C.c_void_p()
        self.result = None
        if self._lib.vcy_clock_probe_start(int(device_id), int(max_samples), C.byref(self._p)) != 0:
            raise RuntimeError("vcy_clock_probe_start: " + last_error())

    def stop(self):
        if self._p:
            mean, settled, lo, hi, cov, n = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_int()
            rc = self._lib.vcy_clock_probe_stop(self._p, C.byref(mean), C.byref(settled), C.byref(lo), C.byref(hi), C.byref(n),
                                                C.byref(cov))
            self._p = C.c_void_p()
            if rc != 0:
                raise RuntimeError("vcy_clock_probe_stop: " + last_error())
            self.result = {"mean_hz": mean.value, "settled_hz": settled.value, "min_hz": lo.value, "max_hz": hi.value, "samples": n.value,
                           "covered_ms": cov.value}
        return self.result

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.stop()
        return False

    def __del__(self):  # a probe that is dropped must not keep its wave resident (it also ends by itself after 2 s)
        try:
            self.stop()
        except Exception:
            pass


def measure_bandwidth(device_id=0, nbytes=1 << 31, reps=3):
    """(read GB/s, device-to-device copy GB/s) measured on this GPU (vcy_measure_bandwidth)."""
    lib = capi.load()
    rd, cp = C.c_double(), C.c_double()
    rc = lib.vcy_measure_bandwidth(int(device_id), int(nbytes), int(reps), C.byref(rd), C.byref(cp))
    if rc != 0:
        raise RuntimeError("vcy_measure_bandwidth: " + last_error())
    return rd.value, cp.value


def make_sdf(mask, roi_min=None, roi_max=None, normalize=True, use_truncation=False, band=0.1):
    """MakeSignedDistanceField (voxel_carver.cc:169-237) through the C-ABI."""
    lib = capi.load()
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    rmin = (C.c_int32 * 2)(*(roi_min or (0, 0)))
    rmax = (C.c_int32 * 2)(*(roi_max or (w - 1, h - 1)))
    out = np.empty((h, w), np.float32)
    rc = lib.vcy_make_sdf(_p(mask), w, h, rmin, rmax, int(normalize), int(use_truncation), band, _p(out))
    if rc != 0:
        raise RuntimeError(last_error())
    return out


def distance_transform_l1(mask, roi_min=None, roi_max=None):
    lib = capi.load()
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    rmin = (C.c_int32 * 2)(*(roi_min or (0, 0)))
    rmax = (C.c_int32 * 2)(*(roi_max or (w - 1, h - 1)))
    out = np.empty((h, w), np.float32)
    rc = lib.vcy_distance_transform_l1(_p(mask), w, h, rmin, rmax, _p(out))
    if rc != 0:
        raise RuntimeError(last_error())
    return out
