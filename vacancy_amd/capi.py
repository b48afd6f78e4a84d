"""ctypes mirror of include/vacancy_hip.h (POD structs + the loader of libvacancy_hip.so).

The product path has no CPU fallback: `load()` raises if the HIP library is missing, and
`vcy_create` fails if no GPU is usable.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VCY_HIP_LIB: another build of the same library (kernel variants under development, profiles/tools/build_variant.sh)
LIB_PATH = os.environ.get("VCY_HIP_LIB") or os.path.join(_HERE, "csrc", "libvacancy_hip.so")

# vcy_status (include/vacancy_hip.h)
VCY_OK, VCY_ERR_INVALID_ARG, VCY_ERR_NOT_INITIALIZED, VCY_ERR_TOO_MANY_VOXELS = 0, -1, -2, -3
VCY_ERR_HIP, VCY_ERR_NO_DEVICE, VCY_ERR_UNSUPPORTED, VCY_ERR_INTERNAL = -4, -5, -6, -7
VCY_UPDATE_MAX, VCY_UPDATE_WEIGHTED_AVERAGE = 0, 1
VCY_INTERP_NN, VCY_INTERP_BILINEAR = 0, 1
VCY_OUTSIDE_NONE, VCY_OUTSIDE_MAX = 0, 1


class UpdateOption(C.Structure):
    """vacancy::VoxelUpdateOption (reference include/vacancy/voxel_carver.h:43-52)."""
    _fields_ = [
        ("voxel_update", C.c_int32),
        ("sdf_interp", C.c_int32),
        ("update_outside", C.c_int32),
        ("voxel_max_update_num", C.c_int32),
        ("voxel_update_weight", C.c_float),
        ("use_truncation", C.c_int32),
        ("truncation_band", C.c_float),
    ]

    def __init__(self, voxel_update=VCY_UPDATE_MAX, sdf_interp=VCY_INTERP_BILINEAR,
                 update_outside=VCY_OUTSIDE_NONE, voxel_max_update_num=255,
                 voxel_update_weight=1.0, use_truncation=False, truncation_band=0.1):
        super().__init__(voxel_update, sdf_interp, update_outside, voxel_max_update_num,
                         voxel_update_weight, int(bool(use_truncation)), truncation_band)


class CarverOption(C.Structure):
    """vacancy::VoxelCarverOption (voxel_carver.h:54-60)."""
    _fields_ = [
        ("bb_max", C.c_float * 3),
        ("bb_min", C.c_float * 3),
        ("resolution", C.c_float),
        ("sdf_minmax_normalize", C.c_int32),
        ("update_option", UpdateOption),
    ]

    def __init__(self, bb_min=(0, 0, 0), bb_max=(1, 1, 1), resolution=0.1,
                 sdf_minmax_normalize=True, update_option=None):
        super().__init__()
        self.bb_max = (C.c_float * 3)(*bb_max)
        self.bb_min = (C.c_float * 3)(*bb_min)
        self.resolution = resolution
        self.sdf_minmax_normalize = int(bool(sdf_minmax_normalize))
        self.update_option = update_option if update_option is not None else UpdateOption()


class View(C.Structure):
    """What Carve() reads from the camera + ROI (see vcy_view in vacancy_hip.h)."""
    _fields_ = [
        ("w2c", C.c_float * 12),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("is_ortho", C.c_int32),
        ("roi_min", C.c_int32 * 2),
        ("roi_max", C.c_int32 * 2),
        ("width", C.c_int32), ("height", C.c_int32),
    ]


class Mesh(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_int64),
        ("n_faces", C.c_int64),
        ("vertices", C.POINTER(C.c_float)),
        ("faces", C.POINTER(C.c_int32)),
        ("edge_keys", C.POINTER(C.c_int64)),
        ("n_foreign_vertices", C.c_int64),
    ]


# vcy_mesh_arrays_fn: int (*)(void* user, int64 n_vertices, int64 n_faces, float** vertices, int32** faces)
MeshArraysFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.POINTER(C.c_float)),
                           C.POINTER(C.POINTER(C.c_int32)))


def make_view(w2c_f32, fx, fy, cx, cy, width, height, roi_min=None, roi_max=None, is_ortho=False):
    v = View()
    flat = [float(x) for x in list(w2c_f32.reshape(-1))]
    assert len(flat) == 12
    v.w2c = (C.c_float * 12)(*flat)
    v.fx, v.fy, v.cx, v.cy = fx, fy, cx, cy
    v.is_ortho = int(bool(is_ortho))
    rmin = roi_min if roi_min is not None else (0, 0)
    rmax = roi_max if roi_max is not None else (width - 1, height - 1)
    v.roi_min = (C.c_int32 * 2)(*rmin)
    v.roi_max = (C.c_int32 * 2)(*rmax)
    v.width, v.height = width, height
    return v


_lib = None


def load():
    """Load libvacancy_hip.so (built by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "vacancy_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`."
            " There is no CPU fallback for the carving path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    P = C.POINTER
    vp = C.c_void_p
    sig = {
        "vcy_create": (C.c_int, [P(CarverOption), C.c_int, C.c_int, C.c_int, P(vp)]),
        "vcy_destroy": (None, [vp]),
        "vcy_grid_dims": (C.c_int, [vp, P(C.c_int32)]),
        "vcy_slab_range": (C.c_int, [vp, P(C.c_int32)]),
        "vcy_compute_dims": (C.c_int, [P(C.c_float), P(C.c_float), C.c_float, P(C.c_int32)]),
        "vcy_axis_positions": (C.c_int, [P(C.c_float), P(C.c_float), C.c_float, C.c_int, vp]),
        "vcy_carve": (C.c_int, [vp, P(View), vp]),
        "vcy_carve_device": (C.c_int, [vp, P(View), vp]),
        "vcy_carve_batch_device": (C.c_int, [vp, C.c_int, P(View), P(vp)]),
        "vcy_carve_silhouette": (C.c_int, [vp, P(View), vp, vp]),
        "vcy_carve_batch_silhouettes": (C.c_int, [vp, C.c_int, P(View), P(vp)]),
        "vcy_carve_batch_silhouettes_sharded": (C.c_int, [P(vp), C.c_int, C.c_int, P(View), P(vp)]),
        "vcy_make_sdf_batch_device": (C.c_int, [vp, C.c_int, P(View), P(vp), P(vp)]),
        "vcy_download_voxels": (C.c_int, [vp, C.c_int64, vp, vp, vp]),
        "vcy_distance_transform_l1": (C.c_int, [vp, C.c_int, C.c_int, P(C.c_int32), P(C.c_int32), vp]),
        "vcy_make_sdf": (C.c_int, [vp, C.c_int, C.c_int, P(C.c_int32), P(C.c_int32),
                                   C.c_int, C.c_int, C.c_float, vp]),
        "vcy_make_sdf_device": (C.c_int, [vp, vp, C.c_int, C.c_int, P(C.c_int32), P(C.c_int32),
                                          C.c_int, C.c_int, C.c_float, P(vp)]),
        "vcy_extract_iso": (C.c_int, [vp, C.c_double, C.c_int, P(Mesh)]),
        "vcy_extract_voxel": (C.c_int, [vp, C.c_int, P(Mesh)]),
        "vcy_extract_voxel_ids": (C.c_int, [vp, C.c_int, P(P(C.c_int64)), P(C.c_int64)]),
        "vcy_ids_free": (None, [P(C.c_int64)]),
        "vcy_voxel_cubes": (C.c_int, [P(CarverOption), C.c_int64, vp, P(Mesh)]),
        "vcy_extract_voxel_into": (C.c_int, [vp, C.c_int, MeshArraysFn, vp]),
        "vcy_voxel_cubes_into": (C.c_int, [P(CarverOption), C.c_int64, vp, MeshArraysFn, vp]),
        "vcy_mesh_free": (None, [P(Mesh)]),
        "vcy_last_extract_ms": (C.c_int, [vp, P(C.c_float)]),
        "vcy_last_extract_wall_ms": (C.c_int, [vp, P(C.c_float)]),
        "vcy_download": (C.c_int, [vp, vp, vp]),
        "vcy_upload": (C.c_int, [vp, vp, vp]),
        "vcy_download_positions": (C.c_int, [vp, vp]),
        "vcy_halo_bytes": (C.c_int64, [vp]),
        "vcy_halo_pack": (C.c_int, [vp, vp]),
        "vcy_halo_unpack": (C.c_int, [vp, vp, C.c_int, C.c_int]),
        "vcy_halo_install": (C.c_int, [vp, vp]),
        "vcy_halo_copy_from": (C.c_int, [vp, vp]),
        "vcy_halo_allgather": (C.c_int, [P(vp), C.c_int]),
        "vcy_halo_shutdown": (None, []),
        "vcy_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, P(vp)]),
        "vcy_comm_destroy": (None, [vp]),
        "vcy_halo_allgather_ranks": (C.c_int, [vp, P(vp), C.c_int]),
        "vcy_rendezvous_exchange": (C.c_int, [C.c_int, C.c_int, C.c_char_p, vp, C.c_int]),
        "vcy_last_collective": (C.c_char_p, []),
        "vcy_state_equal": (C.c_int, [vp, vp, P(C.c_int64)]),
        "vcy_device_count": (C.c_int, [P(C.c_int)]),
        "vcy_sdf_upload": (C.c_int, [vp, vp, C.c_int, C.c_int, P(vp)]),
        "vcy_device_free": (C.c_int, [vp, vp]),
        "vcy_device_alloc": (C.c_int, [vp, C.c_int64, P(vp)]),
        "vcy_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_int64]),
        "vcy_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_int64]),
        "vcy_reset": (C.c_int, [vp]),
        "vcy_set_param": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "vcy_get_param": (C.c_int, [vp, C.c_char_p, P(C.c_int)]),
        "vcy_set_stream": (C.c_int, [vp, vp]),
        "vcy_get_stream": (C.c_int, [vp, P(vp)]),
        "vcy_sync": (C.c_int, [vp]),
        "vcy_timer_begin": (C.c_int, [vp]),
        "vcy_timer_end": (C.c_int, [vp, P(C.c_float)]),
        "vcy_selftest": (C.c_int, [vp]),
        "vcy_last_carve_ms": (C.c_int, [vp, P(C.c_float), P(C.c_float)]),
        "vcy_last_carve_pairs": (C.c_int, [vp, P(C.c_int64), P(C.c_int64), vp, C.c_int, P(C.c_int)]),
        "vcy_plan_z_slabs": (C.c_int, [vp, C.c_int, P(View), P(vp), C.c_int, C.c_int, C.c_float, vp, vp, C.c_int,
                                       P(C.c_int)]),
        "vcy_last_stream_ms": (C.c_int, [vp, P(C.c_float), P(C.c_float), P(C.c_float)]),
        "vcy_partition_layers": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
        "vcy_carve_log": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, P(C.c_int), C.c_int]),
        "vcy_measure_bandwidth": (C.c_int, [C.c_int, C.c_uint64, C.c_int, P(C.c_double), P(C.c_double)]),
        "vcy_clock_probe_start": (C.c_int, [C.c_int, C.c_int, P(vp)]),
        "vcy_clock_probe_stop": (C.c_int, [vp, P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_int),
                                           P(C.c_double)]),
        "vcy_last_error": (C.c_char_p, []),
        "vcy_version": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the C-ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    lib._vcy_symbols = sorted(sig)
    _lib = lib
    return lib
