"""z-slab sharding of the voxel grid across the GPUs of one node (one process per GPU).

Carving needs no communication: voxels are independent (reference voxel_carver.cc:442-491
touches only its own voxel) and z is the slowest index, so rank r owns the contiguous slab
z in [r*nz/G, (r+1)*nz/G).  Marching cubes needs the two slices below each slab: ONE
all-gather of every rank's last two slices (RCCL when the backend is nccl), after which
every rank extracts its own cells.  The per-rank meshes are stitched on the host by edge key.
"""
import ctypes as C

import numpy as np


def slab_range(nz, rank, world):
    """Contiguous z-range of `rank`; the first nz % world ranks get one extra slice."""
    base, rem = divmod(nz, world)
    z0 = rank * base + min(rank, rem)
    z1 = z0 + base + (1 if rank < rem else 0)
    return z0, z1


def exchange_halo(carver, rank, world):
    """Installs rank-1's last two slices as this rank's halo.  world == 1: nothing to do."""
    lib = carver._lib
    if world == 1:
        return
    import torch
    import torch.distributed as dist

    nbytes = int(lib.vcy_halo_bytes(carver.ctx))
    on_gpu = dist.get_backend() == "nccl"
    if on_gpu:
        send = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        recv = torch.empty(nbytes * world, dtype=torch.uint8, device="cuda")
        rc = lib.vcy_halo_pack(carver.ctx, C.c_void_p(send.data_ptr()))
        assert rc == 0, lib.vcy_last_error()
        carver.sync()
        dist.all_gather_into_tensor(recv, send)   # the single RCCL collective of the path
        torch.cuda.synchronize()
        rc = lib.vcy_halo_unpack(carver.ctx, C.c_void_p(recv.data_ptr()), rank, world)
        assert rc == 0, lib.vcy_last_error()
        carver.sync()
    else:
        # gloo (CPU tests / same-device debugging): stage the same bytes through the host
        send = torch.from_numpy(carver.halo_pack_host())
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send)
        carver.halo_unpack_host(np.concatenate([p.numpy() for p in parts]), rank, world)


def merge_meshes(meshes):
    """Stitches per-rank meshes (rank order) into the mesh a single-GPU extraction returns.

    Rank r's first n_foreign vertices duplicate vertices owned by rank r-1 (edges on the shared
    plane); they are dropped and the faces that use them are re-pointed by edge key.  Own
    vertices keep their order, so the result is vertex-for-vertex the serial scan's numbering.
    """
    verts, keys, faces = [], [], []
    offset = 0
    prev_key_to_gid = {}
    for m in meshes:
        nf_ = int(m["n_foreign"])
        v, k, f = m["vertices"], m["keys"], m["faces"]
        nown = len(v) - nf_
        remap = np.empty(len(v), np.int64)
        for i in range(nf_):
            remap[i] = prev_key_to_gid[(int(k[i, 0]), int(k[i, 1]))]
        remap[nf_:] = offset + np.arange(nown)
        verts.append(v[nf_:])
        keys.append(k[nf_:])
        faces.append(remap[f] if len(f) else f.astype(np.int64))
        # only vertices on this slab's top plane can be referenced by the next rank
        prev_key_to_gid = {(int(a), int(b)): offset + i for i, (a, b) in enumerate(k[nf_:])} \
            if len(meshes) > 1 else {}
        offset += nown
    return {
        "vertices": np.concatenate(verts) if verts else np.zeros((0, 3), np.float32),
        "keys": np.concatenate(keys) if keys else np.zeros((0, 2), np.int64),
        "faces": (np.concatenate(faces) if faces else np.zeros((0, 3), np.int64)).astype(np.int32),
    }
