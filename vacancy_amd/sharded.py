"""ShardedVoxelCarver: several GPUs of one node driven from ONE process (Python mirror of
vacancy::ShardedVoxelCarver, include/vacancy/sharded_voxel_carver.h -- the reference's host model
is one process, voxel_carver.cc:516-528).

The grid is cut into S = G * k z-slabs, slab s on device s % G (cyclic, as vacancy_amd.dist).  One host
thread per device drives its slabs through the C-ABI (ctypes releases the GIL during the calls), so the
devices carve and extract concurrently; carving needs no exchange.  Before an extraction every slab needs
the two slices below it: ONE RCCL all-gather of every slab's boundary slices, issued by the library itself
(vcy_halo_allgather: ncclCommInitAll over the devices, one ncclAllGather inside a group call).  No torch,
no rendezvous, no second process: this is what `bench.py --gpus N --launch inprocess` runs, and what
bench.py falls back to when torch.distributed.run or the nccl backend cannot start.

`factory(option, device_id, z_range)` builds a slab object with the interface of
vacancy_amd.carver.VoxelCarver; the CPU tests pass host stand-ins, which exchange halos through
halo_pack_host / halo_install_host instead of RCCL (tests/test_dist_cpu.py).
"""
import threading
import time

from . import dist as vdist


class ShardedVoxelCarver:
    def __init__(self, option, devices, slabs_per_device=1, factory=None, nz=None, z_bounds=None):
        if factory is None:
            from .carver import VoxelCarver as factory  # noqa: N813
        self.option = option
        self.devices = list(devices) if devices else [0]
        self.k = int(slabs_per_device)
        self._factory = factory
        self._nz = nz
        self.z_bounds = list(z_bounds) if z_bounds is not None else None  # cuts of equal predicted cost (plan)
        self.plan_info = None
        self.slabs = []       # every slab of the grid, in z order
        self.by_device = []   # [[slab objects of device g, by slab id]]
        self.z_ranges = []    # [(z0, z1)] in z order
        self.last_collective = None
        self.last_kernel_ms = None   # per device, of the last carve_batch: device ms per step
        self.last_stats = None       # per device: {"prepass_ms", "kernel_ms", "period_ms", "idle_ms"} per step
        self.dims = None

    # -- VoxelCarver::Init on every slab (voxel_carver.cc:373-392)
    def Init(self):
        self.close()
        g = len(self.devices)
        nz = self._nz
        if nz is None:
            import ctypes as C
            from . import capi
            d = (C.c_int32 * 3)()
            if capi.load().vcy_compute_dims(self.option.bb_min, self.option.bb_max, self.option.resolution, d) != 0:
                return False
            nz = d[2]
        count = g * self.k
        if nz < 2 * count:
            raise ValueError("%d z slices cannot be cut into %d slabs of at least 2" % (nz, count))
        bounds = self.z_bounds if self.z_bounds is not None else vdist.equal_bounds(nz, count)
        if len(bounds) != count + 1 or bounds[0] != 0 or bounds[-1] != nz or \
                any(b1 - b0 < 2 for b0, b1 in zip(bounds[:-1], bounds[1:])):
            raise ValueError("z_bounds %s do not cut %d slices into %d slabs of at least 2" % (bounds, nz, count))
        self.by_device = [[] for _ in range(g)]
        for s in range(count):
            z0, z1 = bounds[s], bounds[s + 1]
            c = self._factory(self.option, self.devices[s % g], (z0, z1))
            if not c.Init():
                self.close()
                return False
            # (every slab keeps its own stream: a device's second slab fills the tail of its first one's launch)
            if self.k > 1 and hasattr(c, "set_param"):
                # one host thread drives the k slabs of a device: it must not wait for one slab's live-workgroup count
                # ("livesync" 1) before it can enqueue the next slab's launch
                c.set_param("livesync", 0)
            self.slabs.append(c)
            self.by_device[s % g].append(c)
            self.z_ranges.append((z0, z1))
        self.dims = getattr(self.slabs[0], "dims", None)
        return True

    # -- where to cut: slabs of equal predicted carve cost for these views (vcy_plan_z_slabs), before Init()
    def plan(self, views, sdf_host_images, stride=0, brick_cost=0.0):
        bounds, _, info = vdist.plan_bounds(self.option, self.devices[0], views, sdf_host_images,
                                            len(self.devices) * self.k, stride, brick_cost)
        self.z_bounds, self.plan_info = bounds, info
        return bounds

    def close(self):
        for c in reversed(self.slabs):
            if hasattr(c, "close"):
                c.close()
        self.slabs, self.by_device, self.z_ranges = [], [], []

    def set_param(self, name, value):
        for c in self.slabs:
            c.set_param(name, value)

    def _per_device(self, fn):
        """fn(device index, slabs of that device) on one thread per device; re-raises the first failure."""
        out, err = [None] * len(self.by_device), []

        def run(i):
            try:
                out[i] = fn(i, self.by_device[i])
            except BaseException as e:  # noqa: BLE001 -- reported to the caller below
                err.append(e)

        threads = [threading.Thread(target=run, args=(i,)) for i in range(len(self.by_device))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if err:
            raise err[0]
        return out

    def reset(self):
        for c in self.slabs:
            c.reset()

    def sync(self):
        self._per_device(lambda i, cs: [c.sync() for c in cs])

    # -- Carve(vector<Camera>, ...) loop (voxel_carver.cc:516-528): `batches[g]` = prepare_batch(...) with the
    # images resident on device g.  The steps of a device are QUEUED back to back (nothing synchronises in between;
    # a device that idles between launches drops its clocks) and the device is waited for once at the end.  Returns
    # the wall time of the slowest device in ms; per device, the mean per step of what the library's event log
    # ("carvetimer") holds is left in last_stats: pre-pass, carve kernel, step period, and idle = period - both.
    def carve_batch(self, batches, steps=1, reset=True):
        barrier = threading.Barrier(len(self.by_device))
        walls = [0.0] * len(self.by_device)
        stats = [None] * len(self.by_device)

        def run(i, cs):
            # a failure on one device must not leave the others waiting at a barrier for ever: abort it, so that
            # every thread comes back and _per_device re-raises the real error (BrokenBarrierError is secondary)
            try:
                body(i, cs)
            except threading.BrokenBarrierError:
                pass
            except BaseException:
                barrier.abort()
                raise

        def body(i, cs):
            logged = all(hasattr(c, "carve_log") for c in cs)
            before = [c.get_param("carvetimer") if logged and hasattr(c, "get_param") else 0 for c in cs]
            for c in cs:
                if logged:
                    c.set_param("carvetimer", 1)  # (clears the log)
                c.sync()
            barrier.wait()
            t0 = time.perf_counter()
            for _ in range(steps):
                if reset:
                    for c in cs:
                        c.reset()
                if not all(c.CarveBatchDevice(batches[i]) for c in cs):
                    raise RuntimeError("carve failed on device %d" % self.devices[i])
            for c in cs:
                c.sync()
            walls[i] = (time.perf_counter() - t0) * 1e3
            barrier.wait()
            if logged:
                pre = ker = 0.0
                dropped = 0
                for c, was in zip(cs, before):
                    if hasattr(c, "get_param"):
                        dropped += c.get_param("carvelog_dropped")  # (the log holds 8192 chunks; later ones are counted)
                    log = c.carve_log()
                    pre += sum(r[1] for r in log)
                    ker += sum(r[2] for r in log)
                    c.set_param("carvetimer", was)  # (later launches of the caller do not keep recording events)
                n = float(max(1, steps))
                stats[i] = {"prepass_ms": pre / n, "kernel_ms": ker / n, "period_ms": walls[i] / n,
                            "idle_ms": max(0.0, (walls[i] - pre - ker) / n)}
                if dropped:  # (per-step means over an incomplete log would be silently too small)
                    stats[i] = {"period_ms": walls[i] / n, "log_truncated_chunks": dropped}

        self._per_device(run)
        self.last_stats = stats
        self.last_kernel_ms = [(st["kernel_ms"] + st["prepass_ms"]) if st and "kernel_ms" in st else w / max(1, steps)
                               for st, w in zip(stats, walls)]
        return max(walls)

    # -- Carve(vector<Camera>, vector<Image1b>) from silhouettes in HOST memory (voxel_carver.cc:516-528 around
    # :394-413): the devices share the producer -- device r of G uploads and transforms views r, r + G, ... of every
    # chunk of 32, one RCCL all-gather per chunk hands every device all the images, every slab carves from its device's
    # copy (vcy_carve_batch_silhouettes_sharded; round 4 had every slab build every SDF).  `sharded_producer=False` is
    # that older form: one vcy_carve_batch_silhouettes per slab, a host thread per device.
    def CarveBatchSilhouettes(self, views, silhouettes, sharded_producer=True):
        lib = getattr(self.slabs[0], "_lib", None)
        if sharded_producer and lib is not None and hasattr(lib, "vcy_carve_batch_silhouettes_sharded"):
            from . import capi
            from . import carver as _vc
            try:
                return _vc.carve_batch_silhouettes_sharded(self.slabs, views, silhouettes)
            except RuntimeError as e:
                if getattr(e, "rc", 0) != capi.VCY_ERR_UNSUPPORTED:  # (no librccl: every slab for itself)
                    raise
        ok = self._per_device(lambda i, cs: all(c.CarveBatchSilhouettes(views, silhouettes) for c in cs))
        return all(ok)

    def last_stream_ms(self):
        """[(producer ms, carve ms, wall ms)] per slab of the last CarveBatchSilhouettes."""
        return [c.last_stream_ms() for c in self.slabs]

    # -- the exchange step of MarchingCubes() (marching_cubes.cc:93-101 reads z - 1)
    def exchange_halo(self):
        if len(self.slabs) == 1:
            self.last_collective = {"backend": "none", "ranks": 1, "bytes_per_rank": 0,
                                    "note": "one slab: nothing to exchange"}
            return self.last_collective
        lib = getattr(self.slabs[0], "_lib", None)
        if lib is not None and hasattr(lib, "vcy_halo_allgather"):
            from . import carver as _vc
            self.last_collective = _vc.halo_exchange(self.slabs)  # (peer copies, and says so, without librccl)
        else:  # host stand-ins
            packs = [c.halo_pack_host() for c in self.slabs]
            for s, c in enumerate(self.slabs):
                if s > 0:
                    c.halo_install_host(packs[s - 1])
            self.last_collective = {"backend": "host", "op": "copy", "ranks": 1,
                                    "bytes_per_rank": sum(len(p) for p in packs), "slabs": len(self.slabs)}
        return self.last_collective

    # -- ExtractIsoSurface (voxel_carver.cc:540-543): every slab on its device, then the merge by edge key
    def extract_slabs(self, iso_level=0.0, linear_interp=True, repeat=1):
        """[mesh of slab s] in z order (the last of `repeat` extractions each)."""
        self.exchange_halo()
        meshes = [None] * len(self.slabs)
        index = {id(c): s for s, c in enumerate(self.slabs)}

        def run(i, cs):
            for c in cs:
                for _ in range(repeat):
                    m = c.ExtractIsoSurface(iso_level, linear_interp)
                meshes[index[id(c)]] = m

        self._per_device(run)
        return meshes

    def ExtractIsoSurface(self, iso_level=0.0, linear_interp=True):
        return vdist.merge_meshes(self.extract_slabs(iso_level, linear_interp))

    # -- ExtractVoxel (voxel_carver.cc:530-538, extract_voxel.cc:258-317): the keep predicate and the compaction run on
    # every slab's device (the on-surface test reads the slice below a slab from its halo), the kept ids are walked in z
    # order by ONE drifting cube on the host, exactly the reference's arithmetic (vcy_voxel_cubes)
    def ExtractVoxel(self, inside_empty=False):
        if inside_empty:
            self.exchange_halo()
        ids = [None] * len(self.slabs)
        index = {id(c): s for s, c in enumerate(self.slabs)}

        def run(i, cs):
            for c in cs:
                ids[index[id(c)]] = c.extract_voxel_ids(inside_empty)

        self._per_device(run)
        import numpy as np
        from . import carver as _vc
        return _vc.voxel_cubes(self.option, np.concatenate(ids) if ids else np.zeros(0, np.int64))
