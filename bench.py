#!/usr/bin/env python3
"""Benchmark of the voxel-carving hot path on MI355X.

A step = one pass of the hot path over one batch of synthetic input: reset the grid, then
carve V silhouette SDFs (already resident in HBM) into the N^3 grid.  The headline workload is
BASELINE.json configs[2]/[3]: 1024^3 voxels x 32 views at 1280x720, default update mode (kMax,
bilinear); with --gpus G the grid is sharded by z-slab across G ranks (one process per GPU,
no collective in the carve path, so total work is fixed: strong scaling).

`python bench.py --gpus N` works from a plain shell: without WORLD_SIZE in the environment and
N > 1 the script re-executes itself under `torch.distributed.run --nproc-per-node N` (one process
per GPU).  The driver's own `python -m torch.distributed.run ... bench.py --gpus N` form is the same
code path.  The only collective of the path -- the halo all-gather before marching cubes -- goes
through RCCL (backend "nccl"); if RCCL cannot initialise the run FAILS unless --allow-gloo is given,
and the JSON line records which backend and how many ranks the collective saw ("collective").

Prints ONE JSON line (rank 0).  value = whole-job Mvoxel*views/s.  The marching-cubes rate
(Mcells/s) is measured after the timed region and reported in the same line under "mc".
"""
import argparse
import datetime
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMD = 256 * 4          # 256 CUs x 4 SIMD-32
CLOCK_HZ = 2.4e9          # max shader clock (same guide); the effective clock under load is lower


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--views", type=int, default=32)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--mode", default="default", choices=["default", "tsdf"])
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="shortcut for BASELINE.json configs[i]: 1 = 512^3 x16 @640x480 TSDF, 2/3 = 1024^3 x32 "
                         "@1280x720 (the default; 3 is the same with --gpus N), 4 = 2048^3 x64 @1920x1080")
    ap.add_argument("--batch", type=int, default=1, help="1: fused multi-view carve; 0: one launch per view")
    ap.add_argument("--cull", type=int, default=1,
                    help="1: drop (brick, view) pairs that provably cannot change the brick (results identical)")
    ap.add_argument("--slabs-per-gpu", type=int, default=0,
                    help="z-slabs per GPU, dealt cyclically (0: 1 on one or two GPUs, 2 on more -- evens out "
                         "the data-dependent cost of view dropping)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mc", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra measurements of the same kernel without view dropping and in TSDF mode")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--allow-gloo", action="store_true",
                    help="let the halo exchange fall back to gloo (host staging) when RCCL cannot initialise; "
                         "without this flag that is a failure")
    ap.add_argument("--launch", default="auto", choices=["auto", "torchrun", "inprocess"],
                    help="how N > 1 GPUs are driven: torchrun = one process per GPU (torch.distributed.run, the halo "
                         "all-gather through torch.distributed's nccl backend); inprocess = ONE process, a host thread "
                         "per GPU over the C-ABI, the all-gather issued by the library (vcy_halo_allgather); auto = "
                         "torchrun, and inprocess if that cannot start or its nccl backend fails")
    ap.add_argument("--plumbing-check", action="store_true",
                    help="no GPU work: launch, rendezvous and the halo all-gather of this configuration with "
                         "rank-stamped host buffers (CPU test of the multi-rank launch path)")
    a = ap.parse_args(argv)
    if a.config == 1:
        a.grid, a.views, a.width, a.height, a.mode = 512, 16, 640, 480, "tsdf"
    elif a.config in (2, 3):
        a.grid, a.views, a.width, a.height, a.mode = 1024, 32, 1280, 720, "default"
    elif a.config == 4:
        a.grid, a.views, a.width, a.height, a.mode = 2048, 64, 1920, 1080, "default"
    return a


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell: one process per GPU under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["VCY_BENCH_SELF_LAUNCHED"] = "1"  # a failing nccl backend is then handled by this parent (main)
    return subprocess.call(cmd, env=env)


def default_slabs_per_gpu(n_gpus):
    """z-slabs per GPU, dealt cyclically.  Two halves of a grid cost the same, so 2 GPUs take one slab each; from 4
    GPUs on the slabs through the object cost 1.6x the outer ones (view dropping) and every GPU pairs an outer with a
    central slab.  Measured slab by slab on one GPU (profiles/r03/slab_emulation.txt): 2 GPUs 1.83x (k = 1) against
    1.71x (k = 2); 4 GPUs 3.02x / 3.23x; 8 GPUs 5.50x / 6.14x; 4 slabs per GPU are behind everywhere (every launch
    costs about 0.1 ms of window maxima, gaps and ramp)."""
    return 1 if n_gpus <= 2 else 2


def usable_cores():
    """Host cores this process may really use: the cgroup CPU quota, not nproc (oversubscribing a
    throttled container makes the OpenMP loop 10x slower than it is)."""
    usable = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            usable = max(1, min(usable, int(round(int(quota) / float(period)))))
    except Exception:
        pass
    return usable


def cpu_baseline(args, views, sdfs, budget_s):
    """Times the CPU oracle (faithful restatement of the reference's OpenMP loop, AoS 40-byte
    voxels) on a bounded sample of the same workload: a coarser grid over the same bounding box,
    cameras and SDF images, as many views as fit the time budget."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from vacancy_amd import synth
    from vacancy_amd.capi import UpdateOption

    lib = O.load()
    usable = usable_cores()
    if "VCY_CPU_THREADS" in os.environ:
        usable = int(os.environ["VCY_CPU_THREADS"])
    lib.orc_set_num_threads(usable)
    n_cpu = min(args.grid, 512)
    uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if args.mode == "tsdf" \
        else UpdateOption()
    # same scene, coarser voxels: bb = +-grid/2, resolution = grid / n_cpu
    opt = synth.sphere_option(args.grid, uo)
    opt.resolution = float(args.grid) / n_cpu
    g = O.OracleGrid(opt)
    t_total, n_done = 0.0, 0
    for i in range(len(views)):
        ms = g.carve(views[i], sdfs[i])
        t_total += ms / 1e3
        n_done += 1
        if t_total > budget_s:
            break
    # the same loop on ONE thread (SURVEY 8d asks for both), on a 256^3 version of the scene, one view
    single = None
    try:
        lib.orc_set_num_threads(1)
        opt1 = synth.sphere_option(args.grid, uo)
        opt1.resolution = float(args.grid) / min(args.grid, 256)
        g1 = O.OracleGrid(opt1)
        ms1 = g1.carve(views[0], sdfs[0])
        single = round(g1.n / (ms1 / 1e3) / 1e6, 2)
        del g1
    finally:
        lib.orc_set_num_threads(usable)
    mesh = g.marching_cubes(0.0, True)
    mc_s = mesh["ms"] / 1e3
    cells = (g.dims[0] - 1) * (g.dims[1] - 1) * (g.dims[2] - 1)
    threads = lib.orc_omp_max_threads()
    extrapolated = n_cpu != args.grid
    return {
        "value": round(g.n * n_done / t_total / 1e6, 2),
        "unit": "Mvoxel*views/s",
        "cores": int(threads),
        "kind": "port",
        "extrapolated": extrapolated,
        "single_thread_value": single,
        "sample": "%soracle (OpenMP over z, %d threads = usable host cores of %d visible) on a %d^3 grid over the "
                  "same scene (the rate per voxel*view is what is reported; the %d^3 AoS grid of the reference "
                  "needs 43 GB), %d of %d views at %dx%d; times the Carve main loop only (reference "
                  "voxel_carver.cc:435,492)"
                  % ("EXTRAPOLATED from a smaller grid: " if extrapolated else "", threads, os.cpu_count() or 1,
                     n_cpu, args.grid, n_done, len(views), args.width, args.height),
        "mc_mcells_per_s": round(cells / mc_s / 1e6, 2),
        "mc_sample": "oracle MarchingCubes (serial std::map, like the reference) on the carved %d^3 grid" % n_cpu,
    }


def library_build():
    """vcy_version() of the loaded library: "vacancy_amd <version> (gfx950) src:<hash of its sources>"."""
    from vacancy_amd import capi
    return capi.load().vcy_version().decode()


def load_counters(key, build, path=None):
    """Per-launch hardware counters of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/counters.json, written by profiles/tools/summarize_pmc.py).  Returns (entry, None), or
    (None, reason): counters are only meaningful for the build they were collected on, so an entry whose
    "build" stamp is not `build` -- the vcy_version() of the library loaded now -- is refused."""
    path = path or os.path.join(ROOT, "profiles", "counters.json")
    try:
        entry = json.load(open(path)).get(key)
    except Exception as e:
        return None, "no counters file (%s)" % type(e).__name__
    if entry is None:
        return None, "no counters collected for %s" % key
    if entry.get("build") != build:
        return None, "counters of another build (%s), this library is %s" % (entry.get("build"), build)
    return entry, None


def valu_issue_cycles(ctr):
    """SIMD issue cycles of the VALU work of one launch: wave-level VALU instructions (SQ_INSTS_VALU) x 2 cycles, the
    issue cost of a wave64 VALU instruction on a SIMD-32 (MI355X_MICROARCH.md).  A lower bound: quarter-rate
    instructions (v_rcp_f32: 1 in 39 here) and back-to-back half-rate ones cost more.  How close the kernel is to the
    issue floor of its own instruction stream is MEASURED instead (`issue_floor`, profiles/tools/ab_variants.sh with
    the floor* builds: the same kernel with its tile loads and stores compiled out)."""
    if not ctr or "SQ_INSTS_VALU" not in ctr:
        return None
    return 2.0 * ctr["SQ_INSTS_VALU"]


def plumbing_check(args, rank, world, dist, backend):
    """CPU-only check of the launch path: the halo all-gather of this configuration with host buffers."""
    import numpy as np
    import torch
    from vacancy_amd import dist as vdist
    n = args.grid
    k = args.slabs_per_gpu if args.slabs_per_gpu > 0 else default_slabs_per_gpu(world)
    nbytes = 2 * n * n * 6  # two xy slices of (f32 sdf, u16 update_num)
    slabs = vdist.slabs_of_rank(n, rank, world, k)
    send = torch.from_numpy(np.concatenate([np.full(nbytes, s % 251, np.uint8) for s, _, _ in slabs]))
    ok = True
    if world > 1:
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send)
        flat = np.concatenate([p.numpy() for p in parts])
        for s, _, _ in slabs:
            if s > 0:
                off = vdist._pack_offset(s - 1, world, k, nbytes)
                ok = ok and bool((flat[off:off + nbytes] == (s - 1) % 251).all())
        t = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item() == 1.0)
    if rank == 0:
        emit({"metric": "Mvoxel*views/s (Carve)", "value": None, "unit": "Mvoxel*views/s",
              "plumbing_check": True, "ok": ok, "n_gpus": world,
              "collective": {"backend": backend, "ranks": world, "bytes_per_rank": nbytes * k,
                             "op": "all_gather", "slabs_per_rank": k}})
    return 0 if ok else 1


def run_inprocess(args, why=None):
    """N GPUs from ONE process (vacancy_amd.sharded.ShardedVoxelCarver): a host thread per device over the
    C-ABI, no torch, no rendezvous; the halo exchange is the library's own RCCL all-gather.  Same workload,
    same timed region (barrier -- here a thread barrier after a device sync -- on both sides, the slowest
    device's time) and the same JSON line as the one-process-per-GPU form."""
    from vacancy_amd import carver as vc
    from vacancy_amd import synth
    from vacancy_amd.capi import UpdateOption
    from vacancy_amd.sharded import ShardedVoxelCarver

    quiet_stdout()
    n, nv, G = args.grid, args.views, args.gpus
    uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if args.mode == "tsdf" \
        else UpdateOption()
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, args.width, args.height)
    sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
    devices = list(range(G))
    if "VCY_BENCH_FORCE_DEVICE" in os.environ:  # several "GPUs" on one device (boxes with a single GPU)
        devices = [int(os.environ["VCY_BENCH_FORCE_DEVICE"])] * G
    k = args.slabs_per_gpu if args.slabs_per_gpu > 0 else default_slabs_per_gpu(G)
    sh = ShardedVoxelCarver(opt, devices, k)
    if not sh.Init():
        raise SystemExit("vcy_create failed: " + vc.last_error())
    sh.set_param("fused", args.batch)
    sh.set_param("cull", args.cull)
    # inputs resident in HBM of every device before the timed region
    imgs = [cs[0].upload_sdf(sdf0) for cs in sh.by_device]
    batches = [vc.VoxelCarver.prepare_batch(views, [imgs[g]] * nv) for g in range(G)]
    sh.carve_batch(batches, steps=args.warmup)
    wall_ms = sh.carve_batch(batches, steps=args.steps)
    kernel_ms = list(sh.last_kernel_ms)
    value = float(n) ** 3 * nv * args.steps / (wall_ms * 1e-3) / 1e6
    bpv = 4.0 if args.mode == "default" else 4.0 + (1 if uo.voxel_max_update_num <= 254 else 2)
    views_per_launch = min(nv, 64) if args.batch else 1
    launches = ((nv + views_per_launch - 1) // views_per_launch) * k
    slab_vox = float(n) ** 3 / (G * k)
    avg_launch_ms = max(kernel_ms) / launches
    achieved = slab_vox * views_per_launch * bpv / (avg_launch_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "carve_fused_kernel" if args.batch else "carve_view_kernel",
                "avg_launch_ms": round(avg_launch_ms, 4),
                "note": "per GPU, of the slowest device; algorithmic bytes as in the one-GPU line"}
    mc = None
    collective = None
    if not args.no_mc:
        try:
            sh.set_param("meshkeys", 1)
            sh.extract_slabs(0.0, True)  # first run allocates
            meshes = sh.extract_slabs(0.0, True)
            collective = dict(sh.last_collective)
            per_dev = [sum(m["device_ms"] for m, c in zip(meshes, sh.slabs) if c in cs) for cs in sh.by_device]
            mc_ms = max(per_dev)
            cells = float(n - 1) ** 3
            mc = {"mcells_per_s": round(cells / (mc_ms * 1e-3) / 1e6, 1), "device_ms": round(mc_ms, 3),
                  "device_ms_per_gpu": [round(x, 3) for x in per_dev],
                  "vertices": int(sum(len(m["vertices"]) - m["n_foreign"] for m in meshes)),
                  "faces": int(sum(len(m["faces"]) for m in meshes)),
                  "roofline_frac": round(cells * 4.0 / (mc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS / G, 4),
                  "mesh_arrays": "vertices, faces, edge keys (slab merge)"}
        except Exception as e:
            mc = {"error": "%s: %s" % (type(e).__name__, e)}
    if collective is None:
        collective = {"backend": "none", "ranks": G, "bytes_per_rank": 0}
    collective["launch"] = "in-process: one host thread per GPU over the C-ABI"
    out = {"metric": "Mvoxel*views/s (Carve)", "value": round(value, 1), "unit": "Mvoxel*views/s", "n_gpus": G,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall_ms / args.steps, 3),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%d^3 grid x %d views at %dx%d, %s mode, z-slab sharded over %d GPU(s)"
                                  % (n, nv, args.width, args.height, args.mode, G),
                      "grid": n, "views": nv, "image": [args.width, args.height], "mode": args.mode,
                      "fused_views_per_launch": views_per_launch, "view_dropping": bool(args.cull),
                      "slabs_per_gpu": k, "launch": "inprocess", "devices": devices},
           "roofline": roofline, "mc": mc, "collective": collective,
           "per_gpu": [{"device": devices[g], "kernel_ms_per_step": round(kernel_ms[g], 3),
                        "slabs_z": [list(sh.z_ranges[s]) for s in range(g, G * k, G)]} for g in range(G)]}
    if why:
        out["config"]["launch_note"] = why
    emit(out)
    for g, cs in enumerate(sh.by_device):
        cs[0].free_device(imgs[g])
    sh.close()
    return 0


_REAL_STDOUT = None


def quiet_stdout():
    """Everything native code prints to fd 1 (RCCL's version banner arrives there when its stdio buffer is flushed
    at exit, i.e. AFTER the JSON line) goes to stderr instead: stdout carries the ONE JSON line and nothing else."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def main():
    args = parse()
    if "WORLD_SIZE" in os.environ or args.gpus == 1 or args.launch == "inprocess":
        quiet_stdout()  # (the self-launching parent only relays its children's output)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if args.launch == "inprocess" and not args.plumbing_check:
            raise SystemExit(run_inprocess(args))
        rc = self_launch(args)
        if rc != 0 and args.launch == "auto" and not args.plumbing_check:
            sys.stderr.write("bench: torch.distributed.run ended with rc %d; running the %d GPUs from this process\n"
                             % (rc, args.gpus))
            raise SystemExit(run_inprocess(args, "torch.distributed.run failed (rc %d): in-process fallback" % rc))
        raise SystemExit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    torch = None
    backend = os.environ.get("VCY_BENCH_BACKEND", "nccl")  # "gloo": debugging N ranks on one GPU
    if args.plumbing_check:
        backend = "gloo"
    if backend != "nccl" and not (args.allow_gloo or args.plumbing_check):
        raise SystemExit("bench: backend %s needs --allow-gloo (the halo exchange is an RCCL all-gather)" % backend)
    if "VCY_BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["VCY_BENCH_FORCE_DEVICE"])
    backend_note = None
    if world > 1:
        import torch  # device plumbing + RCCL only
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                        timeout=datetime.timedelta(seconds=300))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)  # fail here, on every rank alike, rather than mid-benchmark
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError("all_reduce over RCCL returned %r for %d ranks" % (probe.item(), world))
            except Exception as e:
                if not args.allow_gloo and args.launch == "auto" and "VCY_BENCH_INPROCESS_FALLBACK" not in os.environ:
                    # torch's nccl backend cannot start: the same job from ONE process (rank 0 drives every GPU
                    # through the C-ABI, the library issues the all-gather itself); the other ranks step aside.
                    # Under `python bench.py --gpus N` the parent does this after rc 3 instead (self_launch).
                    sys.stderr.write("bench: RCCL (backend nccl) failed on rank %d: %s\n" % (rank, e))
                    if os.environ.get("VCY_BENCH_SELF_LAUNCHED") == "1":
                        raise SystemExit(3)
                    if rank != 0:
                        raise SystemExit(0)
                    raise SystemExit(run_inprocess(args, "torch.distributed nccl backend failed (%s): in-process "
                                                         "fallback on rank 0" % str(e)[:160]))
                if not args.allow_gloo:
                    sys.stderr.write("bench: RCCL (backend nccl) failed: %s\n" % e)
                    raise SystemExit(3)
                # explicitly allowed: the carve path needs no collective, so keep measuring it; the halo
                # exchange and the timing reductions go through gloo and the JSON line says so
                sys.stderr.write("bench: nccl backend failed (%s); --allow-gloo: falling back to gloo\n" % e)
                backend_note = "nccl failed: %s" % str(e)[:200]
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(backend)
    if args.plumbing_check:
        rc = plumbing_check(args, rank, world, dist, backend)
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit(rc)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from vacancy_amd import carver as vc
    from vacancy_amd import dist as vdist
    from vacancy_amd import synth
    from vacancy_amd.capi import UpdateOption

    n, nv = args.grid, args.views

    def update_option(mode):
        return UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" \
            else UpdateOption()

    uo = update_option(args.mode)
    opt = synth.sphere_option(n, uo)
    views, masks = synth.sphere_views(n, nv, args.width, args.height)
    sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
    sdfs = [sdf0] * nv  # every view sees the same centred disc; the cameras differ

    k_slabs = args.slabs_per_gpu if args.slabs_per_gpu > 0 else default_slabs_per_gpu(world)
    my_slabs = vdist.slabs_of_rank(n, rank, world, k_slabs)

    def make_carvers(option, cull, slabs):
        out = []
        for _, z0, z1 in slabs:
            c = vc.VoxelCarver(option, device_id=local_rank, z_range=(z0, z1))
            if not c.Init():
                raise SystemExit("vcy_create failed: " + vc.last_error())
            # (every slab on a stream of its own: the second slab's launch fills the tail of the first one's,
            # 0-5 % per step on a rank's two slabs, profiles/r03/two_slabs_streams.txt)
            c.set_param("fused", args.batch)
            c.set_param("cull", cull)
            c.set_param("carvetimer", 1)  # HIP events around the carve kernel itself (vcy_last_carve_ms)
            out.append(c)
        return out

    devs = make_carvers(opt, args.cull, my_slabs)
    dev = devs[0]
    d_sdf = [dev.upload_sdf(s) for s in sdfs]  # inputs resident in HBM before the timed region

    def barrier():
        for c in devs:
            c.sync()
        if dist is not None:
            if backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    def run_steps(carvers, batch, count, record=None):
        lead = carvers[0]
        for _ in range(count):
            for c in carvers:
                c.reset()
            if len(carvers) == 1:
                lead.timer_begin()
                ok = lead.CarveBatchDevice(batch)
                ms = lead.timer_end()
            else:  # several streams: the step is over when the last of them is
                for c in carvers:
                    c.sync()
                t_step = time.perf_counter()
                ok = all(c.CarveBatchDevice(batch) for c in carvers)
                for c in carvers:
                    c.sync()
                ms = (time.perf_counter() - t_step) * 1e3
            if not ok:
                raise SystemExit("carve failed: " + vc.last_error())
            if record is not None:
                record.append(ms)
                parts = [c.last_carve_ms() for c in carvers] if args.batch else [(0.0, ms / len(carvers))] * len(carvers)
                kernel_only.append(sum(p[1] for p in parts))
                prepass_only.append(sum(p[0] for p in parts))

    kernel_ms = []      # per step: everything between vcy_timer_begin / _end (pre-pass, window maxima, carve kernel)
    kernel_only, prepass_only = [], []  # per step: the carve kernel(s) alone / what runs before them
    batch = vc.VoxelCarver.prepare_batch(views, d_sdf)
    run_steps(devs, batch, args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(devs, batch, args.steps, kernel_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_vv = float(n) ** 3 * nv * args.steps
    value = total_vv / elapsed / 1e6
    # per rank: device, kernel time per step (HIP events on its stream) and the z-ranges of its slabs, so that an
    # imbalance between the ranks is visible in the line
    per_gpu = None
    if dist is not None:
        row = torch.zeros(world, 2 + 2 * len(my_slabs), dtype=torch.float64, device=red_dev)
        row[rank, 0] = float(local_rank)
        row[rank, 1] = sum(kernel_ms) / max(1, len(kernel_ms))
        for j, (_, z0, z1) in enumerate(my_slabs):
            row[rank, 2 + 2 * j], row[rank, 3 + 2 * j] = float(z0), float(z1)
        dist.all_reduce(row)
        rows = row.cpu().tolist()
        per_gpu = [{"rank": r, "device": int(rows[r][0]), "kernel_ms_per_step": round(rows[r][1], 3),
                    "slabs_z": [[int(rows[r][2 + 2 * j]), int(rows[r][3 + 2 * j])] for j in range(len(my_slabs))]}
                   for r in range(world)]

    # roofline of the dominant kernel (carve), this rank's slab: algorithmic bytes per launch /
    # launch duration from HIP events on the launch stream.
    def bytes_per_vv(mode, u):
        return 4.0 if mode == "default" else 4.0 + (1 if u.voxel_max_update_num <= 254 else 2)

    slab_vox = sum(c.slab_voxels for c in devs) / float(len(devs))  # per launch
    FUSED_MAX = 64  # views per fused launch (carve_fused.hip)
    views_per_launch = min(nv, FUSED_MAX) if args.batch else 1
    launches_per_step = ((nv + views_per_launch - 1) // views_per_launch) * len(devs)
    n_rec = len(kernel_ms)
    avg_launch_ms = sum(kernel_only[:n_rec]) / n_rec / launches_per_step  # the dominant kernel alone, HIP events
    if avg_launch_ms <= 0.0:  # (a library without the carve timer, e.g. an older kernel linked in for an A/B run)
        avg_launch_ms = sum(kernel_ms) / n_rec / launches_per_step
    avg_step_device_ms = sum(kernel_ms) / n_rec
    avg_prepass_ms = sum(prepass_only[:n_rec]) / n_rec
    alg_bytes = slab_vox * views_per_launch * bytes_per_vv(args.mode, uo)
    achieved = alg_bytes / (avg_launch_ms * 1e-3) / 1e9
    ckey = "%s_%d_%d_b%d_c%d" % (args.mode, n, nv, args.batch, args.cull)
    build = library_build()
    ctr, ctr_note = load_counters(ckey, build) if world == 1 else (None, "counters are collected on one GPU")
    traffic = ctr.get("hbm_bytes_per_launch") if ctr else None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "carve_fused_kernel" if args.batch else "carve_view_kernel",
                "avg_launch_ms": round(avg_launch_ms, 4),
                "avg_launch_ms_note": "carve_fused_kernel alone (HIP events on its stream around the kernel, "
                                      "vcy_last_carve_ms): what rocprofv3 --kernel-trace reports for it",
                "step_device_ms": round(avg_step_device_ms, 4),
                "prepass_ms_per_step": round(avg_prepass_ms, 4),
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "achieved/frac use SURVEY 8(d)'s ALGORITHMIC bytes (one state read per voxel*view, the "
                        "reference's per-view API), so frac > 1 is not a bandwidth claim: the fused kernel keeps the state "
                        "in registers across the views and drops views that provably change nothing.  Its real HBM traffic "
                        "is `traffic` (hbm_real_frac of the peak); what binds it is VALU issue: valu_issue_frac_flat2 (2 "
                        "cycles per wave instruction) and issue_floor (measured: this kernel / the same kernel without "
                        "its tile loads and stores, for the variants whose control flow does not depend on the data)"}
    if ctr is None:
        roofline["counters_note"] = ctr_note
    # the roofs that actually bind the kernel: real HBM traffic and VALU issue slots (counters of the
    # committed PMC passes for this workload, duration measured live above)
    if traffic:
        t_k = ctr.get("trace_avg_ns", avg_launch_ms * 1e6) * 1e-9  # the carve kernel alone, in the profiled run
        roofline["hbm_real_gbs"] = round(traffic / t_k / 1e9, 1)
        roofline["hbm_real_frac"] = round(traffic / t_k / 1e9 / HBM_PEAK_GBS, 4)
        roofline["traffic_note"] = "carve_fused_kernel alone; the footprint pre-pass moves another %s B per step" % (
            ctr.get("prepass_hbm_bytes_per_launch", "?"))
    vcyc = valu_issue_cycles(ctr)
    if vcyc:
        roofline["bound_actual"] = "valu"
        # against the shader clock the profiled launch really ran at (GRBM_GUI_ACTIVE counts the busy cycles of each
        # of the 8 XCDs over the launch; 2.4 GHz is the boost clock)
        clk = CLOCK_HZ
        if ctr.get("GRBM_GUI_ACTIVE") and ctr.get("trace_avg_ns"):
            clk = ctr["GRBM_GUI_ACTIVE"] / 8.0 / (ctr["trace_avg_ns"] * 1e-9)
            roofline["shader_clock_ghz_profiled"] = round(clk / 1e9, 3)
        # (the counters are of the carve kernel alone: its own duration in the profiled run, not the step's)
        t_kernel = ctr.get("trace_avg_ns", avg_launch_ms * 1e6) * 1e-9
        roofline["valu_issue_frac_flat2"] = round(vcyc / (N_SIMD * clk * t_kernel), 4)
        roofline["valu_wave_insts_per_launch"] = ctr["SQ_INSTS_VALU"]
        roofline["valu_insts_per_voxel_view"] = round(ctr["SQ_INSTS_VALU"] * 64.0 / (slab_vox * views_per_launch), 3)
        roofline["counters_source"] = ctr.get("source")
    floor, _ = load_counters("issue_floor", build) if world == 1 else (None, None)
    if floor:
        roofline["issue_floor"] = {k: floor[k] for k in floor if k not in ("build",)}
    # marching cubes (second half of the metric), outside the timed region
    mc = None
    collective = {"backend": "none", "ranks": world, "bytes_per_rank": 0, "note": "one slab: nothing to exchange"}
    if not args.no_mc:
        try:
            info = vdist.exchange_halo(devs, rank, world)
            if info:
                collective = info
            if backend_note:
                collective["note"] = backend_note
            mc_ms, mc_wall, nvert, nface = 0.0, 0.0, 0, 0
            mc_calls = []
            single = world == 1 and len(devs) == 1  # no slab merge: the mesh is vertices + faces, as the reference's
            for c in devs:
                c.set_param("meshkeys", 0 if single else 1)
                mesh = c.ExtractIsoSurface(0.0, True)  # first run: scratch and host buffers are allocated
                runs = []
                for _ in range(3):  # a sequence of extractions, as in the reference's carve-and-extract loop
                    mesh = c.ExtractIsoSurface(0.0, True)
                    mc_calls.append([round(mesh["device_ms"], 3), round(mesh["wall_ms"], 3)])
                    runs.append((mesh["wall_ms"], mesh["device_ms"]))
                runs.sort()
                mc_ms += runs[1][1]   # the MEDIAN of the three by wall time
                mc_wall += runs[1][0]
                nvert += len(mesh["vertices"]) - mesh["n_foreign"]
                nface += len(mesh["faces"])
            if dist is not None:
                t = torch.tensor([mc_ms, mc_wall, float(nvert), float(nface)], dtype=torch.float64, device=red_dev)
                tmax = t.clone()
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                mc_ms, mc_wall = float(tmax[0].item()), float(tmax[1].item())
                nvert, nface = int(t[2].item()), int(t[3].item())
            cells = float(n - 1) ** 2 * (n - 1)
            mc = {"mcells_per_s": round(cells / (mc_ms * 1e-3) / 1e6, 1), "device_ms": round(mc_ms, 3),
                  "wall_ms": round(mc_wall, 3),
                  "wall_note": "vcy_extract_iso entry -> mesh arrays in host memory (what the reference's "
                               "MarchingCubes timer brackets, marching_cubes.cc:65-66,226-227); device_ms = kernels only; "
                               "the median (by wall time) of three consecutive extractions after a first "
                               "that allocates (all listed in calls_device_wall_ms)",
                  "calls_device_wall_ms": mc_calls,
                  "mesh_arrays": "vertices, faces" if single else "vertices, faces, edge keys (slab merge)",
                  "vertices": int(nvert), "faces": int(nface),
                  "roofline_frac": round(cells * 4.0 / (mc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if single:
                # the same extraction reading every brick ("mcskip" 0: no use of the brick minima the carve kernel keeps)
                c = devs[0]
                c.set_param("mcskip", 0)
                dense = sorted(c.ExtractIsoSurface(0.0, True)["device_ms"] for _ in range(3))[1]
                c.set_param("mcskip", 1)
                mc["device_ms_every_brick_read"] = round(dense, 3)
                mc["roofline_frac_every_brick_read"] = round(cells * 4.0 / (dense * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                mc["note"] = ("roofline_frac counts the ALGORITHMIC 4 B per cell; with the brick minima (default) voxels of "
                              "bricks the carve left entirely outside the surface are not read, so the bytes really moved "
                              "are fewer (`traffic`); *_every_brick_read is the dense sweep")
            mctr = load_counters("mc_%d" % n, build)[0] if world == 1 else None
            if mctr:
                mc["traffic"] = mctr.get("hbm_bytes_per_call")
        except Exception as e:  # the carve metric above stands on its own
            mc = {"error": "%s: %s" % (type(e).__name__, e)}

    # measured streaming bandwidth of this box next to the vendor figure (after the timed region and after the extractions:
    # freeing the probe's 4 GiB slows the next few device-to-host copies down by 2 ms)
    try:
        rd, cp = vc.measure_bandwidth(local_rank, 1 << 31, 3)
        roofline["measured_read_gbs"] = round(rd, 1)
        roofline["measured_copy_gbs"] = round(cp, 1)
        roofline["frac_of_measured_read"] = round(achieved / rd, 4) if rd > 0 else None
    except Exception as e:
        roofline["measured_read_gbs"] = None
        roofline["measured_error"] = "%s: %s" % (type(e).__name__, e)

    # the same kernel without view dropping and in TSDF mode (weighted average + truncation), measured in
    # the same run so that the headline's dependence on the scene is visible in the driver's record
    variants = None
    if world == 1 and not args.no_variants and args.batch:
        variants = {}
        for name, mode, cull in (("cull0", args.mode, 0), ("tsdf", "tsdf", 1)):
            if mode == args.mode and cull == args.cull:
                continue
            try:
                u2 = update_option(mode)
                cs = make_carvers(synth.sphere_option(n, u2), cull, my_slabs)
                if mode == args.mode:
                    dsdf2, own = d_sdf, False
                else:
                    s2 = vc.make_sdf(masks[0], use_truncation=bool(u2.use_truncation), band=u2.truncation_band)
                    p = cs[0].upload_sdf(s2)
                    dsdf2, own = [p] * nv, True
                b2 = vc.VoxelCarver.prepare_batch(views, dsdf2)
                ms2 = []
                run_steps(cs, b2, 1)
                run_steps(cs, b2, 3, ms2)
                avg = sum(ms2) / len(ms2)
                bpv = bytes_per_vv(mode, u2)
                variants[name] = {"value": round(float(n) ** 3 * nv / (avg * 1e-3) / 1e6, 1),
                                  "unit": "Mvoxel*views/s", "ms_per_step": round(avg, 3), "mode": mode,
                                  "view_dropping": bool(cull),
                                  "algorithmic_frac": round(float(n) ** 3 * nv * bpv / (avg * 1e-3) / 1e9
                                                            / HBM_PEAK_GBS, 4)}
                if own:
                    cs[0].free_device(dsdf2[0])
                for c in reversed(cs):
                    c.close()
            except Exception as e:
                variants[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        roofline["value_cull0"] = variants.get("cull0", {}).get("value")
        roofline["value_tsdf"] = variants.get("tsdf", {}).get("value")
        # The reference's own call pattern (examples.cc:117-149): `for each view: Carve(one view); ExtractIsoSurface()`.
        # An extraction between two views means every view is a launch of its own (nothing to fuse across): the
        # single-view path, where a wave drops its view against the brick minimum the previous launch left
        # before it reads any state.  "defer" 0 so that the carve is timed by itself (HIP events around each call);
        # per_view_defer0 is the same loop without the extractions.
        for name, extract in (("per_view_interleaved", True), ("per_view_defer0", False)):
            try:
                cs = make_carvers(opt, args.cull, my_slabs)
                c0 = cs[0]
                c0.set_param("defer", 0)
                c0.set_param("meshkeys", 0)
                carve_ms, mc_dev, mc_wall, t_wall = [], [], [], None
                for rep in range(2):  # the second pass is the one reported (buffers allocated, sizes guessed)
                    c0.reset()
                    carve_ms, mc_dev, mc_wall = [], [], []
                    c0.sync()
                    tw = time.perf_counter()
                    for i in range(nv):
                        c0.timer_begin()
                        if not c0.CarveDevice(views[i], d_sdf[i]):
                            raise RuntimeError(vc.last_error())
                        carve_ms.append(c0.timer_end())
                        if extract:
                            mesh = c0.ExtractIsoSurface(0.0, True)
                            mc_dev.append(mesh["device_ms"])
                            mc_wall.append(mesh["wall_ms"])
                    c0.sync()
                    t_wall = (time.perf_counter() - tw) * 1e3
                tot = sum(carve_ms)
                rec = {"value": round(float(n) ** 3 * nv / (tot * 1e-3) / 1e6, 1), "unit": "Mvoxel*views/s",
                       "carve_ms_total": round(tot, 3), "carve_ms_first_view": round(carve_ms[0], 3),
                       "carve_ms_per_view_after_first": round((tot - carve_ms[0]) / max(1, nv - 1), 3),
                       "algorithmic_frac": round(float(n) ** 3 * nv * bytes_per_vv(args.mode, uo) / (tot * 1e-3) / 1e9
                                                 / HBM_PEAK_GBS, 4),
                       "loop_wall_ms": round(t_wall, 2), "launches": nv, "defer": 0}
                if extract:
                    cells = float(n - 1) ** 3
                    med = sorted(mc_dev)[len(mc_dev) // 2]
                    rec["mc_device_ms_median"] = round(med, 3)
                    rec["mc_wall_ms_median"] = round(sorted(mc_wall)[len(mc_wall) // 2], 3)
                    rec["mc_mcells_per_s"] = round(cells / (med * 1e-3) / 1e6, 1)
                variants[name] = rec
                for c in reversed(cs):
                    c.close()
            except Exception as e:
                variants[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    out = {
        "metric": "Mvoxel*views/s (Carve)", "value": round(value, 1), "unit": "Mvoxel*views/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d^3 grid x %d views at %dx%d, %s mode, z-slab sharded over %d GPU(s)"
                               % (n, nv, args.width, args.height, args.mode, world),
                   "grid": n, "views": nv, "image": [args.width, args.height], "mode": args.mode,
                   "fused_views_per_launch": views_per_launch, "view_dropping": bool(args.cull),
                   "slabs_per_gpu": k_slabs,
                   "division": {2: "rcp + 3 (verified for this focal length)", 1: "rcp + 5 (verified)",
                                0: "full IEEE expansion"}.get(dev.get_param("div_level"), "?")},
        "roofline": roofline, "mc": mc, "collective": collective,
    }
    if per_gpu is not None:
        out["per_gpu"] = per_gpu
        out["config"]["launch"] = "torch.distributed.run (one process per GPU)"
        if isinstance(out["collective"], dict):
            out["collective"]["launch"] = "one process per GPU (torch.distributed, backend %s)" % backend
    if variants is not None:
        out["variants"] = variants
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, views, sdfs, args.cpu_seconds)
    if rank == 0:
        emit(out)
    for p in d_sdf:
        dev.free_device(p)
    for c in reversed(devs):
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
