#!/usr/bin/env python3
"""Benchmark of the voxel-carving hot path on MI355X.

A step = one pass of the hot path over one batch of synthetic input: reset the grid, then
carve V silhouette SDFs (already resident in HBM) into the N^3 grid.  The headline workload is
BASELINE.json configs[2]/[3]: 1024^3 voxels x 32 views at 1280x720, default update mode (kMax,
bilinear); with --gpus G the grid is sharded by z-slab across G ranks (one process per GPU,
no collective in the carve path, so total work is fixed: strong scaling).  The cuts between the slabs are
placed where the library's planner predicts equal carve COST for these views (vcy_plan_z_slabs;
`--partition equal` for slabs of equal thickness).

Timing.  W warm-up steps, then K timed steps between two barriers (device sync + process barrier).  The
steps of the timed region are QUEUED back to back -- nothing synchronises between two steps; per-step device
times come from the library's event log afterwards -- and the warm-up is extended until the device has been
busy for --settle-ms (default 60 ms): an MI355X that idles for a millisecond drops its shader clock and needs
~40 ms of continuous load to come back (profiles/r04/clock_ramp.txt: the same 1.1 ms slab launch takes 1.29 ms
cold), so five warm-up steps of an 8-GPU rank (5 ms) would leave that rank timed on the ramp while the one-GPU
run (40 ms of warm-up) is not.  The JSON line says how many warm-up steps ran ("clock_settle").

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
import math
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
    ap.add_argument("--prologue", type=int, default=0, choices=[0, 1, 2],
                    help="where a fused launch gets its footprints (vcy_set_param \"prologue\"): 0 or 2 = pre-pass records "
                         "(chunks of at most 2 GiB); 1 = in the carve kernel's prologue (measured slower: 92.8 against 81.8 ms "
                         "at 2048^3 x 64)")
    ap.add_argument("--slabs-per-gpu", type=int, default=0,
                    help="z-slabs per GPU, dealt cyclically (0 = 1: one slab per GPU, cut by predicted cost)")
    ap.add_argument("--partition", default="planned", choices=["planned", "equal"],
                    help="where the z-slabs are cut: planned = equal predicted carve cost for these views "
                         "(vcy_plan_z_slabs), equal = equal thickness")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="the warm-up is extended (same step, untimed) until the device has been busy this long: "
                         "the shader clock needs ~40 ms of continuous load to settle (0: exactly --warmup steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mc", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra measurements of the same kernel without view dropping and in TSDF mode")
    ap.add_argument("--variants", default="all",
                    help="which of the extra measurements to run, comma separated: modes (cull0, tsdf), scenes (hard_scene, "
                         "two_batches), per_view (the reference's one-view-per-call loops), streamed (silhouettes from host "
                         "memory); default all")
    ap.add_argument("--no-configs", action="store_true",
                    help="one GPU, headline workload: do not append the `configs` block (BASELINE.json's other single-GPU "
                         "configurations -- configs[0] bunny sequence, configs[1] 512^3 x 16 TSDF, the configs[4] shape 2048^3 x 64 "
                         "-- measured after the timed region; also skipped with --no-variants)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--allow-gloo", action="store_true",
                    help="let the halo exchange fall back to gloo (host staging) when RCCL cannot initialise; "
                         "without this flag that is a failure")
    ap.add_argument("--launch", default="auto", choices=["auto", "torchrun", "inprocess"],
                    help="how N > 1 GPUs are driven: torchrun = one process per GPU (torch.distributed.run, the halo "
                         "all-gather through torch.distributed's nccl backend); inprocess = ONE process, a host thread "
                         "per GPU over the C-ABI, the all-gather issued by the library (vcy_halo_allgather); auto = "
                         "torchrun, and inprocess if that cannot start or its nccl backend fails")
    ap.add_argument("--verify-mesh", action="store_true",
                    help="N > 1: merge the slabs' meshes by edge key and compare them, array for array, with the mesh a "
                         "single context holding the whole grid extracts after the same views (on rank 0's GPU; outside "
                         "the timed region)")
    ap.add_argument("--no-preflight", action="store_true",
                    help="N > 1: skip the small-grid run of the whole sharded path (carve, halo all-gather, extraction, merge "
                         "against a single context) that precedes the timed region by default")
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
    """z-slabs per GPU.  One: the cuts are placed where the planner predicts equal cost (vcy_plan_z_slabs), which
    balances the ranks without the second launch per step that round 3's pairing of an outer with a central slab of
    equal thickness cost (profiles/r04/slab_emulation.txt: 8 GPUs 7.3x with planned cuts, 6.4x with equal ones)."""
    return 1


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
    # The same loop at the FULL grid size when the box can hold the reference's 40-byte AoS grid (1024^3: 43 GB) well
    # inside both its free memory and its cgroup limit -- one or two views, not extrapolated.  (Never attempted without
    # that headroom: a box driven out of memory is lost.)
    full = None
    try:
        need = 40.0 * float(args.grid) ** 3
        avail = 0.0
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = float(ln.split()[1]) * 1024.0
        limit = float("inf")
        for pth in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            try:
                t = open(pth).read().strip()
                if t != "max":
                    limit = min(limit, float(t))
            except OSError:
                pass
        used = 0.0
        try:
            used = float(open("/sys/fs/cgroup/memory.current").read().strip())
        except (OSError, ValueError):
            pass
        if n_cpu != args.grid and os.environ.get("VCY_CPU_FULL_SIZE", "1") != "0" and avail > 2.0 * need and limit - used > 2.0 * need:
            t_alloc = time.perf_counter()
            gf = O.OracleGrid(synth.sphere_option(args.grid, uo))
            t_alloc = time.perf_counter() - t_alloc
            tf, nf_ = 0.0, 0
            for i in range(len(views)):  # as many views as fit the budget (each 0.7 s on 16 cores at 1024^3)
                tf += gf.carve(views[i], sdfs[i]) / 1e3
                nf_ += 1
                if tf > budget_s:
                    break
            full = {"value": round(gf.n * nf_ / tf / 1e6, 2), "views": nf_, "grid": args.grid, "seconds": round(tf, 2),
                    "grid_bytes": need, "init_seconds": round(t_alloc, 2)}
            del gf
        else:
            full = {"skipped": "needs %.0f GB twice over: MemAvailable %.0f GB, cgroup headroom %s GB"
                               % (need / 1e9, avail / 1e9, "unlimited" if limit == float("inf") else "%.0f" % ((limit - used) / 1e9))}
    except Exception as e:  # the sample above stands on its own
        full = {"error": "%s: %s" % (type(e).__name__, e)}
    mesh = g.marching_cubes(0.0, True)
    mc_s = mesh["ms"] / 1e3
    cells = (g.dims[0] - 1) * (g.dims[1] - 1) * (g.dims[2] - 1)
    threads = lib.orc_omp_max_threads()
    extrapolated = n_cpu != args.grid
    if isinstance(full, dict) and "value" in full:
        # the full-size run is the baseline; the smaller grid's rate stays next to it
        return {
            "value": full["value"], "unit": "Mvoxel*views/s", "cores": int(threads), "kind": "port", "extrapolated": False,
            "single_thread_value": single,
            "sample": "oracle (OpenMP over z, %d threads = usable host cores of %d visible) on the FULL %d^3 grid (the "
                      "reference's 40-byte AoS voxels: %.1f GB, built in %.1f s), %d of %d views at %dx%d in %.1f s; times "
                      "the Carve main loop only (reference voxel_carver.cc:435,492)"
                      % (threads, os.cpu_count() or 1, args.grid, full["grid_bytes"] / 1e9, full["init_seconds"], full["views"],
                         len(views), args.width, args.height, full["seconds"]),
            "value_on_512_grid": round(g.n * n_done / t_total / 1e6, 2),
            "mc_mcells_per_s": round(cells / mc_s / 1e6, 2),
            "mc_sample": "oracle MarchingCubes (serial std::map, like the reference) on the carved %d^3 grid" % n_cpu,
        }
    return {
        "value": round(g.n * n_done / t_total / 1e6, 2),
        "unit": "Mvoxel*views/s",
        "cores": int(threads),
        "kind": "port",
        "extrapolated": extrapolated,
        "single_thread_value": single,
        "full_size": full,
        "sample": "%soracle (OpenMP over z, %d threads = usable host cores of %d visible) on a %d^3 grid over the "
                  "same scene (the rate per voxel*view is what is reported; the %d^3 AoS grid of the reference "
                  "needs %.1f GB), %d of %d views at %dx%d; times the Carve main loop only (reference "
                  "voxel_carver.cc:435,492)"
                  % ("EXTRAPOLATED from a smaller grid: " if extrapolated else "", threads, os.cpu_count() or 1,
                     n_cpu, args.grid, 40.0 * float(args.grid) ** 3 / 1e9, n_done, len(views), args.width, args.height),
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


def whole_grid_mesh(opt, device, batch_fn, fused, cull):
    """The mesh (with edge keys) of ONE context holding the whole grid after the same views: what the merged slab
    meshes must equal (--verify-mesh)."""
    from vacancy_amd import carver as vc
    w = vc.VoxelCarver(opt, device_id=device)
    if not w.Init():
        raise RuntimeError("vcy_create failed (whole grid for --verify-mesh): " + vc.last_error())
    try:
        w.set_param("fused", fused)
        w.set_param("cull", cull)
        w.set_param("meshkeys", 1)
        if not batch_fn(w):
            raise RuntimeError(vc.last_error())
        return w.ExtractIsoSurface(0.0, True)
    finally:
        w.close()


def compare_meshes(merged, ref):
    import numpy as np
    same = (merged["vertices"].shape == ref["vertices"].shape and merged["faces"].shape == ref["faces"].shape
            and np.array_equal(merged["vertices"].view(np.uint32), ref["vertices"].view(np.uint32))
            and np.array_equal(merged["faces"], ref["faces"]) and np.array_equal(merged["keys"], ref["keys"]))
    return {"merged_equals_single_context": bool(same), "vertices": int(len(merged["vertices"])),
            "faces": int(len(merged["faces"])), "single_context_vertices": int(len(ref["vertices"])),
            "single_context_faces": int(len(ref["faces"])),
            "note": "slab meshes merged by edge key (vacancy_amd.dist.merge_meshes) against one context holding the whole "
                    "grid on rank 0's device: vertex bits, faces and edge keys, array for array"}


def bytes_per_voxel_view(mode, nv, uo):
    """fp32 sdf, plus update_num in the weighted-average modes -- one byte while at most 255 views have
    been applied since the fill (the library widens the counters lazily, vcy_set_param "lazycount"), else two."""
    return 4.0 if mode == "default" else 4.0 + (1 if min(nv, uo.voxel_max_update_num + 1) <= 255 else 2)


def contract_roofline(roofline, *, mode, cull, n, nv, bpv, launches_per_step, pairs_frac, variants, mc, configs, build, world):
    """The `roofline` object of the JSON line as the contract reads it: `achieved`, `frac` and `avg_launch_ms` are SURVEY
    8(d)'s figure -- algorithmic bytes of EVERY voxel*view / duration of the carve kernel -- for a launch that really
    evaluates every voxel*view: the same kernel with view dropping off, measured in this run right after the timed region
    (`frac_source` says so).  The headline launch drops (brick, view) pairs that provably change nothing and keeps the
    state in registers across the views, so the same formula applied to IT is not a bandwidth fraction (it exceeds 1);
    that figure moves to `headline_kernel.per_view_api_equivalent`, next to the per-processed-pair fraction.  Everything a
    reader needs to recompute the fractions of this line sits inside this object (the driver's record keeps `config`,
    `roofline` and `cpu_baseline` only): marching cubes by device and by wall time, the single-view launches of the
    reference's own call pattern, the other single-GPU configurations."""
    vv_bytes = float(n) ** 3 * nv * bpv  # whole grid, all views (all ranks together)
    head = {"avg_launch_ms": roofline.get("avg_launch_ms"), "algorithmic_bytes_per_launch": roofline.get("algorithmic_bytes_per_launch"),
            "traffic": roofline.get("traffic"), "hbm_real_gbs": roofline.get("hbm_real_gbs"),
            "hbm_real_frac": roofline.get("hbm_real_frac"), "pairs_processed_frac": pairs_frac,
            "per_view_api_equivalent": {"gbs": roofline.get("achieved"), "frac": roofline.get("frac"),
                                        "note": "algorithmic bytes of every voxel*view / this launch's duration: what a per-view "
                                                "implementation (the reference's API, one state read per voxel*view) would have to "
                                                "stream to finish the same work in the same time -- NOT bytes this kernel moves"}}
    if isinstance(pairs_frac, (int, float)) and roofline.get("achieved"):
        head["frac_processed_pairs"] = round(pairs_frac * roofline["achieved"] / HBM_PEAK_GBS, 4)
        head["frac_processed_pairs_note"] = ("pairs_processed_frac x algorithmic bytes / this launch's duration / peak: the rate at "
                                             "which the (brick, view) pairs that are really evaluated go through the kernel")
    for k in ("valu_issue_frac_flat2", "valu_issue_frac_flat2_live", "valu_wave_insts_per_launch", "valu_insts_per_voxel_view",
              "shader_clock_ghz_profiled", "traffic_note"):
        if k in roofline:
            head[k] = roofline[k]
    src = None
    c0 = (variants or {}).get("cull0") if cull else None
    if cull and isinstance(c0, dict) and c0.get("kernel_ms"):
        ker = c0["kernel_ms"]
        roofline["achieved"] = round(vv_bytes / (ker * 1e-3) / 1e9, 1)
        roofline["avg_launch_ms"] = round(ker / launches_per_step, 4)
        roofline["step_device_ms_dropping_off"] = c0.get("ms_per_step")
        ctr0, why0 = load_counters("%s_%d_%d_b1_c0" % (mode, n, nv), build) if world == 1 else (None, "counters are collected on one GPU")
        roofline["traffic"] = ctr0.get("hbm_bytes_per_launch") if ctr0 else None
        if ctr0 is None:
            roofline["traffic_note"] = why0
        src = ("carve_fused_kernel with view dropping OFF (vcy_set_param cull 0): every voxel*view of the workload evaluated, "
               "same grid, views and images, measured in this run after the timed region (variants.cull0)")
    elif cull and "frac_processed_pairs" in head:
        roofline["achieved"] = round(head["frac_processed_pairs"] * HBM_PEAK_GBS, 1)
        src = "processed pairs only (no dropping-off run in this invocation): pairs_processed_frac x algorithmic bytes / duration"
    elif cull:
        # (neither a dropping-off run nor a pair count: nothing that is a fraction can be stated)
        roofline["achieved"] = None
        src = "unavailable: run without --no-variants (dropping-off launch) or on one GPU (pair count)"
    else:
        src = "the timed launch itself (view dropping is off: every voxel*view evaluated)"
    roofline["frac"] = round(roofline["achieved"] / HBM_PEAK_GBS, 4) if roofline.get("achieved") else None
    roofline["frac_source"] = src
    roofline["bound"] = "hbm"
    roofline["bound_note"] = ("the contract's roof (SURVEY 8d: algorithmic HBM bytes); what binds the kernel in practice is VALU "
                              "issue: frac_binding / issue_floor / headline_kernel.valu_issue_frac_flat2")
    roofline["headline_kernel"] = head
    for k in ("hbm_real_gbs", "hbm_real_frac", "valu_issue_frac_flat2", "valu_issue_frac_flat2_live", "valu_wave_insts_per_launch",
              "valu_insts_per_voxel_view", "algorithmic_bytes_per_launch", "note", "bound_contract", "bound_actual",
              "frac_of_measured_read", "frac_every_voxel_view"):
        roofline.pop(k, None)
    roofline["algorithmic_bytes_per_launch"] = vv_bytes / launches_per_step / max(1, world)
    roofline["algorithmic_bytes_per_voxel_view"] = bpv
    if roofline.get("measured_read_gbs") and roofline.get("achieved"):
        roofline["frac_of_measured_read"] = round(roofline["achieved"] / roofline["measured_read_gbs"], 4)
    # marching cubes, the second half of the metric: device (kernels) and wall (call -> mesh in host memory)
    if isinstance(mc, dict) and "device_ms" in mc:
        cells = float(n - 1) ** 3
        roofline["mc"] = {k: mc.get(k) for k in ("device_ms", "wall_ms", "mcells_per_s", "mcells_per_s_wall", "roofline_frac",
                                                 "algorithmic_frac_bricks_skipped", "algorithmic_frac_wall", "traffic",
                                                 "vertices", "faces", "device_ms_every_brick_read")}
        roofline["mc"]["cells"] = cells
        roofline["mc"]["algorithmic_bytes_per_cell"] = 4.0
    # the reference's call pattern (one view per launch) and the other modes, from the same run
    pv = {}
    for name in ("per_view_interleaved", "per_view_defer0", "per_view_tsdf"):
        r = (variants or {}).get(name)
        if isinstance(r, dict) and "value" in r:
            pv[name] = {k: r.get(k) for k in ("value", "carve_ms_first_view", "carve_ms_per_view_after_first", "algorithmic_frac",
                                              "mc_device_ms_median", "mc_wall_ms_median", "mode")}
    if pv:
        roofline["per_view_launches"] = pv
    modes = {}
    for name in ("cull0", "tsdf", "hard_scene", "two_batches", "new_views_every_step"):
        r = (variants or {}).get(name)
        if isinstance(r, dict) and "value" in r:
            modes[name] = {k: r.get(k) for k in ("value", "ms_per_step", "kernel_ms", "prepass_ms", "algorithmic_frac_kernel_alone",
                                                 "pairs_processed_frac", "mode")}
    if modes:
        roofline["same_kernel_other_workloads"] = modes
    st = (variants or {}).get("streamed_silhouettes")
    if isinstance(st, dict) and "wall_ms" in st:
        roofline["streamed_silhouettes"] = {k: st.get(k) for k in ("wall_ms", "producer_ms", "carve_ms", "overlap", "value_pcie_inclusive")}
    if isinstance(configs, dict):
        side = {}
        for key, r in configs.items():
            if not isinstance(r, dict) or "value" not in r:
                continue
            e = {"workload": r.get("workload"), "value": r.get("value"), "ms_per_step": r.get("ms_per_step")}
            rf = r.get("roofline")
            if isinstance(rf, dict):
                e.update({"frac": rf.get("frac"), "per_view_api_equivalent_frac": rf.get("per_view_api_equivalent_frac"),
                          "kernel_ms_per_step": rf.get("kernel_ms_per_step"),
                          "kernel_ms_per_step_dropping_off": rf.get("kernel_ms_per_step_dropping_off"),
                          "prepass_ms_per_step": rf.get("prepass_ms_per_step"),
                          "traffic": rf.get("traffic"), "hbm_real_frac": rf.get("hbm_real_frac"),
                          "valu_issue_frac_flat2": rf.get("valu_issue_frac_flat2")})
            for k in ("sequence_wall_ms", "carve_wall_ms", "extract_voxel_wall_ms", "mc_wall_ms"):
                if k in r:
                    e[k] = r[k]
            if isinstance(r.get("mc"), dict):
                e["mc"] = {k: r["mc"].get(k) for k in ("device_ms", "wall_ms", "mcells_per_s", "mcells_per_s_wall", "roofline_frac",
                                                      "algorithmic_frac_bricks_skipped", "algorithmic_frac_wall",
                                                      "device_ms_every_brick_read")}
            side[key] = e
        if side:
            roofline["other_configs"] = side
    return roofline


def side_config(device, build, label, n, nv, w, h, mode, steps, warmup=1, settle_ms=40.0, mc_runs=3):
    """One of BASELINE.json's OTHER single-GPU configurations, measured in the same process after the headline's timed
    region (so that the driver's record holds a number for each): the same step -- reset + fused carve of nv resident SDF
    images -- `steps` times queued back to back between two device syncs, then marching cubes.  A compact record: value,
    ms_per_step, roofline {frac (SURVEY 8d's algorithmic bytes / the carve kernel's own duration), traffic and bound
    from the committed counters of this shape or null}, mc {device_ms, wall_ms}."""
    from vacancy_amd import carver as vc
    from vacancy_amd import synth
    from vacancy_amd.capi import UpdateOption
    t_begin = time.perf_counter()
    uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
    views, masks = synth.sphere_views(n, nv, w, h)
    sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
    c = vc.VoxelCarver(synth.sphere_option(n, uo), device_id=device)
    if not c.Init():
        return {"label": label, "error": "vcy_create failed: " + vc.last_error()}
    imgs = []
    try:
        c.set_param("carvetimer", 1)
        c.set_param("meshkeys", 0)
        imgs = [c.upload_sdf(sdf0) for _ in range(nv)]  # an image per view, as in the headline step (each gets its window planes)
        batch = vc.VoxelCarver.prepare_batch(views, imgs)

        def run(count):
            c.sync()
            t = time.perf_counter()
            for _ in range(count):
                c.reset()
                if not c.CarveBatchDevice(batch):
                    raise RuntimeError("carve failed: " + vc.last_error())
            c.sync()
            return (time.perf_counter() - t) * 1e3

        busy = run(max(1, warmup))
        ran = max(1, warmup)
        if busy < settle_ms:  # clocks (see the module docstring); the device has just run the headline, so less is needed
            extra = int(math.ceil((settle_ms - busy) / (busy / ran)))
            busy += run(extra)
            ran += extra
        c.set_param("carvetimer", 1)  # clears the event log
        wall = run(steps)
        log = c.carve_log()
        pre = sum(r[1] for r in log) / steps
        ker = sum(r[2] for r in log) / steps
        launches = len(log) / float(steps)
        bpv = bytes_per_voxel_view(mode, nv, uo)
        vv = float(n) ** 3 * nv
        achieved = vv * bpv / (ker * 1e-3) / 1e9  # all launches of a step together: the same ratio as per launch
        ctr, why = load_counters("%s_%d_%d_b1_c1" % (mode, n, nv), build)
        # kMax drops (brick, view) pairs that provably change nothing: the algorithmic bytes of the pairs it never touches
        # are not a bandwidth claim.  `frac` is then the SAME kernel with dropping off ("cull" 0: every voxel*view
        # evaluated), one more step; the dropping launch's figure is kept as per_view_api_equivalent_frac.
        dropping = mode != "tsdf"
        equivalent = achieved
        ker_all = None
        if dropping:
            c.set_param("cull", 0)
            run(1)
            c.set_param("carvetimer", 1)
            run(1)
            ker_all = sum(r[2] for r in c.carve_log())
            c.set_param("cull", 1)
            c.set_param("carvetimer", 1)
            achieved = vv * bpv / (ker_all * 1e-3) / 1e9
        roof = {"bound": "valu" if ctr and "SQ_INSTS_VALU" in ctr else None, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": ctr.get("hbm_bytes_per_launch") if ctr else None,
                "kernel": "carve_fused_kernel", "kernel_ms_per_step": round(ker, 4),
                "prepass_ms_per_step": round(pre, 4), "kernel_launches_per_step": round(launches, 2),
                "algorithmic_bytes_per_voxel_view": bpv}
        if dropping:
            roof["kernel_ms_per_step_dropping_off"] = round(ker_all, 4)
            roof["per_view_api_equivalent_frac"] = round(equivalent / HBM_PEAK_GBS, 4)
            roof["frac_note"] = ("frac = the carve kernel with view dropping off (every voxel*view evaluated, "
                                 "kernel_ms_per_step_dropping_off); kernel_ms_per_step and value are the default launch, "
                                 "whose algorithmic figure (per_view_api_equivalent_frac) counts pairs it never touches")
        if ctr:
            t_k = ctr.get("trace_avg_ns", 0.0) * 1e-9
            if roof["traffic"] and t_k > 0:
                roof["hbm_real_frac"] = round(roof["traffic"] / t_k / 1e9 / HBM_PEAK_GBS, 4)
                roof["traffic_note"] = "HBM bytes of ONE launch of the carve kernel (a step is kernel_launches_per_step launches)"
            vcyc = valu_issue_cycles(ctr)
            clk = ctr.get("shader_clock_hz") or CLOCK_HZ
            if vcyc and t_k > 0:
                roof["valu_issue_frac_flat2"] = round(vcyc / (N_SIMD * clk * t_k), 4)
            roof["counters_source"] = ctr.get("source")
        else:
            roof["counters_note"] = why
            roof["bound_note"] = ("no counters of this build at this shape: `bound` is left null; the same kernel at 1024^3 is "
                                  "bound by VALU issue (the headline's roofline block), frac is the algorithmic figure")
        rec = {"label": label, "workload": "%d^3 grid x %d views at %dx%d, %s mode, 1 GPU" % (n, nv, w, h, mode),
               "value": round(vv * steps / (wall * 1e-3) / 1e6, 1), "unit": "Mvoxel*views/s", "steps": steps,
               "warmup_steps_run": ran, "ms_per_step": round(wall / steps, 3), "roofline": roof}
        try:
            c.ExtractIsoSurface(0.0, True)  # allocates
            runs = sorted((m["wall_ms"], m["device_ms"], len(m["vertices"]), len(m["faces"]))
                          for m in (c.ExtractIsoSurface(0.0, True) for _ in range(mc_runs)))
            wl, dv_default, nvert, nface = runs[len(runs) // 2]
            # device_ms = the kernels with the mesh left in HBM ("mcdirect" 0).  By default a mesh of up to 32 MiB is written
            # straight into host memory by the last kernel, whose duration is then the PCIe transfer: that is wall_ms.
            direct = c.get_param("mcdirect")
            c.set_param("mcdirect", 0)
            c.ExtractIsoSurface(0.0, True)
            dv = sorted(c.ExtractIsoSurface(0.0, True)["device_ms"] for _ in range(mc_runs))[mc_runs // 2]
            c.set_param("mcskip", 0)
            dense = sorted(c.ExtractIsoSurface(0.0, True)["device_ms"] for _ in range(3))[1]
            c.set_param("mcskip", 1)
            c.set_param("mcdirect", direct)
            cells = float(n - 1) ** 3
            rec["mc"] = {"device_ms": round(dv, 3), "wall_ms": round(wl, 3),
                         "device_ms_writing_to_host": round(dv_default, 3),
                         "mcells_per_s": round(cells / (dv * 1e-3) / 1e6, 1),
                         "mcells_per_s_wall": round(cells / (wl * 1e-3) / 1e6, 1),
                         # roofline_frac: the sweep that reads every brick ("mcskip" 0) -- 4 B per cell really streamed;
                         # the default skips bricks the carve left outside the surface, so its 4 B per cell are
                         # algorithmic only and that fraction may exceed 1
                         "device_ms_every_brick_read": round(dense, 3),
                         "roofline_frac": round(cells * 4.0 / (dense * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "algorithmic_frac_bricks_skipped": round(cells * 4.0 / (dv * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "algorithmic_frac_wall": round(cells * 4.0 / (wl * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "vertices": int(nvert), "faces": int(nface)}
        except Exception as e:
            rec["mc"] = {"error": "%s: %s" % (type(e).__name__, e)}
        rec["wall_s_spent"] = round(time.perf_counter() - t_begin, 2)
        return rec
    except Exception as e:
        return {"label": label, "error": "%s: %s" % (type(e).__name__, e)}
    finally:
        for img in imgs:
            c.free_device(img)
        c.close()


def bunny_sequence(device, resolution=2.5, reps=2):
    """BASELINE configs[0] on the GPU: the reference's examples.cc:117-149 sequence on the data/ bunny fixture
    (tests/golden/bunny: 6 masks 320x240 + tumpose.txt) -- per view Carve(camera, silhouette) (SDF built on the device),
    ExtractVoxel, MarchingCubes with and without interpolation -- at `resolution` (examples.cc runs 10.0; 2.5 gives a
    216 x 214 x 170 grid).  Wall clock of the calls from pageable host inputs to meshes in host memory."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bunny_data as B
    from vacancy_amd import carver as vc
    from vacancy_amd import synth
    t_begin = time.perf_counter()
    views = B.bunny_views(lambda t, q: synth.affine_inverse(synth.pose_from_tum(t, q)))
    masks = B.load_masks()
    c = vc.VoxelCarver(B.bunny_option(resolution), device_id=device)
    if not c.Init():
        return {"label": "configs[0]", "error": "vcy_create failed: " + vc.last_error()}
    try:
        c.set_param("meshkeys", 0)
        best = None
        for rep in range(reps + 1):  # the first pass allocates; the fastest of the others is reported
            c.reset()
            c.sync()
            carve = xv = mc = mc_dev = 0.0
            mc_lib = []
            t_seq = time.perf_counter()
            for view, mask in zip(views, masks):
                t0 = time.perf_counter()
                if not c.CarveSilhouette(view, mask):
                    raise RuntimeError(vc.last_error())
                c.sync()  # (queued views are applied here: the carve is timed by itself)
                t1 = time.perf_counter()
                vox = c.ExtractVoxel(False, arrays=False)  # (the library call; not numpy's copy of the mesh arrays)
                t2 = time.perf_counter()
                m1 = c.ExtractIsoSurface(0.0, True)
                m2 = c.ExtractIsoSurface(0.0, False)
                t3 = time.perf_counter()
                carve, xv, mc = carve + (t1 - t0), xv + (t2 - t1), mc + (t3 - t2)
                mc_dev += m1["device_ms"] + m2["device_ms"]
                mc_lib += [m1["wall_ms"], m2["wall_ms"]]
            seq = time.perf_counter() - t_seq
            rec = (seq, carve, xv, mc, mc_dev, len(m1["vertices"]), len(m1["faces"]), vox["n_vertices"], sorted(mc_lib))
            if rep > 0 and (best is None or rec[0] < best[0]):
                best = rec
        seq, carve, xv, mc, mc_dev, nvert, nface, nvox, mc_lib = best
        nvox_grid = c.dims[0] * c.dims[1] * c.dims[2]
        cells = float(c.dims[0] - 1) * (c.dims[1] - 1) * (c.dims[2] - 1)
        return {"label": "configs[0]",
                "workload": "data/ bunny, 6 masks 320x240, resolution %g (%d x %d x %d voxels), examples.cc sequence per view: "
                            "Carve(silhouette) + ExtractVoxel + 2 x MarchingCubes" % ((resolution,) + tuple(c.dims)),
                "value": round(nvox_grid * len(views) / carve / 1e6, 1), "unit": "Mvoxel*views/s",
                "value_note": "voxels x views / wall time of the 6 Carve(silhouette) calls (mask upload + SDF build + carve, "
                              "one launch per view, each followed by a sync)",
                "sequence_wall_ms": round(seq * 1e3, 3), "carve_wall_ms": round(carve * 1e3, 3),
                "extract_voxel_wall_ms": round(xv * 1e3, 3), "mc_wall_ms": round(mc * 1e3, 3),
                "mc": {"extractions": 2 * len(views), "wall_ms": round(mc_lib[len(mc_lib) // 2], 3),
                       "wall_ms_last_view": round(max(mc_lib), 3),
                       "wall_note": "vcy_extract_iso entry -> mesh arrays in host memory, median / largest of the 12 calls (the mesh "
                                    "grows with every view); python_call_ms adds the ctypes call and numpy's view of the arrays",
                       "python_call_ms": round(mc * 1e3 / (2 * len(views)), 3),
                       "device_ms": round(mc_dev / (2 * len(views)), 3),
                       "mcells_per_s_wall": round(cells / (mc_lib[len(mc_lib) // 2] * 1e-3) / 1e6, 1)},
                "final_mesh": {"vertices": int(nvert), "faces": int(nface), "voxel_mesh_vertices": int(nvox)},
                "roofline": None, "roofline_note": "a 7.7 M voxel grid: every call is launch latency, not bandwidth",
                "wall_s_spent": round(time.perf_counter() - t_begin, 2)}
    except Exception as e:
        return {"label": "configs[0]", "error": "%s: %s" % (type(e).__name__, e)}
    finally:
        c.close()


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
    if args.partition == "planned" and G * k > 1:
        sh.plan(views, [sdf0] * nv)  # cuts of equal predicted cost (vcy_plan_z_slabs on the first device)
    if not sh.Init():
        raise SystemExit("vcy_create failed: " + vc.last_error())
    sh.set_param("fused", args.batch)
    sh.set_param("cull", args.cull)
    # inputs resident in HBM of every device before the timed region
    imgs = [cs[0].upload_sdf(sdf0) for cs in sh.by_device]
    batches = [vc.VoxelCarver.prepare_batch(views, [imgs[g]] * nv) for g in range(G)]
    # before anything is timed: the whole sharded path once on a small grid, merged mesh against a single context
    # (the one-process-per-GPU form does the same; config.preflight_mesh_check)
    preflight = None
    if G * k > 1 and not args.no_preflight and args.batch:
        try:
            from vacancy_amd import dist as vdist_
            pn, pnv = 256, 8
            popt = synth.sphere_option(pn, uo)
            pviews, pmasks = synth.sphere_views(pn, pnv, 320, 240)
            psdf = vc.make_sdf(pmasks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
            psh = ShardedVoxelCarver(popt, devices, k)
            if not psh.Init():
                raise RuntimeError(vc.last_error())
            psh.set_param("cull", args.cull)
            psh.set_param("meshkeys", 1)
            pimgs = [cs[0].upload_sdf(psdf) for cs in psh.by_device]
            pb = [vc.VoxelCarver.prepare_batch(pviews, [pimgs[g]] * pnv) for g in range(G)]
            psh.carve_batch(pb, steps=1)
            pmeshes = psh.extract_slabs(0.0, True)
            pref = whole_grid_mesh(popt, devices[0], lambda w: w.CarveBatchDevice(pb[0]), args.batch, args.cull)
            preflight = {"mesh_check": compare_meshes(vdist_.merge_meshes(pmeshes), pref), "grid": pn, "views": pnv,
                         "collective": {k_: dict(psh.last_collective or {}).get(k_) for k_ in ("backend", "ranks", "version", "op", "bytes_per_rank")}}
            for g, cs in enumerate(psh.by_device):
                cs[0].free_device(pimgs[g])
            psh.close()
        except Exception as e:
            preflight = {"error": "%s: %s" % (type(e).__name__, e)}
        if not (isinstance(preflight.get("mesh_check"), dict) and preflight["mesh_check"].get("merged_equals_single_context")):
            sys.stderr.write("bench: PREFLIGHT MESH CHECK DID NOT PASS: %r\n" % (preflight,))
    warm_steps, warm_ms = 0, 0.0
    if args.warmup > 0:
        warm_ms = sh.carve_batch(batches, steps=args.warmup)
        warm_steps = args.warmup
        if args.settle_ms > 0 and warm_ms < args.settle_ms:  # until the clocks have settled (see the module docstring)
            extra = int(math.ceil((args.settle_ms - warm_ms) / (warm_ms / args.warmup)))
            warm_ms += sh.carve_batch(batches, steps=extra)
            warm_steps += extra
    wall_ms = sh.carve_batch(batches, steps=args.steps)
    stats = list(sh.last_stats)
    kernel_ms = [st["kernel_ms"] for st in stats]
    value = float(n) ** 3 * nv * args.steps / (wall_ms * 1e-3) / 1e6
    # (update_num is one byte until more than 255 views have been applied since the fill: vcy_set_param "lazycount")
    bpv = 4.0 if args.mode == "default" else 4.0 + (1 if min(nv, uo.voxel_max_update_num + 1) <= 255 else 2)
    views_per_launch = min(nv, 64) if args.batch else 1
    launches = ((nv + views_per_launch - 1) // views_per_launch) * k
    slowest = max(range(G), key=lambda g: kernel_ms[g])
    slab_vox = sum(c.slab_voxels for c in sh.by_device[slowest]) / float(k)
    avg_launch_ms = kernel_ms[slowest] / launches
    achieved = slab_vox * views_per_launch * bpv / (avg_launch_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "carve_fused_kernel" if args.batch else "carve_view_kernel",
                "avg_launch_ms": round(avg_launch_ms, 4),
                "note": "per GPU, of the device whose carve kernel takes longest; algorithmic bytes as in the one-GPU "
                        "line (4 B per voxel*view of that device's slab); what binds the kernel: see the one-GPU line"}
    if args.batch and args.cull:
        # the timed launches drop (brick, view) pairs that provably change nothing: `frac` counts the pairs really
        # evaluated (the one-GPU line measures the dropping-off launch itself; see contract_roofline)
        pairs = None
        try:
            sh.set_param("paircount", 1)
            sh.carve_batch(batches, steps=1)
            proc = tot = 0
            for c in sh.slabs:
                a, b, _ = c.last_carve_pairs()
                proc, tot = proc + a, tot + b
            sh.set_param("paircount", 0)
            pairs = round(proc / float(tot), 4) if tot else None
        except Exception as e:
            pairs = "%s: %s" % (type(e).__name__, e)
        roofline["headline_kernel"] = {"avg_launch_ms": roofline["avg_launch_ms"], "pairs_processed_frac": pairs,
                                       "per_view_api_equivalent": {"gbs": roofline["achieved"], "frac": roofline["frac"]}}
        if isinstance(pairs, float):
            roofline["achieved"] = round(pairs * achieved, 1)
            roofline["frac"] = round(pairs * achieved / HBM_PEAK_GBS, 4)
            roofline["frac_source"] = ("processed pairs: pairs_processed_frac x algorithmic bytes / launch duration (the launch "
                                       "evaluates that share of the voxel*views; the rest is dropped as provably unchanged)")
        else:
            roofline["achieved"] = roofline["frac"] = None
            roofline["frac_source"] = "unavailable (pair count failed)"
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
                  "algorithmic_frac_bricks_skipped": round(cells * 4.0 / (mc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS / G, 4),
                  "mesh_arrays": "vertices, faces, edge keys (slab merge)"}
            if args.verify_mesh:
                from vacancy_amd import dist as vdist_
                ref = whole_grid_mesh(opt, devices[0], lambda w: w.CarveBatchDevice(batches[0]), args.batch, args.cull)
                mc["mesh_check"] = compare_meshes(vdist_.merge_meshes(meshes), ref)
        except Exception as e:
            mc = {"error": "%s: %s" % (type(e).__name__, e)}
    # the streamed entry point with the producer shared by the devices (vcy_carve_batch_silhouettes_sharded) next to
    # round 4's form (every slab builds every SDF); wall time, PCIe inclusive: never `value`
    variants = None
    want = set() if args.no_variants else set(x.strip() for x in args.variants.split(","))
    if ("all" in want or "streamed" in want) and args.batch and G * k > 1:
        variants = {}
        try:
            sh2 = ShardedVoxelCarver(opt, devices, k, z_bounds=sh.z_bounds)
            if not sh2.Init():
                raise RuntimeError(vc.last_error())
            sh2.set_param("cull", args.cull)
            rec = {}
            for name, shx, flag in (("sharded_producer", sh, True), ("replicated_producer", sh2, False)):
                walls = []
                for _ in range(3):
                    shx.reset()
                    shx.sync()
                    tw = time.perf_counter()
                    if not shx.CarveBatchSilhouettes(views, masks, sharded_producer=flag):
                        raise RuntimeError(vc.last_error())
                    shx.sync()
                    walls.append((time.perf_counter() - tw) * 1e3)
                w = sorted(walls[1:])[len(walls[1:]) // 2]
                rec[name] = {"wall_ms": round(w, 3), "value_pcie_inclusive": round(float(n) ** 3 * nv / (w * 1e-3) / 1e6, 1),
                             "per_slab_producer_carve_wall_ms": [[round(x, 3) for x in t] for t in shx.last_stream_ms()]}
            rec["voxels_differing_between_the_two"] = int(sum(a.state_diff(b) for a, b in zip(sh.slabs, sh2.slabs)))
            rec["unit"] = "Mvoxel*views/s"
            rec["note"] = ("silhouettes in pageable host memory; sharded: device r uploads and transforms views r, r + G, ... of "
                           "every chunk of 32, one ncclAllGather per chunk hands every device all the images "
                           "(vcy_carve_batch_silhouettes_sharded); replicated: vcy_carve_batch_silhouettes per slab")
            variants["streamed_silhouettes"] = rec
            sh2.close()
        except Exception as e:
            variants["streamed_silhouettes"] = {"error": "%s: %s" % (type(e).__name__, e)}
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
                      "slabs_per_gpu": k, "launch": "inprocess", "devices": devices,
                      "partition": args.partition if G * k > 1 else "one slab", "plan": sh.plan_info},
           "clock_settle": {"warmup_steps_run": warm_steps, "warmup_device_ms": round(warm_ms, 2),
                            "settle_ms": args.settle_ms},
           "roofline": roofline, "mc": mc, "collective": collective,
           "per_gpu": [{"device": devices[g], "prepass_ms": round(stats[g]["prepass_ms"], 3),
                        "kernel_ms": round(stats[g]["kernel_ms"], 3), "idle_ms": round(stats[g]["idle_ms"], 3),
                        "step_ms": round(stats[g]["period_ms"], 3),
                        "slabs_z": [list(sh.z_ranges[s]) for s in range(g, G * k, G)]} for g in range(G)]}
    if preflight is not None:
        out["config"]["preflight_mesh_check"] = preflight.get("mesh_check", preflight)
        out["config"]["collective"] = preflight.get("collective")
    if why:
        out["config"]["launch_note"] = why
    if variants is not None:
        out["variants"] = variants
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
    # where the slabs are cut: equal predicted carve cost for these views (every rank computes the same cuts on its
    # own GPU from the same inputs: vcy_plan_z_slabs is deterministic), or equal thickness
    bounds, plan_info = None, None
    if world * k_slabs > 1 and args.partition == "planned":
        bounds, _, plan_info = vdist.plan_bounds(opt, local_rank, views, sdfs, world * k_slabs)
    my_slabs = vdist.slabs_of_rank(n, rank, world, k_slabs, bounds)

    def make_carvers(option, cull, slabs):
        out = []
        for _, z0, z1 in slabs:
            c = vc.VoxelCarver(option, device_id=local_rank, z_range=(z0, z1))
            if not c.Init():
                raise SystemExit("vcy_create failed: " + vc.last_error())
            # (every slab on a stream of its own: a second slab's launch fills the tail of the first one's)
            c.set_param("fused", args.batch)
            c.set_param("cull", cull)
            c.set_param("prologue", args.prologue)
            c.set_param("carvetimer", 1)  # HIP events around pre-pass and carve kernel of every launch (vcy_carve_log)
            out.append(c)
        return out

    devs = make_carvers(opt, args.cull, my_slabs)
    dev = devs[0]
    d_sdf = [dev.upload_sdf(s) for s in sdfs]  # inputs resident in HBM before the timed region

    def sync_all(carvers):
        for c in carvers:
            c.sync()

    def barrier():
        sync_all(devs)
        if dist is not None:
            if backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    def run_steps(carvers, step, count):
        """`count` steps queued back to back: nothing synchronises in between (per-step times: the event log)."""
        for _ in range(count):
            for c in carvers:
                c.reset()
            for c in carvers:
                if not step(c):
                    raise SystemExit("carve failed: " + vc.last_error())

    def clear_logs(carvers):
        for c in carvers:
            c.set_param("carvetimer", 1)

    kernel_launches = [0]

    def read_logs(carvers, steps):
        """(pre-pass ms, carve kernel ms) per step, summed over the carvers' launches; the number of carve kernel
        launches per step (a fused launch is cut into groups of brick layers) is left in kernel_launches[0]."""
        pre = ker = 0.0
        n = 0
        for c in carvers:
            log = c.carve_log()
            pre += sum(r[1] for r in log)
            ker += sum(r[2] for r in log)
            n += len(log)
        kernel_launches[0] = n / float(max(1, steps))
        return pre / max(1, steps), ker / max(1, steps)

    def warm_up(carvers, step, steps, settle_ms):
        """`steps` warm-up steps, then the same step until the device has been busy for settle_ms (clock settling, see
        the module docstring).  Returns (steps run, device-busy ms)."""
        if steps <= 0:
            return 0, 0.0
        sync_all(carvers)
        t = time.perf_counter()
        run_steps(carvers, step, steps)
        sync_all(carvers)
        busy = (time.perf_counter() - t) * 1e3
        ran = steps
        if settle_ms > 0 and busy < settle_ms:
            extra = int(math.ceil((settle_ms - busy) / (busy / steps)))
            t = time.perf_counter()
            run_steps(carvers, step, extra)
            sync_all(carvers)
            busy += (time.perf_counter() - t) * 1e3
            ran += extra
        return ran, busy

    def measure(carvers, step, reps=5, settle_ms=40.0):
        """Mean ms per step of `reps` steps queued back to back on a warm device (variants, outside the timed region);
        also (pre-pass, kernel) ms per step from the event log."""
        warm_up(carvers, step, 1, settle_ms)
        clear_logs(carvers)
        sync_all(carvers)
        t = time.perf_counter()
        run_steps(carvers, step, reps)
        sync_all(carvers)
        ms = (time.perf_counter() - t) * 1e3 / reps
        pre, ker = read_logs(carvers, reps)
        return ms, pre, ker

    batch = vc.VoxelCarver.prepare_batch(views, d_sdf)

    def main_step(c):
        return c.CarveBatchDevice(batch)

    # N > 1: before anything is timed, the WHOLE multi-GPU path once on a small grid -- sharded carve, the halo all-gather
    # through the collective backend, per-slab extraction, merge by edge key -- compared with one context holding that
    # grid: a record from a node nobody has run on before says by itself whether its exchange worked
    # (config.preflight_mesh_check, config.collective).
    preflight = None
    if world > 1 and not args.no_preflight and args.batch:
        try:
            pn, pnv, pw, ph = 256, 8, 320, 240
            popt = synth.sphere_option(pn, uo)
            pviews, pmasks = synth.sphere_views(pn, pnv, pw, ph)
            psdf = vc.make_sdf(pmasks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
            pslabs = vdist.slabs_of_rank(pn, rank, world, k_slabs)
            pcs = make_carvers(popt, args.cull, pslabs)
            pimg = pcs[0].upload_sdf(psdf)
            pbatch = vc.VoxelCarver.prepare_batch(pviews, [pimg] * pnv)
            for c in pcs:
                c.set_param("meshkeys", 1)
                if not c.CarveBatchDevice(pbatch):
                    raise RuntimeError(vc.last_error())
            pinfo = vdist.exchange_halo(pcs, rank, world) or {}
            pmeshes = [c.ExtractIsoSurface(0.0, True) for c in pcs]

            def pbarrier():
                sync_all(pcs)
                if backend == "nccl":
                    torch.cuda.synchronize()
                dist.barrier()

            pcheck = vdist.merged_mesh_check(
                pmeshes, [sid for sid, _, _ in pslabs], rank, world, world * k_slabs,
                lambda: whole_grid_mesh(popt, local_rank, lambda w: w.CarveBatchDevice(pbatch), args.batch, args.cull), pbarrier)
            preflight = {"grid": pn, "views": pnv, "collective": {k_: pinfo.get(k_) for k_ in ("backend", "ranks", "version", "op", "bytes_per_rank")},
                         "mesh_check": pcheck}
            pcs[0].free_device(pimg)
            for c in reversed(pcs):
                c.close()
        except Exception as e:
            preflight = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0 and not (isinstance(preflight.get("mesh_check"), dict) and preflight["mesh_check"].get("merged_equals_single_context")):
            sys.stderr.write("bench: PREFLIGHT MESH CHECK DID NOT PASS: %r\n" % (preflight,))

    warm_steps, warm_ms = warm_up(devs, main_step, args.warmup, args.settle_ms)
    clear_logs(devs)
    barrier()
    t0 = time.perf_counter()
    run_steps(devs, main_step, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    my_elapsed_ms = elapsed * 1e3
    avg_prepass_ms, avg_kernel_ms = read_logs(devs, args.steps) if args.batch else (0.0, my_elapsed_ms / args.steps)
    main_kernel_launches = kernel_launches[0] if args.batch else float(nv * len(devs))
    # The shader clock of THIS run, for the VALU issue fraction below: one wave on a stream of its own samples the clock
    # counter (vcy_clock_probe_*) during six more steps of the same workload queued right behind the timed region --
    # not inside it: beside the timed steps the probe's second hardware queue costs the carve kernel 1.7 %
    # (profiles/r05/clock_probe_ab.txt); an idle device would have dropped its clock, six more steps keep it where it was
    # (the clock over the second half of the probe's span is the one used: setting the probe up idles the device for a millisecond).
    live_clock = None
    if world == 1 and args.batch and "VCY_BENCH_NO_CLOCK_PROBE" not in os.environ:
        try:
            probe = vc.ClockProbe(local_rank)
            try:
                run_steps(devs, main_step, 6)
                sync_all(devs)
            finally:
                live_clock = probe.stop()
                clear_logs(devs)
        except Exception as e:
            live_clock = {"error": "%s: %s" % (type(e).__name__, e)}
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_vv = float(n) ** 3 * nv * args.steps
    value = total_vv / elapsed / 1e6
    # per rank: what its step is made of -- pre-pass (window maxima + footprint records), carve kernel (HIP events on its
    # stream), and idle = the rest of its step period (launch gaps, host) -- and the z-ranges of its slabs: a bad
    # scaling curve explains itself (an unbalanced partition shows in kernel_ms, a starved device in idle_ms)
    my_period = my_elapsed_ms / args.steps
    per_gpu = None
    if dist is not None:
        row = torch.zeros(world, 5 + 2 * len(my_slabs), dtype=torch.float64, device=red_dev)
        row[rank, 0] = float(local_rank)
        row[rank, 1], row[rank, 2] = avg_prepass_ms, avg_kernel_ms
        row[rank, 3] = max(0.0, my_period - avg_prepass_ms - avg_kernel_ms)
        row[rank, 4] = my_period
        for j, (_, z0, z1) in enumerate(my_slabs):
            row[rank, 5 + 2 * j], row[rank, 6 + 2 * j] = float(z0), float(z1)
        dist.all_reduce(row)
        rows = row.cpu().tolist()
        per_gpu = [{"rank": r, "device": int(rows[r][0]), "prepass_ms": round(rows[r][1], 3),
                    "kernel_ms": round(rows[r][2], 3), "idle_ms": round(rows[r][3], 3), "step_ms": round(rows[r][4], 3),
                    "slabs_z": [[int(rows[r][5 + 2 * j]), int(rows[r][6 + 2 * j])] for j in range(len(my_slabs))]}
                   for r in range(world)]

    # roofline of the dominant kernel (carve), this rank's slab: algorithmic bytes per launch /
    # launch duration from HIP events on the launch stream.
    def bytes_per_vv(mode, u):
        return bytes_per_voxel_view(mode, nv, u)

    slab_vox = sum(c.slab_voxels for c in devs) / float(len(devs))  # per launch
    FUSED_MAX = 64  # views per fused launch (carve_fused.hip)
    views_per_launch = min(nv, FUSED_MAX) if args.batch else 1
    # kernel launches per step: one per group of brick layers of every fused launch (the pre-pass of the next group runs
    # beside the carve of this one, vcy_set_param "overlap"), counted from the event log
    launches_per_step = max(1.0, main_kernel_launches)
    avg_launch_ms = avg_kernel_ms / launches_per_step  # the dominant kernel alone, HIP events around it
    avg_step_device_ms = my_period
    alg_bytes = sum(c.slab_voxels for c in devs) * float(nv) * bytes_per_vv(args.mode, uo) / launches_per_step
    achieved = alg_bytes / (avg_launch_ms * 1e-3) / 1e9
    ckey = "%s_%d_%d_b%d_c%d" % (args.mode, n, nv, args.batch, args.cull)
    build = library_build()
    ctr, ctr_note = load_counters(ckey, build) if world == 1 else (None, "counters are collected on one GPU")
    traffic = ctr.get("hbm_bytes_per_launch") if ctr else None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "carve_fused_kernel" if args.batch else "carve_view_kernel",
                "avg_launch_ms": round(avg_launch_ms, 4),
                "kernel_launches_per_step": round(launches_per_step, 2),
                "avg_launch_ms_note": "carve_fused_kernel alone (HIP events on its stream around every launch of it, "
                                      "vcy_carve_log): what rocprofv3 --kernel-trace reports for it; a step's fused launch is "
                                      "kernel_launches_per_step launches (groups of brick layers), algorithmic bytes per launch "
                                      "accordingly",
                "step_device_ms": round(avg_step_device_ms, 4),
                "prepass_ms_per_step": round(avg_prepass_ms, 4),
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "achieved/frac use SURVEY 8(d)'s ALGORITHMIC bytes (one state read per voxel*view, the "
                        "reference's per-view API), so frac > 1 is not a bandwidth claim: the fused kernel keeps the state "
                        "in registers across the views and drops views that provably change nothing.  Its real HBM traffic "
                        "is `traffic` (hbm_real_frac of the peak); what binds it is VALU issue: valu_issue_frac_flat2 (2 "
                        "cycles per wave instruction) and issue_floor (measured: this kernel / the same kernel without "
                        "its tile loads and stores, for the variants whose control flow does not depend on the data)"}
    if live_clock is not None:
        roofline["shader_clock_live"] = ({"ghz_settled": round(live_clock["settled_hz"] / 1e9, 4), "ghz_mean": round(live_clock["mean_hz"] / 1e9, 4),
                                          "ghz_min": round(live_clock["min_hz"] / 1e9, 4),
                                          "ghz_max": round(live_clock["max_hz"] / 1e9, 4), "samples": live_clock["samples"],
                                          "covered_ms": round(live_clock["covered_ms"], 3),
                                          "note": "s_memtime against s_memrealtime (100 MHz), one probe wave on its own stream "
                                                  "during six more steps of the same workload right behind the timed region; "
                                                  "ghz_settled = over the second half of that span (vcy_clock_probe_*)"}
                                         if "mean_hz" in live_clock else live_clock)
    if ctr is None:
        roofline["counters_note"] = ctr_note
        roofline["bound_note"] = ("`bound` is the contract's label for the algorithmic figures; no counters of this build were "
                                  "collected at this shape, so what binds the kernel here is not established (at 1024^3 x 32 it is "
                                  "VALU issue, not HBM)")
    # the roofs that actually bind the kernel: real HBM traffic and VALU issue slots (counters of the
    # committed PMC passes for this workload, duration measured live above)
    if traffic:
        t_k = ctr.get("trace_avg_ns", avg_launch_ms * 1e6) * 1e-9  # the carve kernel alone, in the profiled run
        roofline["hbm_real_gbs"] = round(traffic / t_k / 1e9, 1)
        roofline["hbm_real_frac"] = round(traffic / t_k / 1e9 / HBM_PEAK_GBS, 4)
        roofline["traffic_note"] = "carve_fused_kernel alone; the footprint pre-pass moves another %s B per step" % (
            ctr.get("prepass_hbm_bytes_per_launch", "?"))
    vcyc = valu_issue_cycles(ctr)
    roofline["bound_contract"] = "hbm: achieved / peak / frac are SURVEY 8(d)'s algorithmic HBM figures (the contract's definition)"
    if vcyc:
        # what binds the kernel is VALU issue, not HBM (hbm_real_frac ~ 0.13): `bound` names that roof and frac_binding is
        # the kernel's distance from it -- the measured issue floor where the control flow does not depend on the data
        # (cull 0, TSDF), the flat 2-cycle issue fraction for the default workload
        roofline["bound"] = "valu"
        roofline["bound_actual"] = "valu"
        # against the shader clock the profiled launch really ran at (GRBM_GUI_ACTIVE counts the busy cycles of each
        # of the 8 XCDs over the launch; 2.4 GHz is the boost clock)
        clk = CLOCK_HZ
        if ctr.get("shader_clock_hz"):  # busy cycles / duration of the same dispatch, median over the counter pass
            clk = ctr["shader_clock_hz"]
            roofline["shader_clock_ghz_profiled"] = round(clk / 1e9, 3)
        elif ctr.get("GRBM_GUI_ACTIVE") and ctr.get("trace_avg_ns"):
            clk = min(CLOCK_HZ, ctr["GRBM_GUI_ACTIVE"] / 8.0 / (ctr["trace_avg_ns"] * 1e-9))
            roofline["shader_clock_ghz_profiled"] = round(clk / 1e9, 3)
        # (the counters are of the carve kernel alone: its own duration in the profiled run, not the step's)
        t_kernel = ctr.get("trace_avg_ns", avg_launch_ms * 1e6) * 1e-9
        roofline["valu_issue_frac_flat2"] = round(vcyc / (N_SIMD * clk * t_kernel), 4)
        roofline["valu_wave_insts_per_launch"] = ctr["SQ_INSTS_VALU"]
        roofline["valu_insts_per_voxel_view"] = round(ctr["SQ_INSTS_VALU"] * 64.0 / (slab_vox * views_per_launch), 3)
        roofline["counters_source"] = ctr.get("source")
        roofline["frac_binding"] = roofline["valu_issue_frac_flat2"]
        roofline["frac_binding_note"] = ("VALU issue cycles (2 per wave instruction) / SIMD cycles of the launch, counters "
                                         "and shader clock of the PROFILED box (shader_clock_ghz_profiled)")
        if live_clock and live_clock.get("settled_hz", 0) > 1e8:
            # the same instruction count (it does not depend on the box: same kernel, same inputs) against THIS run's
            # kernel duration (HIP events) and THIS run's shader clock (the probe wave beside the timed steps)
            roofline["valu_issue_frac_flat2_live"] = round(vcyc / (N_SIMD * live_clock["settled_hz"] * avg_launch_ms * 1e-3), 4)
            roofline["frac_binding"] = roofline["valu_issue_frac_flat2_live"]
            roofline["frac_binding_note"] = ("VALU issue cycles (2 per wave instruction; the instruction count of the committed "
                                             "counter pass, which does not depend on the box) / SIMD cycles of the launch at "
                                             "THIS run's kernel duration and shader clock (shader_clock_live: a probe wave "
                                             "sampling the clock counter during six more steps right behind the timed region, second half of its span)")
    floor, _ = load_counters("issue_floor", build) if world == 1 else (None, None)
    if floor:
        roofline["issue_floor"] = {k: floor[k] for k in floor if k not in ("build",)}
        key = {"default": None, "tsdf": "tsdf"}.get(args.mode) if args.cull else "cull0"
        if key and isinstance(floor.get(key), (int, float)):
            roofline["frac_binding"] = floor[key]
            roofline["frac_binding_note"] = "measured: this kernel's rate / the rate of its own instruction stream without tile loads and stores"
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
            my_meshes = []
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
                my_meshes.append(mesh)
            mesh_check = None
            if args.verify_mesh and not single:
                mesh_check = vdist.merged_mesh_check(
                    my_meshes, [sid for sid, _, _ in my_slabs], rank, world, world * k_slabs,
                    lambda: whole_grid_mesh(opt, local_rank, lambda w: w.CarveBatchDevice(batch), args.batch, args.cull), barrier)
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
                  "mcells_per_s_wall": round(cells / (mc_wall * 1e-3) / 1e6, 1),
                  "algorithmic_frac_bricks_skipped": round(cells * 4.0 / (mc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS / max(world, 1), 4),
                  "algorithmic_frac_wall": round(cells * 4.0 / (mc_wall * 1e-3) / 1e9 / HBM_PEAK_GBS / max(world, 1), 4)}
            if mesh_check is not None:
                mc["mesh_check"] = mesh_check
            if single:
                # the same extraction reading every brick ("mcskip" 0: no use of the brick minima the carve kernel keeps)
                c = devs[0]
                c.set_param("mcskip", 0)
                dense = sorted(c.ExtractIsoSurface(0.0, True)["device_ms"] for _ in range(3))[1]
                c.set_param("mcskip", 1)
                mc["device_ms_every_brick_read"] = round(dense, 3)
                mc["roofline_frac"] = round(cells * 4.0 / (dense * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                mc["note"] = ("roofline_frac is the sweep that reads EVERY brick (device_ms_every_brick_read: 4 B per cell "
                              "really streamed); the default (device_ms) skips, by the brick minima the carve kernel keeps, "
                              "bricks left entirely outside the surface: its 4 B per cell are algorithmic only "
                              "(algorithmic_frac_bricks_skipped, may exceed 1), the bytes really moved are `traffic`")
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
    def pairs_of(c, step):
        """Fraction of the (8^3 brick, view) pairs one step really processes (the rest is dropped as provably idle)."""
        c.set_param("paircount", 1)
        c.reset()
        proc = tot = 0
        for st in (step if isinstance(step, (list, tuple)) else [step]):
            if not st(c):
                raise RuntimeError(vc.last_error())
            a, b, _ = c.last_carve_pairs()
            proc, tot = proc + a, tot + b
        c.set_param("paircount", 0)
        return round(proc / float(tot), 4) if tot else None

    pairs_main = None
    if args.batch:
        # (every rank counts the pairs of its slabs; the fraction is over the whole grid)
        proc = tot = 0
        err = None
        try:
            for c in devs:
                c.set_param("paircount", 1)
                c.reset()
                if not main_step(c):
                    raise RuntimeError(vc.last_error())
                a, b, _ = c.last_carve_pairs()
                proc, tot = proc + a, tot + b
                c.set_param("paircount", 0)
        except Exception as e:
            err = "%s: %s" % (type(e).__name__, e)
            proc = tot = 0
        if dist is not None:
            t = torch.tensor([float(proc), float(tot), 1.0 if err else 0.0], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t)
            proc, tot = t[0].item(), t[1].item()
            if t[2].item() > 0 and err is None:
                err = "pair count failed on another rank"
        pairs_main = err if err else (round(proc / float(tot), 4) if tot else None)

    # The same library in other modes, on another scene and through its other entry points, measured in the same run
    # (warm device, steps queued back to back) so that the headline's dependence on scene and call pattern is in the
    # driver's record.
    variants = None
    want = set() if args.no_variants else set(x.strip() for x in args.variants.split(","))
    if "all" in want:
        want = {"modes", "scenes", "per_view", "streamed"}
    if world == 1 and want and args.batch:
        variants = {}

        def rate_record(ms, pre, ker, mode, u, extra=None):
            bpv = bytes_per_vv(mode, u)
            rec = {"value": round(float(n) ** 3 * nv / (ms * 1e-3) / 1e6, 1), "unit": "Mvoxel*views/s",
                   "ms_per_step": round(ms, 3), "prepass_ms": round(pre, 3), "kernel_ms": round(ker, 3), "mode": mode,
                   "algorithmic_frac": round(float(n) ** 3 * nv * bpv / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "algorithmic_frac_kernel_alone": round(float(n) ** 3 * nv * bpv / (ker * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                   if ker > 0 else None}
            rec.update(extra or {})
            return rec

        for name, mode, cull in (("cull0", args.mode, 0), ("tsdf", "tsdf", 1)):
            if (mode == args.mode and cull == args.cull) or "modes" not in want:
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
                ms, pre, ker = measure(cs, lambda c: c.CarveBatchDevice(b2), reps=3)
                variants[name] = rate_record(ms, pre, ker, mode, u2, {"view_dropping": bool(cull)})
                if own:
                    cs[0].free_device(dsdf2[0])
                for c in reversed(cs):
                    c.close()
            except Exception as e:
                variants[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        roofline["value_cull0"] = variants.get("cull0", {}).get("value")
        roofline["value_tsdf"] = variants.get("tsdf", {}).get("value")
        # a figure <= 1 that follows from SURVEY 8(d)'s formula: every voxel*view evaluated (no view dropping), the carve
        # kernel alone
        roofline["frac_every_voxel_view"] = variants.get("cull0", {}).get("algorithmic_frac_kernel_alone")

        # (a) a harder scene: two overlapping off-centre spheres, a DISTINCT silhouette and SDF image per view, same
        #     grid, cameras and image size; (b) the views in two batches of V/2: the second launch reads a carved state
        #     (6.4 GB at 1024^3) instead of starting from a fresh grid.  voxel_carver.cc:442-491 costs the same whatever
        #     the scene; this path does not, and the line says by how much.
        try:
            if "scenes" not in want:
                raise StopIteration
            cs = make_carvers(opt, args.cull, my_slabs)
            c0 = cs[0]
            hv, hm = synth.blob_views(n, nv, args.width, args.height)
            himg = [c0.make_sdf_device(m, use_truncation=bool(uo.use_truncation), band=uo.truncation_band) for m in hm]
            hb = vc.VoxelCarver.prepare_batch(hv, himg)
            ms, pre, ker = measure(cs, lambda c: c.CarveBatchDevice(hb), reps=5)
            variants["hard_scene"] = rate_record(ms, pre, ker, args.mode, uo, {
                "scene": "two overlapping off-centre spheres (radii 0.24 N and 0.17 N), %d distinct SDF images" % nv,
                "pairs_processed_frac": pairs_of(c0, lambda c: c.CarveBatchDevice(hb))})
            for pimg in himg:
                c0.free_device(pimg)
            half = nv // 2
            ba = vc.VoxelCarver.prepare_batch(views[:half], d_sdf[:half])
            bb = vc.VoxelCarver.prepare_batch(views[half:], d_sdf[half:])
            ms, pre, ker = measure(cs, lambda c: c.CarveBatchDevice(ba) and c.CarveBatchDevice(bb), reps=5)
            variants["two_batches"] = rate_record(ms, pre, ker, args.mode, uo, {
                "batches": [half, nv - half],
                "pairs_processed_frac": pairs_of(c0, [lambda c: c.CarveBatchDevice(ba), lambda c: c.CarveBatchDevice(bb)])})
            # The timed step repeats ONE set of views, so the tables derived from the views (x tables, bounding boxes, the
            # view records: host work + a 0.5 MB upload) come from the context's one-entry cache every step.  A sequence of NEW
            # views per launch pays them: two sets -- the views, and the same views rotated by one position -- alternate here,
            # so every launch misses the cache; same work on the device.
            rot = views[1:] + views[:1]
            bsets = [batch, vc.VoxelCarver.prepare_batch(rot, d_sdf[1:] + d_sdf[:1])]
            flip = [0]

            def alternating(c):
                flip[0] ^= 1
                return c.CarveBatchDevice(bsets[flip[0]])

            ms, pre, ker = measure(cs, alternating, reps=6)
            variants["new_views_every_step"] = rate_record(ms, pre, ker, args.mode, uo, {
                "view_table_cache": "missed by every launch (two alternating view sets); the headline step hits it",
                "ms_per_step_over_headline": round(ms - elapsed / args.steps * 1e3, 3)})
            for c in reversed(cs):
                c.close()
        except StopIteration:
            pass
        except Exception as e:
            variants.setdefault("hard_scene", {"error": "%s: %s" % (type(e).__name__, e)})
            variants.setdefault("two_batches", {"error": "%s: %s" % (type(e).__name__, e)})

        # The reference's own call pattern (examples.cc:117-149): `for each view: Carve(one view); ExtractIsoSurface()`.
        # An extraction between two views means every view is a launch of its own (nothing to fuse across): the
        # single-view path, where a wave drops its view against the brick minimum the previous launch left
        # before it reads any state.  "defer" 0 so that the carve is timed by itself (HIP events around each call);
        # per_view_defer0 is the same loop without the extractions, per_view_tsdf that loop in TSDF mode
        # (VoxelUpdate::kWeightedAverage + truncation: every view changes nearly every brick).
        for name, extract, mode in (("per_view_interleaved", True, args.mode), ("per_view_defer0", False, args.mode),
                                    ("per_view_tsdf", False, "tsdf")):
            if (name == "per_view_tsdf" and args.mode == "tsdf") or "per_view" not in want:
                continue
            try:
                u2 = update_option(mode)
                cs = make_carvers(synth.sphere_option(n, u2), args.cull, my_slabs)
                c0 = cs[0]
                c0.set_param("defer", 0)
                c0.set_param("meshkeys", 0)
                if mode == args.mode:
                    dimg, own = d_sdf, False
                else:
                    s2 = vc.make_sdf(masks[0], use_truncation=bool(u2.use_truncation), band=u2.truncation_band)
                    dimg, own = [c0.upload_sdf(s2)] * nv, True
                carve_ms, mc_dev, mc_wall, t_wall = [], [], [], None
                for rep in range(2):  # the second pass is the one reported (buffers allocated, sizes guessed, clocks up)
                    c0.reset()
                    carve_ms, mc_dev, mc_wall = [], [], []
                    c0.set_param("carvetimer", 1)
                    c0.sync()
                    tw = time.perf_counter()
                    for i in range(nv):
                        if not c0.CarveDevice(views[i], dimg[i]):
                            raise RuntimeError(vc.last_error())
                        if extract:
                            mesh = c0.ExtractIsoSurface(0.0, True)
                            mc_dev.append(mesh["device_ms"])
                            mc_wall.append(mesh["wall_ms"])
                    c0.sync()
                    t_wall = (time.perf_counter() - tw) * 1e3
                    carve_ms = [r[1] + r[2] for r in c0.carve_log()]  # per launch: pre-pass + kernel (no sync between views)
                tot = sum(carve_ms)
                rec = {"value": round(float(n) ** 3 * nv / (tot * 1e-3) / 1e6, 1), "unit": "Mvoxel*views/s", "mode": mode,
                       "carve_ms_total": round(tot, 3), "carve_ms_first_view": round(carve_ms[0], 3),
                       "carve_ms_per_view_after_first": round((tot - carve_ms[0]) / max(1, nv - 1), 3),
                       "algorithmic_frac": round(float(n) ** 3 * nv * bytes_per_vv(mode, u2) / (tot * 1e-3) / 1e9
                                                 / HBM_PEAK_GBS, 4),
                       "loop_wall_ms": round(t_wall, 2), "launches": nv, "defer": 0}
                if extract:
                    cells = float(n - 1) ** 3
                    med = sorted(mc_dev)[len(mc_dev) // 2]
                    rec["mc_device_ms_median"] = round(med, 3)
                    rec["mc_wall_ms_median"] = round(sorted(mc_wall)[len(mc_wall) // 2], 3)
                    rec["mc_mcells_per_s"] = round(cells / (med * 1e-3) / 1e6, 1)
                variants[name] = rec
                if own:
                    c0.free_device(dimg[0])
                for c in reversed(cs):
                    c.close()
            except Exception as e:
                variants[name] = {"error": "%s: %s" % (type(e).__name__, e)}

        # The streamed entry point (BASELINE configs[4]: "streamed per-view SDF upload overlapped with fuse"; replaces
        # Carve(vector<Camera>, vector<Image1b>), voxel_carver.cc:516-528 around :394-413): silhouettes in HOST memory ->
        # page-locked staging -> DMA -> SDF build on the device -> fused carve, in chunks of 32 views with chunk i + 1
        # produced while chunk i is carved.  Wall time of the call (PCIe inclusive: never `value`).
        try:
            if "streamed" not in want:
                raise StopIteration
            cs = make_carvers(opt, args.cull, my_slabs)
            c0 = cs[0]
            walls = []
            for rep in range(4):
                c0.reset()
                c0.sync()
                tw = time.perf_counter()
                if not c0.CarveBatchSilhouettes(views, masks):
                    raise RuntimeError(vc.last_error())
                walls.append(((time.perf_counter() - tw) * 1e3,) + c0.last_stream_ms())
            wall, prod, carve, wall_lib = sorted(walls[1:])[len(walls[1:]) // 2]
            variants["streamed_silhouettes"] = {
                "wall_ms": round(wall, 3), "producer_ms": round(prod, 3), "carve_ms": round(carve, 3),
                "overlap": round(max(prod, carve) / wall_lib, 4), "chunks": (nv + 31) // 32,
                "value_pcie_inclusive": round(float(n) ** 3 * nv / (wall * 1e-3) / 1e6, 1), "unit": "Mvoxel*views/s",
                "note": "vcy_carve_batch_silhouettes from pageable host masks; producer = staging copy + H2D + device SDF "
                        "build per chunk of 32 views, carve = the fused launches; overlap = max(producer, carve) / wall: "
                        "1.0 when the shorter side is hidden completely; with a single chunk (<= 32 views) nothing can "
                        "overlap and overlap = the longer side's share of the wall time"}
            for c in reversed(cs):
                c.close()
        except StopIteration:
            pass
        except Exception as e:
            variants["streamed_silhouettes"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # N > 1: the streamed entry point (silhouettes in host memory -> SDF images -> fused carve, BASELINE configs[4]) with
    # the producer SHARED by the ranks (rank r builds the SDFs of views r, r + N, ... of every chunk, one all-gather per
    # chunk: vacancy_amd.dist.carve_silhouettes_sharded) next to round 4's form (every rank builds every SDF), and a
    # device-side comparison of the two states.  Wall time of the slowest rank, PCIe inclusive: never `value`.
    if world > 1 and "streamed" in want and args.batch:
        variants = variants or {}
        try:
            ca, cb = make_carvers(opt, args.cull, my_slabs), make_carvers(opt, args.cull, my_slabs)

            def timed(carvers, fn, reps=3):
                walls, last = [], None
                for _ in range(reps):
                    for c in carvers:
                        c.reset()
                    sync_all(carvers)
                    if dist is not None:
                        dist.barrier()
                    tw = time.perf_counter()
                    last = fn(carvers)
                    sync_all(carvers)
                    if dist is not None:
                        dist.barrier()
                    walls.append((time.perf_counter() - tw) * 1e3)
                w = sorted(walls[1:] or walls)[len(walls[1:] or walls) // 2]
                t = torch.tensor([w], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item()), last

            def replicated(carvers):
                for c in carvers:
                    if not c.CarveBatchSilhouettes(views, masks):
                        raise RuntimeError(vc.last_error())
                return None

            w_sh, info = timed(ca, lambda cs_: vdist.carve_silhouettes_sharded(cs_, rank, world, views, masks))
            w_rep, _ = timed(cb, replicated)
            diff = torch.tensor([float(sum(x.state_diff(y) for x, y in zip(ca, cb)))], dtype=torch.float64, device=red_dev)
            dist.all_reduce(diff)
            variants["streamed_silhouettes"] = {
                "sharded_producer": {"wall_ms": round(w_sh, 3), "value_pcie_inclusive": round(float(n) ** 3 * nv / (w_sh * 1e-3) / 1e6, 1),
                                     "rank0": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (info or {}).items()}},
                "replicated_producer": {"wall_ms": round(w_rep, 3),
                                        "value_pcie_inclusive": round(float(n) ** 3 * nv / (w_rep * 1e-3) / 1e6, 1)},
                "voxels_differing_between_the_two": int(diff.item()), "unit": "Mvoxel*views/s",
                "note": "silhouettes in pageable host memory; sharded: rank r uploads and transforms views r, r + N, ... of "
                        "every chunk of 32 (vcy_make_sdf_batch_device), one all-gather per chunk (backend %s) hands every rank "
                        "all the images, fused carve; replicated: vcy_carve_batch_silhouettes on every rank (every GPU builds "
                        "every SDF); wall of the slowest rank, median of the repetitions after the first" % backend}
            for c in reversed(ca + cb):
                c.close()
        except Exception as e:
            variants["streamed_silhouettes"] = {"error": "%s: %s" % (type(e).__name__, e)}

    out = {
        "metric": "Mvoxel*views/s (Carve)", "value": round(value, 1), "unit": "Mvoxel*views/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d^3 grid x %d views at %dx%d, %s mode, z-slab sharded over %d GPU(s)"
                               % (n, nv, args.width, args.height, args.mode, world),
                   "grid": n, "views": nv, "image": [args.width, args.height], "mode": args.mode,
                   "fused_views_per_launch": views_per_launch, "view_dropping": bool(args.cull),
                   "slabs_per_gpu": k_slabs, "partition": args.partition if world * k_slabs > 1 else "one slab",
                   "plan": plan_info, "pairs_processed_frac": pairs_main,
                   "division": {2: "rcp + 3 (verified for this focal length)", 1: "rcp + 5 (verified)",
                                0: "full IEEE expansion"}.get(dev.get_param("div_level"), "?")},
        "clock_settle": {"warmup_steps_run": warm_steps, "warmup_device_ms": round(warm_ms, 2),
                         "settle_ms": args.settle_ms,
                         "note": "the warm-up runs --warmup steps and then the same step until the device has been busy "
                                 "for settle_ms: an idle MI355X needs ~40 ms of continuous load to settle its clocks"},
        "roofline": roofline, "mc": mc, "collective": collective,
    }
    if preflight is not None:
        out["config"]["preflight_mesh_check"] = preflight.get("mesh_check", preflight)
        out["config"]["collective"] = preflight.get("collective")
    if per_gpu is not None:
        out["per_gpu"] = per_gpu
        out["config"]["launch"] = "torch.distributed.run (one process per GPU)"
        if isinstance(out["collective"], dict):
            out["collective"]["launch"] = "one process per GPU (torch.distributed, backend %s)" % backend
    if variants is not None:
        out["variants"] = variants
    # BASELINE.json's other single-GPU configurations in the same line (the driver only runs `bench.py --gpus 1`):
    # configs[0] the bunny sequence of examples.cc, configs[1] 512^3 x 16 TSDF, and the configs[4] shape 2048^3 x 64 on
    # this one GPU.  After the timed region and the variants, each on contexts of its own; < 60 s together.
    headline = (n, nv, args.width, args.height, args.mode, args.cull) == (1024, 32, 1280, 720, "default", 1)
    if rank == 0 and world == 1 and args.batch and headline and not (args.no_configs or args.no_variants):
        t_cfg = time.perf_counter()
        out["configs"] = {
            "note": "BASELINE.json configs other than the headline (configs[2]; configs[3] is the same grid with --gpus N), "
                    "measured after the timed region on contexts of their own; `value` of this line is the headline's only",
            "configs[0]": bunny_sequence(local_rank),
            "configs[1]": side_config(local_rank, build, "configs[1]", 512, 16, 640, 480, "tsdf", steps=20, warmup=3),
            "configs[4] shape on one GPU": side_config(local_rank, build, "configs[4] on 1 of its 8 GPUs' worth of hardware",
                                                        2048, 64, 1920, 1080, "default", steps=3, warmup=1, settle_ms=0.0),
        }
        out["configs"]["wall_s_spent"] = round(time.perf_counter() - t_cfg, 2)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, views, sdfs, args.cpu_seconds)
    if args.batch:
        out["roofline"] = contract_roofline(roofline, mode=args.mode, cull=args.cull, n=n, nv=nv, bpv=bytes_per_vv(args.mode, uo),
                                            launches_per_step=launches_per_step, pairs_frac=pairs_main, variants=variants, mc=mc,
                                            configs=out.get("configs"), build=build, world=world)
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
