"""Runs ON the GPU box: the STREAMED path (silhouettes in host memory -> SDF images -> fused carve; BASELINE configs[4],
reference voxel_carver.cc:516-528 around :394-413) of a G-GPU run, emulated rank by rank on ONE GPU.

Per rank of G (planned cuts, one slab per GPU) and per chunk of 32 views:
  P_all   producer for every view of the chunk   (round 4: every GPU builds every SDF)
  P_share producer for the rank's share, views r, r + G, ...   (vcy_make_sdf_batch_device, staging copy + H2D + transform)
  A       the all-gather of the chunk's images: NOT measurable on one GPU -- bytes received / an ASSUMED bus bandwidth
          (VCY_ALLGATHER_GBS, default 300 GB/s: what rccl-tests report for large all-gathers over 7 xGMI links on this
          class of node; the table is printed for half of that as well)
  C       the rank's fused carve of the chunk at steady clocks, images resident
and the pipeline of vcy_carve_batch_silhouettes(_sharded): chunk i + 1 is produced (and gathered) while chunk i is carved:
  wall = P + A + (chunks - 1) * max(C, P + A) + C.
Prints per G the slowest rank's wall for both producers and the speed-up over G = 1."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ctypes as C  # noqa: E402

from vacancy_amd import carver as vc, synth, dist as vdist  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402

CONFIGS = [(1024, 32, 1280, 720), (2048, 64, 1920, 1080)]
if os.environ.get("VCY_STREAM_CONFIGS"):
    CONFIGS = [tuple(int(x) for x in c.split("x")) for c in os.environ["VCY_STREAM_CONFIGS"].split(",")]
STEPS = int(os.environ.get("VCY_PLAN_STEPS", "24"))
BUS = float(os.environ.get("VCY_ALLGATHER_GBS", "300"))
CHUNK = 32

for n, nv, w, h in CONFIGS:
    views, masks = synth.sphere_views(n, nv, w, h)
    opt = synth.sphere_option(n, UpdateOption())
    sdf0 = vc.make_sdf(masks[0])
    chunks = (nv + CHUNK - 1) // CHUNK
    m = min(CHUNK, nv)

    def producer(count):
        """ms to build `count` SDF images from host silhouettes into one device allocation (median of 5)."""
        c = vc.VoxelCarver(opt, device_id=0, z_range=(0, 8))
        assert c.Init()
        buf = C.c_void_p()
        stride = w * h * 4
        assert c._lib.vcy_device_alloc(c.ctx, count * stride, C.byref(buf)) == 0
        outs = [buf.value + i * stride for i in range(count)]
        ts = []
        for _ in range(6):
            c.sync()
            t = time.perf_counter()
            assert c.make_sdf_batch_into(views[:count], masks[:count], outs)
            ts.append((time.perf_counter() - t) * 1e3)
        c.free_device(buf)
        c.close()
        return sorted(ts[1:])[2]

    def steady(z0, z1):
        """ms per fused carve of one chunk over slab [z0, z1) at steady clocks (images resident)."""
        c = vc.VoxelCarver(opt, device_id=0, z_range=(z0, z1))
        assert c.Init()
        d = c.upload_sdf(sdf0)
        batch = vc.VoxelCarver.prepare_batch(views[:m], [d] * m)
        c.reset(); c.CarveBatchDevice(batch); c.sync()
        c.set_param("carvetimer", 1)
        for _ in range(STEPS):
            c.reset()
            c.CarveBatchDevice(batch)
        c.sync()
        log = c.carve_log()
        c.free_device(d); c.close()
        starts = [r[0] for r in log if r[3]]  # (a launch may be several records: groups of brick layers)
        k = max(4, STEPS // 3)
        return (starts[-1] - starts[-1 - k]) / k

    p_all = producer(m)
    print("== %d^3 x %d views at %dx%d: %d chunk(s) of %d views; producer for a whole chunk %.3f ms (%.3f ms per image)"
          % (n, nv, w, h, chunks, m, p_all, p_all / m))
    base = None
    for G in (1, 2, 4, 8):
        if G == 1:
            bounds = [0, n]
        else:
            bounds, _, _ = vdist.plan_bounds(opt, 0, views, [sdf0] * nv, G)
        carve = [steady(bounds[r], bounds[r + 1]) for r in range(G)]
        share = (m + G - 1) // G
        p_share = producer(share)
        img_bytes = w * h * 4
        rows = []
        for bus in (BUS, BUS / 2):
            gather = 0.0 if G == 1 else (G - 1) * share * img_bytes / (bus * 1e9) * 1e3 + 0.03
            worst = {"sharded": 0.0, "replicated": 0.0}
            for r in range(G):
                c_ = carve[r]
                for name, p, a in (("sharded", p_share, gather), ("replicated", p_all, 0.0)):
                    wall = p + a + (chunks - 1) * max(c_, p + a) + c_
                    worst[name] = max(worst[name], wall)
            rows.append((bus, gather, worst))
        if G == 1:
            base = rows[0][2]["replicated"]
        for bus, gather, worst in rows:
            print("G=%d  carve/chunk per rank %s ms; share %d image(s): producer %.3f ms; all-gather (assumed %.0f GB/s) %.3f ms | "
                  "streamed wall: sharded %.3f ms (%.2fx), replicated %.3f ms (%.2fx)"
                  % (G, [round(x, 3) for x in carve], share, p_share, bus, gather, worst["sharded"], base / worst["sharded"],
                     worst["replicated"], base / worst["replicated"]))
            if G == 1:
                break
