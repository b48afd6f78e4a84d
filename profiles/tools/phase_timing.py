"""s_memtime breakdown of the fused carve kernel (development build, -DVCY_PHASE_TIMING).

  profiles/tools/build_variant.sh phase -DVCY_PHASE_TIMING
  VCY_HIP_LIB=build/variants/phase/libvacancy_hip.so python profiles/tools/phase_timing.py [--grid 1024 ...]

Runs the bench workload (1024^3 x 32 views, sphere scene) in three configurations -- default, view
dropping off, TSDF -- and prints, per configuration, the share of wave time each phase of the kernel
takes and how many (brick, view) pairs each of the three view loops processed.
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from vacancy_amd import capi, synth  # noqa: E402
from vacancy_amd import carver as vc  # noqa: E402
from vacancy_amd.capi import UpdateOption  # noqa: E402

PHASES = ["prologue+load", "tile staging", "view: select-free", "view: sure", "view: checked", "re-bound", "write-back"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--views", type=int, default=32)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    a = ap.parse_args()
    lib = capi.load()
    ticks = lib.vcy_debug_phase_ticks
    ticks.restype = C.c_int
    ticks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    out = {}
    for name, mode, cull in (("default", "default", 1), ("cull0", "default", 0), ("tsdf", "tsdf", 1)):
        uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
        opt = synth.sphere_option(a.grid, uo)
        views, masks = synth.sphere_views(a.grid, a.views, a.width, a.height)
        sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
        c = vc.VoxelCarver(opt)
        assert c.Init(), vc.last_error()
        c.set_param("cull", cull)
        d = [c.upload_sdf(sdf0)] * a.views
        batch = vc.VoxelCarver.prepare_batch(views, d)
        for it in range(2):
            c.reset()
            buf = (C.c_ulonglong * 16)()
            assert ticks(buf, 1) == 0  # clear
            c.timer_begin()
            assert c.CarveBatchDevice(batch), vc.last_error()
            ms = c.timer_end()
            assert ticks(buf, 1) == 0
        t = [int(x) for x in buf]
        tot = float(sum(t[:7]))
        pairs = t[7] + t[8] + t[9]
        rec = {"kernel_ms_instrumented": round(ms, 3), "waves": t[10],
               "phase_share": {p: round(t[i] / tot, 4) for i, p in enumerate(PHASES)},
               "ticks_per_wave": round(tot / max(1, t[10]), 1),
               "pairs": {"select-free": t[7], "sure": t[8], "checked": t[9]},
               "pairs_per_wave": round(pairs / max(1, t[10]), 2),
               "pairs_that_changed_the_brick": round(t[11] / max(1, pairs), 4),
               "ticks_per_pair": {"select-free": round(t[2] / max(1, t[7]), 1), "sure": round(t[3] / max(1, t[8]), 1),
                                  "checked": round(t[4] / max(1, t[9]), 1),
                                  "staging": round(t[1] / max(1, pairs), 1), "re-bound": round(t[5] / max(1, pairs), 1)}}
        rec["prologue_ticks_per_wave"] = {"arguments + axis tables": round(t[12] / max(1, t[10]), 1),
                                          "brick_footprints": round((t[13] - t[12]) / max(1, t[10]), 1),
                                          "state, live set, first tile request": round((t[0] - t[13]) / max(1, t[10]), 1)}
        out[name] = rec
        print(name, json.dumps(rec))
        del c
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "phase_timing.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
