#!/bin/bash
# Runs ON the GPU box: marching cubes of the bench scene (1024^3) with the one-sweep cell search and with the
# bit planes in memory ("mcsweep" 1 / 0), same context, alternating; kernel ms, wall ms, mesh hash.
#   profiles/tools/ab_mc_sweep.sh [<variant>...]     (default "prod" = vacancy_amd/csrc/libvacancy_hip.so)
[ $# -eq 0 ] && set -- prod
for v in "$@"; do
  lib=build/variants/$v/libvacancy_hip.so
  [ "$v" = "prod" ] && lib=vacancy_amd/csrc/libvacancy_hip.so
  VCY_HIP_LIB=$lib python - "$v" <<'PY'
import hashlib, sys
sys.path.insert(0, ".")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n, nv = 1024, 32
opt = synth.sphere_option(n, UpdateOption())
views, masks = synth.sphere_views(n, nv, 1280, 720)
sdf0 = vc.make_sdf(masks[0])
c = vc.VoxelCarver(opt)
assert c.Init()
d = [c.upload_sdf(sdf0)] * nv
assert c.CarveBatchDevice(vc.VoxelCarver.prepare_batch(views, d))
for rnd in range(2):
    for sweep in (1, 0):
        c.set_param("mcsweep", sweep)
        every = []
        best = (1e9, 1e9)
        for it in range(5):
            m = c.ExtractIsoSurface(0.0, True)
            best = min(best, (m["device_ms"], m["wall_ms"]))
            every.append("%.3f" % m["device_ms"])
        h = hashlib.sha1(m["vertices"].tobytes() + m["faces"].tobytes() + m["keys"].tobytes()).hexdigest()[:12]
        print("%-8s mcsweep %d  device %.3f ms  wall %.3f ms  verts %d faces %d  mesh %s  (%s)" % (sys.argv[1], sweep, best[0], best[1], len(m["vertices"]), len(m["faces"]), h, " ".join(every)))
PY
done
