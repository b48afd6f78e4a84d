"""Runs ON the GPU box: phases of vcy_extract_voxel on the bunny (VCY_XV_TIMING=1 prints them on stderr).
usage: VCY_XV_TIMING=1 python profiles/tools/xv_timing.py [resolution]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bunny_data as B
from vacancy_amd import carver as vc, synth
res = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
views = B.bunny_views(lambda t, q: synth.affine_inverse(synth.pose_from_tum(t, q)))
masks = B.load_masks()
c = vc.VoxelCarver(B.bunny_option(res)); assert c.Init()
for rep in range(2):
    c.reset()
    for v, m in zip(views, masks):
        assert c.CarveSilhouette(v, m); c.sync()
        t = time.perf_counter(); r = c.ExtractVoxel(False, arrays=False)
        print("rep %d: ExtractVoxel call %.2f ms, %d vertices" % (rep, (time.perf_counter() - t) * 1e3, r["n_vertices"]), flush=True)
