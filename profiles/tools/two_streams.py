"""Runs ON the GPU box: is there anything to gain from running the pre-pass of one fused launch beside the carve kernel
of another?  Two contexts (two streams) on ONE device, the headline step on each: `steps` steps on A then `steps` on B
(one after the other) against the same steps enqueued alternately on both streams (the device interleaves them)."""
import sys, time
sys.path.insert(0, ".")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption

n, nv, steps = 1024, 32, 10
opt = synth.sphere_option(n, UpdateOption())
views, masks = synth.sphere_views(n, nv, 1280, 720)
sdf0 = vc.make_sdf(masks[0])
cs = []
for _ in range(2):
    c = vc.VoxelCarver(opt)
    assert c.Init()
    imgs = [c.upload_sdf(sdf0) for _ in range(nv)]
    cs.append((c, vc.VoxelCarver.prepare_batch(views, imgs)))


def run(order):
    for c, _ in cs:
        c.sync()
    t = time.perf_counter()
    for i in order:
        c, b = cs[i]
        c.reset()
        assert c.CarveBatchDevice(b)
    for c, _ in cs:
        c.sync()
    return (time.perf_counter() - t) * 1e3


run([0, 1] * 3)
for rep in range(3):
    seq = run([0] * steps + [1] * steps)
    alt = run([0, 1] * steps)
    one = run([0] * (2 * steps))
    print("rep %d: %d steps one stream %.2f ms (%.3f per step) | A then B %.2f | A and B interleaved on two streams %.2f (%.3f per step)"
          % (rep, 2 * steps, one, one / (2 * steps), seq, alt, alt / (2 * steps)))
