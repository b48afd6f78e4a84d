"""Runs ON the GPU box: how fast this device WRITES (nothing read), next to the launches of this library that only write --
the first single-view launch on a fresh 1024^3 grid stores 5.4 GB (4 B sdf + 1 B update_num per voxel) in 1.87-1.92 ms.
torch is plumbing here: fill_ kernels of the same byte count, float32 and uint8, and a device-to-device copy."""
import torch
n = 1024 ** 3
dev = torch.device("cuda", 0)
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.uint8, device=dev)
c = torch.empty(n, dtype=torch.float32, device=dev)


def timed(fn, reps=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def both():
    a.fill_(1.5); b.fill_(1)


t_f = timed(lambda: a.fill_(1.5)); t_b = timed(lambda: b.fill_(1)); t_fb = timed(both); t_c = timed(lambda: c.copy_(a))
print("fill f32 4.29 GB: %.3f ms = %.2f TB/s" % (t_f, 4 * n / t_f / 1e9))
print("fill u8  1.07 GB: %.3f ms = %.2f TB/s" % (t_b, n / t_b / 1e9))
print("both (5.37 GB, two launches): %.3f ms = %.2f TB/s" % (t_fb, 5 * n / t_fb / 1e9))
print("copy f32 4.29 GB read + 4.29 GB written: %.3f ms = %.2f TB/s (read + write)" % (t_c, 8 * n / t_c / 1e9))
