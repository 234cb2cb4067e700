"""What this part sustains for pure writes, pure reads and a copy (ATen kernels over 1 GiB, HIP events): the rate second output views and
fp32 outputs are paid at."""
import torch
n = 1 << 28                                             # 1 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
a.normal_()


def timed(f, reps=20):
    for _ in range(3): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gib = n * 4 / 1e9
t = timed(lambda: b.fill_(1.0)); print("fill   (write only)        %.3f ms  %.2f TB/s" % (t, gib / t))
t = timed(lambda: a.sum());      print("sum    (read only)         %.3f ms  %.2f TB/s" % (t, gib / t))
t = timed(lambda: b.copy_(a));   print("copy   (read + write)      %.3f ms  %.2f TB/s moved" % (t, 2 * gib / t))
h = a[: n // 2].view(torch.int32)
t = timed(lambda: torch.add(a[: n // 2], 1.0, out=b[: n // 2])); print("add    (read + write, 0.5 GiB each) %.3f ms  %.2f TB/s moved" % (t, gib / t))
