import sys, os
sys.path.insert(0, os.getcwd())
import torch
from treelearn_amd import _hip, ops
L = _hip.lib()
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, cin, cout in ((6264, 160, 160), (6264, 320, 160), (12000, 128, 128), (4500, 192, 192), (16000, 96, 96)):
    x = torch.randn((n, cin), device="cuda", generator=gen).bfloat16(); res = torch.randn((n, cout), device="cuda", generator=gen).bfloat16()
    tab = torch.randint(-1, n, (27, n), device="cuda", generator=gen, dtype=torch.int64).to(torch.int32)
    tab[torch.rand((27, n), device="cuda", generator=gen) < 0.2] = -1
    w = ops.pack_weight(torch.randn((cout, 3, 3, 3, cin), device="cuda", generator=gen) / (27 * cin) ** 0.5, torch.bfloat16)
    run = lambda: ops.conv_fwd(x, w, tab, n, residual=res)
    a = run().float(); t_full = timeit(run)
    _hip.check(L.tl_set_tuning(b"small_mode", 1), "sm"); b = run().float(); t_old = timeit(run); _hip.check(L.tl_set_tuning(b"small_mode", 0), "sm")
    print(f"n {n} {cin}->{cout}: full-width {t_full:.1f} us, 32x32/32x64 blocks {t_old:.1f} us, max diff {float((a - b).abs().max() / b.abs().max()):.1e}", flush=True)
