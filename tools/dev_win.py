"""Window conv kernel (tl_conv_win) vs the register-gather kernels on the REAL config-2 rulebooks: same results, time per launch.
    python tools/dev_win.py [reps=30]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import _hip, ops
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
_hip.WIN_KERNEL = True                      # column-form rulebooks on every big level
g = build_geometry(b["coords"].cuda(), b["batch_ids"].cuda(), 1, 0.1, 7, [500, 500, 1000])
L = _hip.lib()
def tune(**kw):
    for k, v in kw.items():
        _hip.check(L.tl_set_tuning(k.encode(), v), k)
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
for level, cin, cout in ((1, 64, 64), (1, 128, 64), (2, 96, 96), (2, 192, 96), (3, 128, 128), (3, 256, 128), (0, 32, 32), (0, 64, 32)):
    lv = g.levels[level]
    x = torch.randn((lv.n, cin), device="cuda", generator=gen).bfloat16()
    res = torch.randn((lv.n, cout), device="cuda", generator=gen).bfloat16()
    w = ops.pack_weight(torch.randn((cout, 3, 3, 3, cin), device="cuda", generator=gen) / (27 * cin) ** 0.5, torch.bfloat16)
    sc = torch.rand(cout, device="cuda") + 0.5; sh = torch.randn(cout, device="cuda")
    o2 = torch.empty((lv.n, cout), dtype=torch.bfloat16, device="cuda")
    run = lambda: ops.conv_fwd(x, w, lv.nbr, lv.n, residual=res, out2=(o2, sc, sh, True))
    line = f"level {level + 1} {cin:3d}->{cout:3d} N={lv.n:8d}:"
    tune(win=0); ref = run().float(); t0 = timeit(run); line += f"  gather kernels {t0:.3f} ms"
    for wr, ct in ((0, 1), (0, 0), (512, 1)):
        tune(win=2, win_min_rows=0, win_rows=wr, win_ct=ct)
        out = run().float(); t = timeit(run)
        err = float((out - ref).abs().max() / ref.abs().max())
        line += f" | window {'4w/256' if wr == 0 else '8w/512'}{' ct' if ct else ''}: {t:.3f} ms (x{t0 / t:.2f}, diff {err:.1e})"
    tune(win=0, win_min_rows=65536, win_rows=0, win_ct=1)
    print(line, flush=True)
