"""Board power and shader clock while ONE conv layer loops for a few seconds (rocm-smi sampled from a thread), for the layers that
carry the forward of config 2 -- the evidence behind DESIGN.md's "the level-2/3 kernels run at the board power cap".  Output is a
table; `python tools/power_kernels.py > profiles/<tag>/power_per_layer.txt` on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import power_probe
from treelearn_amd import geometry as G, ops
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
geom = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 7, [500, 500, 1000])
# level 1 as the product runs it: rows in the block-local order, the staged-unit kernel (k_conv_blk); the 64 -> 32 decoder conv as its two halves
blk = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 7, [500, 500, 1000], blocked=True)
print(f"{'layer':24s} {'rows':>9s} {'ms':>7s} {'TFLOP/s (present pairs)':>24s} {'sclk MHz':>9s} {'board W':>8s}")
for li, ci, co in ((0, 32, 32), (0, -64, 32), (1, 64, 64), (1, 128, 64), (2, 96, 96), (2, 192, 96), (3, 128, 128), (4, 160, 160)):
    lv = geom.levels[li]
    halves = ci < 0
    ci = abs(ci)
    x = torch.randn(lv.n, ci, device="cuda").bfloat16()
    w = ops.pack_weight(torch.randn(co, 3, 3, 3, ci, device="cuda") * 0.05, torch.bfloat16)
    res = torch.randn(lv.n, co, device="cuda").bfloat16()
    out = torch.empty(lv.n, co, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.conv_fwd(x, w, lv.nbr, lv.n, out=out, residual=res)
    tag = ""
    if li == 0:
        bn = blk.levels[0].nbr
        if halves:                                                # two staged 32 -> 32 launches, the second takes the first one's result as residual
            wa = ops.pack_weight(torch.randn(co, 3, 3, 3, 32, device="cuda") * 0.05, torch.bfloat16); part = torch.empty_like(out)
            def f():
                ops.conv_fwd(x[:, :32], wa, bn, lv.n, out=part)
                ops.conv_fwd(x[:, 32:], wa, bn, lv.n, out=out, residual=part)
            tag = " (2 staged halves)"
        else:
            f = lambda: ops.conv_fwd(x, w, bn, lv.n, out=out, residual=res)
            tag = " (staged units)"
    f(); torch.cuda.synchronize()
    pw = power_probe(f, seconds=3.0) or {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    pairs = int((lv.nbr >= 0).sum())
    print(f"l{li+1} subm {ci:3d}->{co:3d}{tag}".ljust(24), f"{lv.n:9d} {ms:7.3f} {2.0 * pairs * ci * co / ms / 1e9:24.0f} {pw.get('sclk_mhz', 0):9d} {pw.get('board_w', 0.0):8.0f}", flush=True)
