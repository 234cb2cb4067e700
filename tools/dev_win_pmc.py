"""One level-2 64->64 conv (real config-2 rulebook, residual + second view) through the gather kernel and through the window kernel,
5 launches each -- run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / `--kernel-trace --stats` to compare their traffic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import _hip, ops
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_batch, make_tile
_hip.WIN_KERNEL = True
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
g = build_geometry(b["coords"].cuda(), b["batch_ids"].cuda(), 1, 0.1, 7, [500, 500, 1000])
L = _hip.lib()
lv = g.levels[1]
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
x = torch.randn((lv.n, 64), device="cuda", generator=gen).bfloat16(); res = torch.randn((lv.n, 64), device="cuda", generator=gen).bfloat16()
w = ops.pack_weight(torch.randn((64, 3, 3, 3, 64), device="cuda", generator=gen) / (27 * 64) ** 0.5, torch.bfloat16)
o2 = torch.empty((lv.n, 64), dtype=torch.bfloat16, device="cuda"); sc = torch.rand(64, device="cuda") + 0.5; sh = torch.randn(64, device="cuda")
for win in (0, 2):
    _hip.check(L.tl_set_tuning(b"win", win), "win"); _hip.check(L.tl_set_tuning(b"win_min_rows", 0), "wmr")
    for _ in range(5):
        ops.conv_fwd(x, w, lv.nbr, lv.n, residual=res, out2=(o2, sc, sh, True))
    torch.cuda.synchronize()
print("rows", lv.n, "algorithmic bytes (in + out + out2 + residual + 8 B/pair): %.1f MB" % ((lv.n * 64 * 2 * 4 + 8 * int((lv.nbr >= 0).sum())) / 1e6))
