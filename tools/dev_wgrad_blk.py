"""The staged-unit weight gradient of level 1 against the dense-over-taps kernel on the config-3 rulebook (two 40 m crops): time per launch."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from treelearn_amd import ops
from treelearn_amd.geometry import build_geometry
from treelearn_amd.synth import CONFIGS, make_batch, make_tile
b = make_batch([make_tile(**CONFIGS["config2"], seed=s) for s in (0, 1)])
geom = build_geometry(b["coords"].cuda(), b["batch_ids"].cuda(), 2, 0.1, 7, [500, 500, 1000], blocked=True, nn_table=True)
r = geom.levels[0].nbr
n = r.n
print("rows", n, "units", int(r.counter[0]))
for dt in (torch.bfloat16, torch.float16):
    for ci in (32, 64):
        x = (torch.randn(n, ci, device="cuda") * 0.7).to(dt); g = (torch.randn(n, 32, device="cuda") * 0.3).to(dt)
        for flag in (True, False):
            ops.WGRAD_BLK = flag
            for _ in range(3): out = ops.conv_wgrad(x, g, r, n, 27, ref_layout=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): out = ops.conv_wgrad(x, g, r, n, 27, ref_layout=True)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
            print(f"{dt} {ci}->32 {'staged units' if flag else 'dense over taps'}: {ms:.3f} ms", flush=True)
            if flag: a = out
            else: print("   max |difference| / max |value|:", float((a - out).abs().max() / out.abs().max()))
