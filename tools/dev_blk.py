"""Developer timing of the block-local level-1 path on the config-2 tile: geometry build (canonical / blocked), the 32 -> 32 conv on the gather
kernel vs the staged-unit kernel, the whole forward with TL_BLK=0 / 1 (one tile at a time).      python tools/dev_blk.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import geometry as G, ops
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict


def timeit(f, reps=20, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
b = make_batch([make_tile(**CONFIGS[cfg], seed=0)])
vs = CONFIGS[cfg]["voxel"]
ss = [500, 500, 1000] if vs >= 0.1 else None
c, bi = b["coords"].cuda().float(), b["batch_ids"].cuda().long()
for blocked in (False, True):
    print(f"build_geometry blocked={blocked}: {timeit(lambda: G.build_geometry(c, bi, 1, vs, 7, ss, blocked=blocked), 10):.3f} ms", flush=True)
can = G.build_geometry(c, bi, 1, vs, 7, ss)
blk = G.build_geometry(c, bi, 1, vs, 7, ss, blocked=True)
r = blk.levels[0].nbr
n = can.levels[0].n
nu = int(r.counter[0]); u = r.unit[:nu].long()
print(f"rows {n}, units {nu} (chunks {(n + 63) // 64}), own/unit {float(u[:, 1].float().mean()):.1f}, halo/unit mean {float(u[:, 2].float().mean()):.1f} max {int(u[:, 2].max())}, "
      f"staged rows per output row {float((u[:, 1] + u[:, 2]).sum()) / n:.2f}", flush=True)
torch.manual_seed(0)
x = torch.randn(n, 32, device="cuda").bfloat16(); res = torch.randn(n, 32, device="cuda").bfloat16()
w = ops.pack_weight(torch.randn(32, 3, 3, 3, 32, device="cuda") * 0.1, torch.bfloat16)
sc, sh = torch.rand(32, device="cuda") + 0.5, torch.randn(32, device="cuda")
o1, o2 = torch.empty_like(x), torch.empty_like(x)
for name, kw in (("plain", {}), ("residual", dict(residual=res)), ("residual + second view", dict(residual=res, out2=(o2, sc, sh, True))),
                 ("act view only", dict(out_scale=sc, out_shift=sh, out_relu=True))):
    ta = timeit(lambda: ops.conv_fwd(x, w, can.levels[0].nbr, n, out=o1, **kw))
    tb = timeit(lambda: ops.conv_fwd(x, w, r, n, out=o1, **kw))
    print(f"32->32 {name}: gather kernel {ta:.3f} ms | staged-unit kernel {tb:.3f} ms", flush=True)

m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=ss, voxel_size=vs, compute_dtype=torch.bfloat16)
m.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
m = m.cuda().eval()
m.return_backbone_feats = False
dev_b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
for flag in ("0", "1", "0", "1"):
    os.environ["TL_BLK"] = flag
    with torch.no_grad():
        t = timeit(lambda: m(dev_b, return_loss=False), 20)
    print(f"forward, one tile at a time, TL_BLK={flag}: {t:.3f} ms", flush=True)
