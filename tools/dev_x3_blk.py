"""Dev: the staged-unit kernel of the parity-fast mode (tl_conv_blk_x3.hip) on the config-2 level-1 rulebook -- per conv form, against the
gather kernel it replaces (canonical table, per-tap split) and for several chunk counts of its two-launch scheme (tl_set_tuning "x3_chunks":
launch A writes the first 16 channels' sums, launch B adds them; a chunk's A and B run back to back so that B finds the sums in the
memory-side cache).  Also the level-1 inverse conv (one-hot) in its gather-once form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import _hip, geometry as G, ops
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

L = _hip.lib()
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
c, bi = b["coords"].cuda().float(), b["batch_ids"].cuda().long()
can = G.build_geometry(c, bi, 1, 0.1, 7, [500, 500, 1000])
blk = G.build_geometry(c, bi, 1, 0.1, 7, [500, 500, 1000], blocked=True)
n = can.levels[0].n
torch.manual_seed(0)
x = torch.randn(n, 32, device="cuda"); res = torch.randn(n, 32, device="cuda")
ops.PACK_X3 = True
w = ops.pack_weight(torch.randn(32, 3, 3, 3, 32, device="cuda") * 0.1, torch.float32)
wu = ops.pack_weight(torch.randn(32, 2, 2, 2, 64, device="cuda") * 0.1, torch.float32)
ops.PACK_X3 = False
sc = torch.rand(32, device="cuda") + 0.5; sh = torch.randn(32, device="cuda") * 0.1
o1 = torch.empty_like(x); o2 = torch.empty_like(x)


def timeit(f, reps=20, warm=4):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


forms = (("plain", dict()), ("prologue", dict(in_scale=sc, in_shift=sh, in_relu=True)), ("residual", dict(residual=res)),
         ("residual + activated second view", dict(residual=res, out2=(o2, sc, sh, True))))
print(f"config-2 level 1: {n} rows, fp32 rows, split-bf16 contraction")
for name, kw in forms:
    if "in_scale" not in kw:
        t = timeit(lambda: ops.conv_fwd(x, w, can.levels[0].nbr, n, out=o1, **kw))
        print(f"gather kernel (canonical table)            {name:36s} {t:.4f} ms")
for ch in (0, 1, 2, 3, 4, 6):
    _hip.check(L.tl_set_tuning(b"x3_chunks", ch), "x3_chunks")
    for name, kw in forms:
        t = timeit(lambda: ops.conv_fwd(x, w, blk.levels[0].nbr, n, out=o1, **kw))
        print(f"staged kernel, x3_chunks {ch} (0 = one chunk)   {name:36s} {t:.4f} ms", flush=True)
_hip.check(L.tl_set_tuning(b"x3_chunks", 2), "x3_chunks")
lv = can.levels[0]; n2 = can.levels[1].n
e = torch.randn(n2, 64, device="cuda")
t = timeit(lambda: ops.conv_fwd(e, wu, lv.inv, n, out=o1, one_hot=True))
print(f"level-1 inverse conv 64 -> 32 (one-hot, fp32 rows, split-bf16): {t:.4f} ms")
