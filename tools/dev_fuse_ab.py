"""Dev: one training step with the conv-epilogue BatchNorm fusion on / off (autograd.FUSE_BN): per-parameter cosine and norm ratio."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from treelearn_amd import autograd as ag
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import make_batch, make_tile, random_state_dict
cfg = dict(channels=32, num_blocks=4)
batch = make_batch([make_tile(extent=14.0, voxel=0.1, n_trees=8, fill=0.10, seed=s) for s in (3, 4)])
gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
ref32 = None
for dtype in (torch.float32, torch.bfloat16):
    res = {}
    for tag, fuse in (("fused", True), ("separate", False), ("separate2", False)):
        ag.FUSE_BN = fuse
        model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, compute_dtype=dtype, **cfg)
        model.load_state_dict(random_state_dict(5, **cfg), strict=True); model = model.cuda().train()
        loss, _ = model(gb, return_loss=True); loss.backward()
        res[tag] = (float(loss), {n: p.grad.detach().float().cpu().numpy().ravel() for n, p in model.named_parameters()})
    ag.FUSE_BN = True
    print(dtype, "loss", res["fused"][0], res["separate"][0], res["separate2"][0])
    for other in ("separate2", "fused"):
        rows = []
        for n, b in res["separate"][1].items():
            a = res[other][1][n]
            nb = np.linalg.norm(b)
            if nb > 0:
                rows.append((1 - float(a @ b / (np.linalg.norm(a) * nb + 1e-30)), abs(np.linalg.norm(a) / nb - 1), n))
        rows.sort(reverse=True)
        print(f"  {other} vs separate: worst 1-cos / |norm ratio - 1|")
        for r in rows[:12]:
            print(f"    {r[0]:.2e} {r[1]:.2e} {r[2]}")

    if dtype == torch.float32:
        ref32 = res["separate"][1]
    else:
        rows = []
        for n, r in ref32.items():
            nr = np.linalg.norm(r)
            if nr > 0:
                cf = 1 - float(res["fused"][1][n] @ r / (np.linalg.norm(res["fused"][1][n]) * nr + 1e-30))
                cs = 1 - float(res["separate"][1][n] @ r / (np.linalg.norm(res["separate"][1][n]) * nr + 1e-30))
                rows.append((max(cf, cs), cf, cs, n))
        rows.sort(reverse=True)
        print("  bf16 vs the fp32 gradients: 1-cos of fused / separate")
        for r in rows[:25]:
            print(f"    {r[1]:.2e} {r[2]:.2e} {r[3]}")
        print("  mean 1-cos: fused %.3e separate %.3e" % (np.mean([r[1] for r in rows]), np.mean([r[2] for r in rows])))
