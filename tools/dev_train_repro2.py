"""Dev: which gradient tensors of the config-3 step are not bit-reproducible?  Same weights, repeated forward + backward in one process;
optionally compares with / writes a reference file from another process.      python tools/dev_train_repro2.py [reps] [ref.pt]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ref_path = sys.argv[2] if len(sys.argv) > 2 else None
cfg = CONFIGS["config2"]
batch = make_batch([make_tile(**cfg, seed=s) for s in (0, 1)])
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=cfg["voxel"], compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
model = model.cuda().train()
gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
runs = []
for i in range(reps):
    model.zero_grad(set_to_none=True)
    loss, ld = model(gb, return_loss=True)
    loss.backward()
    torch.cuda.synchronize()
    runs.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    # BatchNorm running statistics move with every forward; they do not enter the gradients in training mode
bad = collections = {}
for i in range(1, reps):
    for n in runs[0]:
        if not torch.equal(runs[0][n], runs[i][n]):
            d = float((runs[0][n].double() - runs[i][n].double()).abs().max() / (runs[0][n].double().abs().max() + 1e-30))
            bad.setdefault(n, []).append((i, d))
print(f"{len(bad)} of {len(runs[0])} gradient tensors differ between repeats in this process")
for n, v in list(bad.items())[:40]:
    print("  ", n, tuple(runs[0][n].shape), v)
if ref_path:
    if os.path.exists(ref_path):
        ref = torch.load(ref_path)
        diff = [n for n in ref if not torch.equal(ref[n].cuda(), runs[0][n])]
        print(f"{len(diff)} of {len(ref)} gradient tensors differ from the other process's first run:", diff[:30])
    else:
        torch.save({n: g.cpu() for n, g in runs[0].items()}, ref_path)
        print("reference written")
