"""Dev: config-3 training FORWARD only -- with the packed-weight cache warm (no pack launches) and cold (every parameter's version bumped,
as after an optimizer step): what the ~210 weight-packing launches per step cost in wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict
cfg = CONFIGS["config2"]
batch = make_batch([make_tile(**cfg, seed=s) for s in (0, 1)])
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().train()
g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
def fwd():
    loss, _ = model(g, return_loss=True)
    return loss
for _ in range(3): fwd()
torch.cuda.synchronize()
for mode in ("warm", "cold", "warm", "cold"):
    ts = []
    for _ in range(6):
        if mode == "cold":
            with torch.no_grad():
                for p in model.parameters(): p.add_(0.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fwd(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{mode}: forward {1e3 * min(ts):.2f} ms (min of 6), median {1e3 * sorted(ts)[3]:.2f}")
