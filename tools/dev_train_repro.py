"""Dev: is the config-3 training step reproducible run to run?  Prints the losses of N steps and a checksum of all parameters after them;
run twice per mode (default; TL_WGRAD_JOIN=layer; TL_WGRAD_STREAM=0) and compare.      python tools/dev_train_repro.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = CONFIGS["config2"]
batch = make_batch([make_tile(**cfg, seed=s) for s in (0, 1)])
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=cfg["voxel"], compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
model = model.cuda().train()
opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=1e-3)
gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
out = []
for i in range(steps):
    opt.zero_grad()
    loss, ld = model(gb, return_loss=True)
    loss.backward()
    gsum = float(sum(p.grad.double().abs().sum() for p in model.parameters() if p.grad is not None))
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, norm_type=2)
    opt.step()
    out.append(f"{float(loss):.9g}/{gsum:.12g}")
chk = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
print(os.environ.get("TL_WGRAD_JOIN", "-"), os.environ.get("TL_WGRAD_STREAM", "-"), " ".join(out), f"params {chk:.14g}")
