"""Dev: where the HOST time of a config-3 training step goes (cProfile over 3 steps, top functions by own and cumulative time)."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict

cfg = CONFIGS["config2"]
batch = make_batch([make_tile(**cfg, seed=s) for s in (0, 1)])
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().train()
opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=1e-3)
g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}

def step():
    opt.zero_grad()
    loss, ld = model(g, return_loss=True)
    vals = [v.detach().cpu().item() for v in ld.values()]
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()

for _ in range(2): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(35); print(s.getvalue()[:9000])
