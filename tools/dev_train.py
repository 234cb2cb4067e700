"""Dev: time one training step (fwd + bwd, batch of 2 crops) -- BASELINE config 3."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd.synth import random_state_dict
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import make_batch, make_tile
E = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
tiles = [make_tile(extent=E, voxel=0.1, n_trees=int(64 * (E / 40) ** 2), fill=0.10, seed=s) for s in (0, 1)]
batch = make_batch(tiles)
dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32          # bf16 = autocast-like mixed precision
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=dt)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().train()
opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=1e-3)
g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
print("points", g["coords"].shape[0], flush=True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    opt.zero_grad()
    loss, ld = model(g, return_loss=True)
    torch.cuda.synchronize(); t1 = time.time()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    torch.cuda.synchronize(); t2 = time.time()
    print(f"step {it}: fwd {t1-t0:.3f}s bwd+opt {t2-t1:.3f}s loss {float(loss):.4f} mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
