"""Dev: is the config-3 training step host-bound?  Times, per phase, how long Python takes to ENQUEUE the work (no sync) against the
time until the device has finished it.  enqueue ~= done means the launch thread is the limit in that phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict

cfg = CONFIGS["config2"]
batch = make_batch([make_tile(**cfg, seed=s) for s in (0, 1)])
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().train()
fused = os.environ.get("TL_BENCH_ADAMW", "fused") == "fused"              # as bench.py: torch's single-kernel AdamW unless TL_BENCH_ADAMW=foreach
opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=1e-3, fused=fused)
print("AdamW:", "fused=True" if fused else "for-each")
g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
sync = torch.cuda.synchronize
for it in range(5):
    sync(); t0 = time.perf_counter()
    opt.zero_grad()
    loss, ld = model(g, return_loss=True)
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter(); sync(); t4 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    t5 = time.perf_counter(); sync(); t6 = time.perf_counter()
    print(f"step {it}: forward enqueue {1e3 * (t1 - t0):.1f} ms, done {1e3 * (t2 - t0):.1f} | backward enqueue {1e3 * (t3 - t2):.1f}, done {1e3 * (t4 - t2):.1f} | "
          f"clip+AdamW enqueue {1e3 * (t5 - t4):.1f}, done {1e3 * (t6 - t4):.1f} | total {1e3 * (t6 - t0):.1f} ms", flush=True)
