"""Which host-side ops of ONE eval forward launch something besides the library's kernels (torch.profiler, grouped by source line)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict
gb = make_batch([make_tile(**CONFIGS["config2"], seed=0)]); gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in gb.items()}
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval(); model.return_backbone_feats = True
with torch.no_grad():
    for _ in range(3): model(gb, return_loss=False)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model(gb, return_loss=False); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print("device events in one forward:", len(ev))
from collections import Counter
c = Counter(e.name[:70] for e in ev)
for k, v in c.most_common(40): print(f"{v:4d}  {k}")
print("--- aten ops with a device kernel, by python source line")
tab = prof.key_averages(group_by_stack_n=6)
rows = [r for r in tab if r.device_time_total > 0 and r.key.startswith("aten::")]
for r in sorted(rows, key=lambda r: -r.count)[:40]:
    st = [s for s in r.stack if "treelearn_amd" in s or "bench" in s][:2]
    print(f"{r.count:4d} {r.key:28s} {r.device_time_total:8.0f} us  {' <- '.join(s.split('/')[-1] for s in st)}")
