"""Dev: which ATen kernels remain in one config-3 training step (torch.profiler on the box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import Counter
from torch.profiler import profile, ProfilerActivity
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict
b = make_batch([make_tile(**CONFIGS["config2"], seed=s) for s in (0, 1)])
g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().train()
opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=1e-3)
def step():
    opt.zero_grad(); loss, _ = model(g, return_loss=True); loss.backward(); torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
tot = sum(e.device_time for e in ev)
c = Counter(); t = Counter()
for e in ev:
    if "at::" in e.name or "Cat" in e.name or "indexing" in e.name or "rocprim" in e.name or "Memcpy" in e.name or "Memset" in e.name:
        k = e.name[:110]; c[k] += 1; t[k] += e.device_time
print(f"device time of the step {tot / 1e3:.1f} ms, {len(ev)} device events; ATen / copies: {sum(t.values()) / 1e3:.2f} ms")
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{c[k]:4d} {v / 1e3:7.2f} ms  {k}")
# which Python lines the big ATen ops come from
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof2:
    step(); torch.cuda.synchronize()
rows = [r for r in prof2.key_averages(group_by_stack_n=8) if r.device_time_total > 300 and r.key.startswith("aten::")]
for r in sorted(rows, key=lambda r: -r.device_time_total)[:14]:
    st = [s_ for s_ in r.stack if ("treelearn_amd" in s_ or "bench" in s_ or "dev_train" in s_)][:3]
    print(f"{r.count:4d} {r.key:26s} {r.device_time_total / 1e3:7.2f} ms  {' <- '.join(x.split('/')[-1][:60] for x in st)}")
# the big casts / copies by input shape
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof3:
    step(); torch.cuda.synchronize()
rows = [r for r in prof3.key_averages(group_by_input_shape=True) if r.device_time_total > 150 and r.key in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::sum", "aten::index", "aten::_index_put_impl_", "aten::add", "aten::add_", "aten::mul")]
for r in sorted(rows, key=lambda r: -r.device_time_total)[:18]:
    print(f"{r.count:4d} {r.key:24s} {r.device_time_total / 1e3:7.2f} ms  {str(r.input_shapes)[:110]}")
