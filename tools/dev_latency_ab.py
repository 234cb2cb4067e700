"""Same-box A/B of a lone forward's latency under an environment switch: tools/dev_latency_ab.py VAR [a b]  (alternates child processes)."""
import os, subprocess, sys
CHILD = r'''
import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
import torch
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
m.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); m = m.cuda().eval()
ts = []
with torch.no_grad():
    for i in range(45):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m(gb, return_loss=False); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
ts = ts[5:]
print("%.3f median  %.3f min" % (statistics.median(ts), min(ts)))
'''
var = sys.argv[1]; vals = sys.argv[2:4] if len(sys.argv) > 3 else ["0", "1"]
for rep in range(3):
    for v in vals:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **{var: v}), capture_output=True, text=True)
        print(f"{var}={v}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
