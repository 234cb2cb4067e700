"""Dev: one pipelined pass of get_pointwise_preds over six device-resident 40 m tiles, for rocprofv3 timelines:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_loop -o p -- python3 tools/loop_prof.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from treelearn_amd.synth import random_state_dict
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import make_batch, make_tile
from treelearn_amd.util import get_pointwise_preds
from bench import host_cores
torch.set_num_threads(host_cores())
gt = []
for s in range(6):
    t = make_tile(extent=40.0, voxel=0.1, n_trees=64, fill=0.10, seed=s)
    t["center"] = np.array([8.0 * s, 0.0, 0.0])
    b = make_batch([t], inner_square_edge_length=8.0)
    gt.append({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()})
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
get_pointwise_preds(model, gt, dict(voxel_size=0.1))
torch.cuda.synchronize(); t0 = time.time()
get_pointwise_preds(model, gt, dict(voxel_size=0.1))
torch.cuda.synchronize(); print(f"tile loop: {(time.time() - t0) * 1e3 / len(gt):.2f} ms/tile")
