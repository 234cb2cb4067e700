"""Run the sharded tile loop over RCCL (torchrun, one rank per visible GPU; works with a single GPU) and compare it with
the plain single-process loop:   python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/shard_demo.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from treelearn_amd.synth import random_state_dict
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import make_batch, make_tile
from treelearn_amd.util.pipeline import get_pointwise_preds
from treelearn_amd.util.sharding import get_pointwise_preds_sharded

local = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
tiles = []
for s in range(5):
    t = make_tile(extent=10.0 + 2 * s, voxel=0.1, n_trees=3 + s, fill=0.10, seed=s)
    t["center"] = np.array([8.0 * s, 0.0, 0.0])
    tiles.append(make_batch([t], inner_square_edge_length=6.0))
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
res = get_pointwise_preds_sharded(model, tiles, dict(voxel_size=0.1))
if dist.get_rank() == 0:
    ref = get_pointwise_preds(model, tiles, dict(voxel_size=0.1))
    ok = all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(res, ref))
    print(f"world {dist.get_world_size()}: sharded == single-process loop: {ok}; rows {res[0].shape[0]}", flush=True)
dist.barrier()
dist.destroy_process_group()
