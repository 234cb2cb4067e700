import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from treelearn_amd.synth import random_state_dict
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import make_batch, make_tile
t = make_tile(extent=40, voxel=0.1, n_trees=64, fill=0.10, seed=0)
b = make_batch([t], inner_square_edge_length=8.0)
b = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()}
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
def T(): torch.cuda.synchronize(); return time.time()
with torch.no_grad():
    for rep in range(3):
        t0 = T()
        g = {k: (v.cuda(non_blocking=True) if torch.is_tensor(v) and k in ("coords", "input_feats", "batch_ids", "masks_inner") else v) for k, v in b.items()}
        t1 = T()
        out = model(g, return_loss=False)
        t2 = T()
        idx = torch.nonzero(g["masks_inner"]).squeeze(1)
        off = out["offset_predictions"].index_select(0, idx).cpu(); sem = out["semantic_prediction_logits"].index_select(0, idx).cpu(); bb = out["backbone_feats"].index_select(0, idx).cpu()
        t3 = T()
        ci = idx.cpu()
        x = [b[k].index_select(0, ci) for k in ("semantic_labels", "offset_labels", "coords", "centers", "instance_labels", "input_feats")]
        t4 = T()
        print(f"h2d {1e3*(t1-t0):.2f}  fwd {1e3*(t2-t1):.2f}  select+d2h {1e3*(t3-t2):.2f}  cpu select {1e3*(t4-t3):.2f} ms")
