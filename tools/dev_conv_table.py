"""Every conv launch of ONE eval forward of the config-2 tile, in order, with its own duration (ops.PROFILE events; one tile at a time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import ops
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict
gb = make_batch([make_tile(**CONFIGS["config2"], seed=0)]); gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in gb.items()}
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
REP = 8
acc = None
with torch.no_grad():
    for _ in range(3): model(gb, return_loss=False)
    for _ in range(REP):
        ops.PROFILE = []
        model(gb, return_loss=False); torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b, _ in ops.PROFILE]
        acc = ms if acc is None else [x + y for x, y in zip(acc, ms)]
        metas = [m for _, _, m in ops.PROFILE]
    ops.PROFILE = None
tot = 0.0
for t, m in zip(acc, metas):
    t /= REP; tot += t
    print("K=%2d %3d->%3d rows %8d  %-16s res=%d pro=%d split=%s  %.3f ms" % (m["K"], m["Cin"], m["Cout"], m["n_out"], type(m["table"]).__name__, m["residual"], m.get("in_scale", 0), m["split"], t))
print("launches %d, sum %.3f ms" % (len(metas), tot))
