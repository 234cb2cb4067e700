"""Dev: do two tiles in flight on two streams (whole forwards, ping-pong) beat one after the other?  Same tile, same model; the
deep levels (36 small launches) and the geometry leave most CUs idle, which the other tile's big convs could fill."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd.synth import random_state_dict
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile
cfg = CONFIGS["config2"]
gs = []
for seed in range(2):
    b = make_batch([make_tile(**cfg, seed=seed)])
    gs.append({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()})
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=cfg["voxel"], compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
NS = [int(a) for a in sys.argv[1:]] or [2, 3, 4]
st = [torch.cuda.Stream() for _ in range(max(NS))]
K = 48
with torch.no_grad():
    for _ in range(4): model(gs[0], return_loss=False)
    for rnd in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K): model(gs[i & 1], return_loss=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        msg = f"one stream: {(t1 - t0) / K * 1e3:.3f} ms/tile"
        for ns in NS:
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for i in range(K):
                with torch.cuda.stream(st[i % ns]):
                    model(gs[i & 1], return_loss=False)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            msg += f"   {ns} streams: {(t2 - t1) / K * 1e3:.3f}"
        print(msg, flush=True)
