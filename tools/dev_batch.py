"""Throughput when B tiles form ONE batch per forward (batch ids 0..B-1; eval-mode BatchNorm: tiles do not interact) instead of B
forwards on B streams."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict
tiles = [make_tile(**CONFIGS["config2"], seed=s) for s in range(4)]
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval(); model.return_backbone_feats = True
def bench(B, nstreams, steps=24):
    gb = make_batch(tiles[:B]); gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in gb.items()}
    n = gb["coords"].shape[0]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    def run(k):
        cur = torch.cuda.current_stream()
        for st in streams: st.wait_stream(cur)
        for i in range(k):
            with torch.cuda.stream(streams[i % nstreams]), torch.no_grad():
                model(gb, return_loss=False)
        for st in streams: cur.wait_stream(st)
    with torch.no_grad(): model(gb, return_loss=False)
    run(4); torch.cuda.synchronize(); t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"tiles per batch {B}, streams {nstreams}: {dt / steps / B * 1e3:.3f} ms per tile = {n * steps / dt / 1e6:.1f} Mpoints/s", flush=True)
for B, ns in ((1, 1), (1, 3), (2, 1), (2, 2), (3, 1), (3, 2), (4, 1), (4, 2)):
    bench(B, ns)
