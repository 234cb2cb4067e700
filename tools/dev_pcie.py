"""Dev: the PCIe-inclusive rate of the tile path (the reference's tile loop hands `forward` HOST tensors -- DataLoader batches,
pin_memory=True, tree_learn/util/train.py:140 -- and copies the per-point outputs back, util/pipeline.py:88-90).  bench.py's `value` is quoted
with the inputs resident in HBM; this prints the same forward fed from pinned host memory, one tile at a time, and the production tile loop
(H2D of the next tile on a copy stream, inner-square filter on the device, ONE packed D2H copy per tile) over eight host-resident tiles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict
from treelearn_amd.util import get_pointwise_preds
from bench import host_cores
torch.set_num_threads(host_cores())          # (the box shows 256 cores and grants 16: torch's default pool would oversubscribe the host-side row selections)

tiles = []
for s in range(8):
    t = make_tile(**CONFIGS["config2"], seed=s % 4)
    t["center"] = np.array([8.0 * s, 0.0, 0.0])
    b = make_batch([t], inner_square_edge_length=8.0)
    tiles.append({k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()})
npts = sum(b["coords"].shape[0] for b in tiles) / len(tiles)
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
dev = [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()} for b in tiles[:2]]
sync = torch.cuda.synchronize
with torch.no_grad():
    for _ in range(3): model(dev[0], return_loss=False); model(tiles[0], return_loss=False)
    sync()
    for name, src in (("device-resident input", dev[0]), ("pinned host input (H2D of coords + batch ids inside the call)", tiles[0])):
        lat = []
        for _ in range(16):
            t0 = time.perf_counter(); out = model(src, return_loss=False); sync(); lat.append(time.perf_counter() - t0)
        lat.sort(); m = 0.5 * (lat[7] + lat[8])
        print(f"one forward, {name}: {1e3 * m:.2f} ms = {npts / m / 1e6:.1f} Mpoints/s")
    t0 = time.perf_counter(); out = model(tiles[0], return_loss=False); host = [v.cpu() for v in out.values()]; sync()
    print(f"one forward from pinned host memory + the reference's full-tile D2H of all three outputs (util/pipeline.py:90, 277 MB): {1e3 * (time.perf_counter() - t0):.2f} ms")
get_pointwise_preds(model, tiles, dict(voxel_size=0.1))
for rep in range(3):
    sync(); t0 = time.perf_counter()
    res = get_pointwise_preds(model, tiles, dict(voxel_size=0.1))
    sync(); dt = time.perf_counter() - t0
    print(f"tile loop over {len(tiles)} HOST-resident tiles (H2D prefetch, device-side inner filter, packed D2H of {len(res[0])} inner rows): "
          f"{1e3 * dt / len(tiles):.2f} ms per tile = {npts * len(tiles) / dt / 1e6:.1f} Mpoints/s")
