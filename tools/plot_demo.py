"""End-to-end plot pipeline on the device (every SURVEY 8 row chained; mirrors tools/pipeline/pipeline.py:40-94 of the
reference minus file I/O):  raw cloud -> voxel down-sample + verticality -> overlapping tiles -> tile loop -> ensemble ->
grouping -> k-NN fill -> back to the raw points.  Synthetic plot, random-init weights (timing / plumbing demo).

    python tools/plot_demo.py [plot_edge_m=60] [inner=8] [outer=13.5] [stride=0.5]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from treelearn_amd.synth import random_state_dict
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import make_tile
from treelearn_amd.util import get_instances, get_pointwise_preds
from treelearn_amd.util.postprocess import assign_remaining_points_nearest_neighbor, ensemble
from treelearn_amd.util.prepare import compute_features, propagate_to_original, voxelize
from treelearn_amd.util.tiles import PlotTiler
from bench import host_cores
torch.set_num_threads(host_cores())

E = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
inner, outer, stride = (float(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 8.0), (3, 13.5), (4, 0.5)))
t = make_tile(extent=E, voxel=0.1, n_trees=int(64 * (E / 40) ** 2), fill=0.10, seed=9)
rng = np.random.default_rng(0)
raw = np.vstack([t["points"].astype(np.float64) + rng.normal(0, 0.02, size=t["points"].shape) for _ in range(2)])
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
T = {}
def lap(name, t0):
    torch.cuda.synchronize(); T[name] = time.time() - t0; return time.time()
d_raw = torch.from_numpy(raw).cuda(); torch.cuda.synchronize(); t0 = time.time()
down, trace = voxelize(d_raw, 0.1); t0 = lap("voxel down-sample", t0)
feats = compute_features(down[:, :3], 0.6); t0 = lap("verticality", t0)
tiler = PlotTiler(down[:, :3].float(), torch.full((len(down),), -1.0), feats)
ntiles = [0]
def tiles():
    for b in tiler.tiles(inner, outer, stride, inner, offset_labels="none"):
        ntiles[0] += 1; yield b
res = get_pointwise_preds(model, tiles(), dict(voxel_size=0.1), keep_on_device=True); t0 = lap("tiling + tile loop", t0)
ens = ensemble(res[4], res[0], res[1], res[2], res[3], res[5], res[6], res[7], return_device=True); t0 = lap("ensemble", t0)
coords, sem, off, infeat = (ens[i].cpu().numpy() for i in (0, 1, 3, 7)); t0 = lap("D2H of what grouping reads", t0)
cfg = dict(tree_conf_thresh=0.5, tau_vert=0.0, tau_off=1e9, tau_group=0.3, tau_min=20, use_hdbscan=False)
inst = get_instances(coords, off, sem, cfg, infeat[:, -1], 0, 0, -1, 1); t0 = lap("grouping (DBSCAN)", t0)
cfg_h = dict(cfg, use_hdbscan=True, tau_min=50)
inst_h = get_instances(coords, off, sem, cfg_h, infeat[:, -1], 0, 0, -1, 1); t0 = lap("grouping (HDBSCAN, extra)", t0)
print(f"HDBSCAN grouping over {int((inst_h != 0).sum())} points: {int(inst_h.max())} instances, {int((inst_h == -1).sum())} unassigned")
tree = inst != 0
if tree.any() and (inst[tree] != -1).any():
    inst[tree] = assign_remaining_points_nearest_neighbor(coords[tree] + off[tree], inst[tree], -1)
t0 = lap("k-NN fill", t0)
print(f"plot {E:.0f} x {E:.0f} m: {len(raw)} raw points -> {len(down)} voxels, {ntiles[0]} tiles, {len(res[0])} inner predictions -> {len(coords)} ensembled points")
for k, v in T.items(): print(f"  {k:22s} {v * 1e3:9.1f} ms")
print(f"  tile loop per tile      {T['tiling + tile loop'] * 1e3 / max(ntiles[0], 1):9.2f} ms   total {sum(T.values()):.2f} s")
