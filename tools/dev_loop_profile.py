import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import host_cores
torch.set_num_threads(host_cores())
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, tile_variant, random_state_dict
from treelearn_amd.util import get_pointwise_preds
base = [make_tile(**CONFIGS["config2"], seed=100 + s) for s in range(2)]
tiles = []
for i in range(16):
    b = make_batch([tile_variant(base[i % 2], (i // 2) % 8)], inner_square_edge_length=8.0)
    tiles.append({k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()})
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
get_pointwise_preds(model, tiles, dict(voxel_size=0.1))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    get_pointwise_preds(model, tiles, dict(voxel_size=0.1))
    torch.cuda.synchronize(); print("ms per tile", 1e3 * (time.perf_counter() - t0) / 16)
pr = cProfile.Profile(); pr.enable()
get_pointwise_preds(model, tiles, dict(voxel_size=0.1)); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
dev = [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()} for b in tiles]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    get_pointwise_preds(model, dev, dict(voxel_size=0.1), keep_on_device=True)
    torch.cuda.synchronize(); print("device-resident, keep_on_device: ms per tile", 1e3 * (time.perf_counter() - t0) / 16)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    get_pointwise_preds(model, dev, dict(voxel_size=0.1))
    torch.cuda.synchronize(); print("device-resident, numpy out: ms per tile", 1e3 * (time.perf_counter() - t0) / 16)


def prof(label, **kw):
    pr = cProfile.Profile(); torch.cuda.synchronize(); t0 = time.perf_counter(); pr.enable()
    get_pointwise_preds(model, dev, dict(voxel_size=0.1), **kw); torch.cuda.synchronize()
    pr.disable(); dt = time.perf_counter() - t0
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(9)
    print(label, "wall ms", 1e3 * dt); print("\n".join(s.getvalue().splitlines()[6:18]))


prof("device tiles, keep_on_device", keep_on_device=True)
prof("device tiles, numpy out")
prof("device tiles, numpy out, no backbone columns", return_backbone_feats=False)
print("transparent huge pages:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
# the PCIe link itself: one 12.5 MB block device -> pinned host, and pinned host -> pageable (what the final cat does)
x = torch.empty(80000, 39, device="cuda"); h = torch.empty(x.shape, pin_memory=True)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); h.copy_(x, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("D2H pinned 12.5 MB: %.3f ms = %.1f GB/s" % (1e3 * dt, x.numel() * 4 / dt / 1e9))
for _ in range(3):
    t0 = time.perf_counter(); y = x.cpu(); dt = time.perf_counter() - t0
    print("D2H pageable 12.5 MB: %.3f ms = %.1f GB/s" % (1e3 * dt, x.numel() * 4 / dt / 1e9))
hs = [torch.empty(x.shape, pin_memory=True) for _ in range(16)]
for _ in range(3):
    t0 = time.perf_counter(); y = torch.cat(hs, 0); dt = time.perf_counter() - t0
    print("cat of 16 pinned blocks into fresh pageable memory: %.3f ms = %.1f GB/s" % (1e3 * dt, y.numel() * 4 / dt / 1e9)); del y
dev64 = [dev[i % 16] for i in range(64)]
for kw in (dict(keep_on_device=True), dict(), dict(return_backbone_feats=False)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    get_pointwise_preds(model, dev64, dict(voxel_size=0.1), **kw)
    torch.cuda.synchronize(); print("64 device tiles", kw, "ms per tile", 1e3 * (time.perf_counter() - t0) / 64)
import gc
for kw in (dict(), dict(keep_on_device=True)):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = get_pointwise_preds(model, dev, dict(voxel_size=0.1), **kw); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        del res; t3 = time.perf_counter(); gc.collect(); t4 = time.perf_counter()
        print(kw, "call %.1f ms, sync after %.1f, del result %.1f, gc %.1f" % tuple(1e3 * d for d in (t1 - t0, t2 - t1, t3 - t2, t4 - t3)))
