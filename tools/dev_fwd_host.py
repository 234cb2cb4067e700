"""Dev: where does the wall time of ONE eval forward go on the host?  Per phase, the time Python takes to ENQUEUE the work (no sync)
against the time until the device has finished it (SURVEY 8d latency = input resident -> outputs resident, one tile at a time).
Phases: geometry up to host sync #1 (grid extent), up to host sync #2 (level counts), the rest of the geometry launches, the conv
launches level by level (marks from ops.conv_fwd call counts) and the head.  `enqueue ~= done` in a phase = the launch thread is
the limit there.  TL_EXEC=0 / 1 selects the Python-driven engine / the C-side executor (tl_forward) when the library has it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import geometry as G, ops
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict

cfg = CONFIGS["config2"]
batch = make_batch([make_tile(**cfg, seed=0)])
model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
model.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
sync = torch.cuda.synchronize
pc = time.perf_counter

with torch.no_grad():
    for _ in range(4):
        model(g, return_loss=False)
    sync()
    # (a) whole forward: enqueue vs done
    rows = []
    for it in range(12):
        sync(); t0 = pc()
        out = model(g, return_loss=False)
        t1 = pc(); sync(); t2 = pc()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    rows.sort(key=lambda r: r[1])
    med = rows[len(rows) // 2]
    print(f"whole forward: enqueue returned after {med[0]:.2f} ms, done after {med[1]:.2f} ms (median of 12; min done {rows[0][1]:.2f}, max {rows[-1][1]:.2f})")

    # (b) the two halves the model runs: geometry (with its host syncs) and the network
    if hasattr(G, "TRACE"):
        acc = {}
        for it in range(12):
            sync(); G.TRACE = []; t0 = pc()
            vf, geom = model._voxelize(g["coords"].float(), g["input_feats"].float(), g["batch_ids"].long(), 1, blocked=model._plan.supports_blocked())
            t1 = pc(); sync(); t2 = pc()
            tr = G.TRACE; G.TRACE = None
            bb, lo, of = model._plan.run(vf, geom, want_backbone=True, all_ones=True)
            t3 = pc(); sync(); t4 = pc()
            prev = t0
            for name, t in tr:
                acc.setdefault("geom: " + name, []).append(1e3 * (t - prev)); prev = t
            acc.setdefault("geom: enqueue total", []).append(1e3 * (t1 - t0)); acc.setdefault("geom: done", []).append(1e3 * (t2 - t0))
            acc.setdefault("net: enqueue", []).append(1e3 * (t3 - t2)); acc.setdefault("net: done", []).append(1e3 * (t4 - t2))
        for k, v in acc.items():
            v.sort(); print(f"  {k:46s} {v[len(v) // 2]:7.3f} ms (min {v[0]:.3f})")
    # (c) host cost of a bare conv launch call (Python wrapper + ctypes + hipLaunch), no device work to wait for
    x = torch.zeros((64, 32), dtype=torch.bfloat16, device="cuda"); w = torch.zeros((1, 32, 32), dtype=torch.bfloat16, device="cuda")
    sync(); t0 = pc()
    for _ in range(2000):
        ops.conv_fwd(x, w, None, 64, out=x)
    t1 = pc(); sync()
    print(f"host cost of one ops.conv_fwd call (tiny launch): {1e6 * (t1 - t0) / 2000:.1f} us")
    sync(); t0 = pc()
    for _ in range(2000):
        torch.empty((1000, 32), dtype=torch.bfloat16, device="cuda")
    t1 = pc()
    print(f"host cost of one torch.empty: {1e6 * (t1 - t0) / 2000:.1f} us")
