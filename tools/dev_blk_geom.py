"""Developer: the blocked geometry build of the config-2 tile in a loop (for rocprofv3 --pmc runs on the tl_blk kernels).  python tools/dev_blk_geom.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import geometry as G
from treelearn_amd.synth import CONFIGS, make_batch, make_tile
b = make_batch([make_tile(**CONFIGS["config2"], seed=0)])
c, bi = b["coords"].cuda().float(), b["batch_ids"].cuda().long()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    g = G.build_geometry(c, bi, 1, 0.1, 7, [500, 500, 1000], blocked=True)
torch.cuda.synchronize()
print("units", int(g.levels[0].nbr.counter[0]))
