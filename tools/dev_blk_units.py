"""Dev (developer build: python -m treelearn_amd.build --dev): where does k_blk_units spend its time?  The kernel alone on the config-2 tile
(and on the two crops of a training batch), stopped after each of its stages (tl_dev_blk_abl; the outputs of the ablated runs are incomplete
on purpose): 1 = neighbour ids + presence masks, 2 = + hash insert / verify / probing, 3 = + compaction of the table, 4 = + rank sort and
halo list, 0 = the whole kernel (+ local rulebook rows, unit records)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from treelearn_amd import _hip, geometry as G
from treelearn_amd.synth import CONFIGS, make_batch, make_tile
L = _hip.lib()
if not hasattr(L, "tl_dev_blk_abl"):
    sys.exit("needs the developer build (python -m treelearn_amd.build --dev)")
L.tl_dev_blk_abl.argtypes = [ctypes.c_int]; L.tl_dev_blk_abl.restype = ctypes.c_int
import time
os.environ["TL_BLK_SIDE"] = "0"
for seeds in ((0,), (0, 1)):
    b = make_batch([make_tile(**CONFIGS["config2"], seed=s) for s in seeds])
    c, bi = b["coords"].cuda().float(), b["batch_ids"].cuda().long()
    base = None
    for stage in (0, 1, 2, 3, 4, 0):
        _hip.check(L.tl_dev_blk_abl(stage), "abl")
        ts = []
        for rep in range(25):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            G.build_geometry(c, bi, len(seeds), 0.1, 7, [500, 500, 1000], blocked=True)
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        t = min(ts[3:])
        print(f"tiles {len(seeds)}  stage {stage}: whole geometry {t:.3f} ms", flush=True)
_hip.check(L.tl_dev_blk_abl(0), "abl")
