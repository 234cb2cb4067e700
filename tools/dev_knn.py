"""k-NN vote: cell grid vs brute force on a plot-sized clustered set (timing)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from treelearn_amd.util.postprocess import knn_vote
rng = np.random.default_rng(0)
for nr, nq in ((200_000, 50_000), (2_000_000, 200_000)):
    c = rng.uniform(-50, 50, size=(400, 3)); c[:, 2] = 0
    ref = c[rng.integers(0, 400, nr)] + rng.normal(size=(nr, 3)) * np.array([0.15, 0.15, 0.5])
    qry = c[rng.integers(0, 400, nq)] + rng.normal(size=(nq, 3)) * np.array([0.5, 0.5, 1.0])
    R = torch.from_numpy(ref.astype(np.float32)).cuda(); Q = torch.from_numpy(qry.astype(np.float32)).cuda(); Lb = torch.from_numpy(rng.integers(1, 400, nr)).cuda()
    for mode in ("grid", "brute"):
        if mode == "brute" and nr * nq > 2e11: print(f"  nr {nr} nq {nq} brute: skipped"); continue
        knn_vote(R, Lb, Q[:1000], 5, force=mode); torch.cuda.synchronize(); t0 = time.time()
        out = knn_vote(R, Lb, Q, 5, force=mode); torch.cuda.synchronize()
        print(f"  nr {nr} nq {nq} {mode}: {(time.time() - t0) * 1e3:.1f} ms", flush=True)
