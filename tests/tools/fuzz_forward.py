"""Soak: random small tiles (odd point counts, thin slabs, unequal batches, voxel 0.1-0.3, with / without input features and a
preset spatial_shape) through the HIP forward in fp32 against the oracle, the bf16 mode against fp32, and one training step for
finiteness.      python tests/tools/fuzz_forward.py [first_seed] [count]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import model as om
from treelearn_amd.model import TreeLearn
from treelearn_amd.synth import make_batch, make_tile

def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0; t_start = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    voxel = float(rng.choice([0.1, 0.15, 0.2, 0.3]))
    use_feats = bool(rng.integers(0, 2)); use_coords = bool(rng.integers(0, 2))
    shape = [500, 500, 1000] if rng.integers(0, 2) else None
    tiles = []
    for _ in range(int(rng.integers(1, 4))):
        t = make_tile(extent=float(rng.uniform(3, 12)), voxel=voxel, n_trees=int(rng.integers(0, 5)), fill=float(rng.uniform(0.02, 0.15)), seed=int(rng.integers(1 << 30)))
        keep = rng.uniform(size=len(t["points"])) < rng.uniform(0.05, 1.0)
        keep[:5] = True
        tiles.append({k: (v[keep] if k != "center" else v) for k, v in t.items()})
    batch = make_batch(tiles)
    n = len(batch["coords"])
    sd = om.random_state_dict(seed, channels=32, num_blocks=7)
    kw = dict(use_feats=use_feats, use_coords=use_coords, spatial_shape=shape, voxel_size=voxel)
    model = TreeLearn(**kw); model.load_state_dict(sd, strict=True); model = model.cuda().eval()
    tag = f"seed {seed}: {n} points, batch {len(tiles)}, voxel {voxel}, feats {use_feats}, coords {use_coords}, shape {shape}"
    try:
        ref = om.forward(sd, batch["coords"].numpy(), batch["input_feats"].numpy(), batch["batch_ids"].numpy(), len(tiles), voxel_size=voxel,
                         num_blocks=7, spatial_shape=shape, use_feats=use_feats, use_coords=use_coords)
    except ValueError as e:
        assert "reach zero!!!" in str(e), e
        try:
            with torch.no_grad(): model(batch, return_loss=False)
            print("MISMATCH (oracle raised reach zero, HIP did not):", tag, flush=True); bad += 1
        except ValueError as e2:
            assert "reach zero!!!" in str(e2), e2
            print("ok (reach zero in both):", tag, flush=True)
        continue
    with torch.no_grad():
        out = model(batch, return_loss=False)
    errs = {k: rel_err(out[k].float().cpu().numpy(), ref[k].numpy()) for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions")}
    mb = TreeLearn(**kw, compute_dtype=torch.bfloat16); mb.load_state_dict(sd, strict=True); mb = mb.cuda().eval()
    with torch.no_grad():
        ob = mb(batch, return_loss=False)
    eb = {k: rel_err(ob[k].float().cpu().numpy(), ref[k].numpy()) for k in ("semantic_prediction_logits", "offset_predictions")}
    mt = TreeLearn(**kw, compute_dtype=torch.bfloat16); mt.load_state_dict(sd, strict=True); mt = mt.cuda().train()
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    try:
        loss, _ = mt(gb, return_loss=True); loss.backward()
        gfin = all(torch.isfinite(p.grad).all().item() for p in mt.parameters() if p.grad is not None)
    except ValueError as e:                                  # a level of ONE voxel: nn.BatchNorm1d refuses batch statistics, in the reference as here
        assert "Expected more than 1 value per channel" in str(e), e
        loss, gfin = torch.zeros(()), True
    ok = max(errs.values()) < 1e-3 and all(np.isfinite(list(eb.values()))) and max(eb.values()) < 0.25 and torch.isfinite(loss).item() and gfin
    print(("ok  " if ok else "FAIL"), tag, {k: f"{v:.1e}" for k, v in errs.items()}, "bf16", {k: f"{v:.1e}" for k, v in eb.items()}, f"loss {float(loss):.3f}", flush=True)
    bad += not ok
print(f"{count} cases, {bad} failures, {time.time() - t_start:.0f} s")
sys.exit(1 if bad else 0)
