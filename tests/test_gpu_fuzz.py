"""Randomised point clouds through the whole forward: shapes the synthetic forest generator never makes (thin slabs, single columns, blobs far
from the origin, duplicated points, points on the borders of the voxel grid, batches of very different sizes).  Three checks per case:
tl_forward == the Python-driven engine bit for bit (same launches), geometry == the numpy oracle bit for bit, and the fp32 / bf16x3 forward
within the north star's 1e-3 of the CPU oracle's forward.  Reference path: tree_learn/model/tree_learn.py:87-167."""
import os

import numpy as np
import pytest
import torch

from oracle import model as om, voxel as ov

pytestmark = pytest.mark.gpu
REL_TOL = 1e-3


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def _cloud(rng, kind, n):
    if kind == "slab":                                     # one voxel thick in z
        p = np.column_stack([rng.uniform(0, 14, n), rng.uniform(0, 9, n), rng.uniform(3.0, 3.09, n)])
    elif kind == "column":                                 # a single vertical pole
        p = np.column_stack([rng.uniform(5.0, 5.25, n), rng.uniform(7.0, 7.25, n), rng.uniform(0, 30, n)])
    elif kind == "blobs":                                  # dense clusters, empty space between them
        c = rng.uniform(0, 20, (6, 3)); c[:, 2] *= 0.5
        p = c[rng.integers(0, 6, n)] + rng.normal(0, 0.35, (n, 3))
    elif kind == "far":                                    # coordinates far from the origin and negative (the tile's minimum is subtracted)
        p = rng.uniform(0, 8, (n, 3)) + np.array([-4321.7, 98765.4, -12.3])
    elif kind == "dupes":                                  # many points per voxel, exact duplicates included
        base = rng.uniform(0, 6, (max(n // 8, 1), 3))
        p = np.concatenate([base[rng.integers(0, len(base), n - len(base))], base])
    elif kind == "lattice":                                # points exactly on voxel borders (multiples of the voxel size)
        p = rng.integers(0, 60, (n, 3)).astype(np.float64) * 0.1
    else:                                                  # "sparse": isolated voxels, almost no neighbours
        p = rng.uniform(0, 40, (n, 3)); p[:, 2] *= 0.5
    return p.astype(np.float32)


CASES = [("slab", 4000), ("column", 1500), ("blobs", 20000), ("far", 6000), ("dupes", 9000), ("lattice", 5000), ("sparse", 3000), ("blobs", 300),
         ("dupes", 1), ("lattice", 2), ("sparse", 17)]                     # one point; two points; a handful of isolated voxels


def _model(dtype, seed=7):
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import random_state_dict
    m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=dtype)
    m.load_state_dict(random_state_dict(seed, channels=32, num_blocks=7), strict=True)
    return m.cuda().eval()


def _batch(clouds):
    coords = np.concatenate(clouds)
    bids = np.concatenate([np.full(len(c), i, np.int64) for i, c in enumerate(clouds)])
    return dict(coords=torch.from_numpy(coords), input_feats=torch.ones(len(coords), 1), batch_ids=torch.from_numpy(bids), batch_size=len(clouds))


def _run_both(model, batch):
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    with torch.no_grad():
        assert model._executor(model.active_dtype(False)) is not None
        a = model(gb, return_loss=False)
        os.environ["TL_EXEC"] = "0"
        try:
            b = model(gb, return_loss=False)
        finally:
            del os.environ["TL_EXEC"]
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize("case", range(len(CASES)), ids=[f"{k}_{n}" for k, n in CASES])
def test_random_cloud_forward(case):
    kind, n = CASES[case]
    rng = np.random.default_rng(1000 + case)
    batch = _batch([_cloud(rng, kind, n)])
    ref = None
    for dtype in (torch.bfloat16, torch.float32, "bf16x3"):
        m = _model(dtype)
        a, b = _run_both(m, batch)
        for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
            same = (a[k] == b[k]) | (torch.isnan(a[k]) & torch.isnan(b[k]))
            assert bool(same.all()), (kind, dtype, k, int((~same).sum()))
        if dtype == torch.bfloat16:
            continue
        if ref is None:
            sd = {k: v.detach().cpu().float() for k, v in m.state_dict().items()}
            ref = om.forward(sd, batch["coords"].numpy(), batch["input_feats"].numpy(), batch["batch_ids"].numpy(), 1, voxel_size=0.1, num_blocks=7,
                             spatial_shape=[500, 500, 1000])
        for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
            e = rel_err(a[k].cpu().numpy(), ref[k].numpy())
            assert e < REL_TOL, (kind, dtype, k, e)


def test_random_batch_of_three_unequal_clouds():
    """Three clouds of very different sizes in one batch (tree_learn.py:129-167 voxelizes per batch entry and concatenates): geometry against
    the numpy oracle bit for bit, forward against the oracle's."""
    from treelearn_amd import geometry as G
    rng = np.random.default_rng(77)
    batch = _batch([_cloud(rng, "blobs", 12000), _cloud(rng, "column", 200), _cloud(rng, "slab", 2500)])
    geom = G.build_geometry(batch["coords"].cuda(), batch["batch_ids"].cuda(), 3, 0.1, 7, [500, 500, 1000])
    _, vc, v2p, _ = ov.voxelize(batch["coords"].numpy(), batch["input_feats"].numpy(), batch["batch_ids"].numpy(), 3, 0.1)
    assert np.array_equal(geom.levels[0].coords.cpu().numpy(), vc) and np.array_equal(geom.v2p.cpu().numpy(), v2p)
    m = _model(torch.float32)
    a, b = _run_both(m, batch)
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        assert torch.equal(a[k], b[k]), k
    ref = om.forward({k: v.detach().cpu() for k, v in m.state_dict().items()}, batch["coords"].numpy(), batch["input_feats"].numpy(),
                     batch["batch_ids"].numpy(), 3, voxel_size=0.1, num_blocks=7, spatial_shape=[500, 500, 1000])
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        assert rel_err(a[k].cpu().numpy(), ref[k].numpy()) < REL_TOL, k


@pytest.mark.parametrize("kind,n", [("blobs", 9000), ("slab", 3000), ("dupes", 5000)])
def test_random_cloud_training_step_gradients(kind, n):
    """One training-mode step (batch-statistics BatchNorm, loss, backward) on two random clouds with random labels and masks: the loss and
    EVERY parameter's gradient against float64 autograd through the oracle, kink-pinned (the ReLU branches the HIP forward took are handed
    to the float64 run, oracle.model.train_step_grads(relu_masks=...)).  Reference: tools/training/train.py:30-44."""
    from treelearn_amd import autograd as ag
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import random_state_dict
    rng = np.random.default_rng(500 + n)
    clouds = [_cloud(rng, kind, n), _cloud(rng, "blobs", n // 3)]
    batch = _batch(clouds)
    N = batch["coords"].shape[0]
    batch.update(semantic_labels=torch.from_numpy(rng.integers(0, 2, N)).long(), offset_labels=torch.from_numpy(rng.normal(0, 1, (N, 3)).astype(np.float32)),
                 masks_sem=torch.from_numpy(rng.random(N) < 0.8), masks_off=torch.from_numpy(rng.random(N) < 0.4))
    cfg = dict(channels=16, num_blocks=5)
    sd = random_state_dict(11, **cfg)
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[512, 512, 512], voxel_size=0.1, **cfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    ag.RELU_MASK_SINK = {}
    try:
        loss, _ = model(batch, return_loss=True)
        sink = ag.RELU_MASK_SINK
    finally:
        ag.RELU_MASK_SINK = None
    loss.backward()
    mod_name = {m: k for k, m in model.named_modules()}
    masks = {mod_name[m]: v.cpu() for m, v in sink.items()}
    loss64, g64 = om.train_step_grads(sd, batch, 0.1, cfg["num_blocks"], [512, 512, 512], relu_masks=masks)
    assert float(loss.detach()) == pytest.approx(loss64, rel=1e-4)
    gmax = max(float(v.abs().max()) for v in g64.values())
    worst = (0.0, None)
    for name, p in model.named_parameters():
        b = g64[name].numpy().astype(np.float64)
        if np.abs(b).max() <= 1e-9 * gmax:
            continue
        e = rel_err(p.grad.cpu().numpy().astype(np.float64), b)
        worst = max(worst, (e, name))
        assert e < 2e-4, (kind, name, e)
    print(kind, "worst gradient tensor vs float64:", worst)


@pytest.mark.parametrize("use_coords,use_feats", [(True, False), (False, True), (True, True)])
def test_random_cloud_with_voxel_features(use_coords, use_feats):
    """`use_coords` / `use_feats` (tree_learn.py:129-167: the voxel features are the mean of the first <= 3 points' coordinates / features
    instead of ones): eval forward in fp32 and bf16x3 against the oracle, and one training step's gradients against float64, on a cloud with
    many points per voxel and random point features."""
    from treelearn_amd import autograd as ag
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import random_state_dict
    rng = np.random.default_rng(900 + 2 * use_coords + use_feats)
    batch = _batch([_cloud(rng, "dupes", 6000), _cloud(rng, "blobs", 2500)])
    N = batch["coords"].shape[0]
    batch["input_feats"] = torch.from_numpy(rng.normal(0, 1, (N, 1)).astype(np.float32))
    batch.update(semantic_labels=torch.from_numpy(rng.integers(0, 2, N)).long(), offset_labels=torch.from_numpy(rng.normal(0, 1, (N, 3)).astype(np.float32)),
                 masks_sem=torch.from_numpy(rng.random(N) < 0.7), masks_off=torch.from_numpy(rng.random(N) < 0.5))
    cfg = dict(channels=16, num_blocks=4)
    sd = random_state_dict(13, **cfg)
    ref = om.forward(sd, batch["coords"].numpy(), batch["input_feats"].numpy(), batch["batch_ids"].numpy(), 2, voxel_size=0.1, num_blocks=4,
                     use_coords=use_coords, use_feats=use_feats, spatial_shape=[512, 512, 512])
    for dtype in (torch.float32, "bf16x3"):
        m = TreeLearn(use_feats=use_feats, use_coords=use_coords, spatial_shape=[512, 512, 512], voxel_size=0.1, compute_dtype=dtype, **cfg)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        with torch.no_grad():
            out = m(batch, return_loss=False)
        for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
            e = rel_err(out[k].cpu().numpy(), ref[k].numpy())
            assert e < REL_TOL, (dtype, k, e)
    m = TreeLearn(use_feats=use_feats, use_coords=use_coords, spatial_shape=[512, 512, 512], voxel_size=0.1, **cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    ag.RELU_MASK_SINK = {}
    try:
        loss, _ = m(batch, return_loss=True)
        sink = ag.RELU_MASK_SINK
    finally:
        ag.RELU_MASK_SINK = None
    loss.backward()
    mod_name = {mod: k for k, mod in m.named_modules()}
    masks = {mod_name[mod]: v.cpu() for mod, v in sink.items()}
    loss64, g64 = om.train_step_grads(sd, batch, 0.1, 4, [512, 512, 512], relu_masks=masks, use_coords=use_coords, use_feats=use_feats)
    assert float(loss.detach()) == pytest.approx(loss64, rel=1e-4)
    gmax = max(float(v.abs().max()) for v in g64.values())
    for name, p in m.named_parameters():
        b = g64[name].numpy().astype(np.float64)
        if np.abs(b).max() <= 1e-9 * gmax:
            continue
        e = rel_err(p.grad.cpu().numpy().astype(np.float64), b)
        assert e < 2e-4, (name, e)


def test_forward_input_formats_and_rejections():
    """What a caller may hand `forward` besides contiguous float32 device tensors (the reference's `cuda_cast` + `voxelize`, util/train.py:28-43,
    tree_learn.py:129-167, take whatever the DataLoader collated): a strided view, float64 coordinates, int32 batch ids and host tensors give
    the SAME outputs bit for bit; a batch entry without a single point (batch_size 3, ids 0 and 2) runs -- tl_forward and the Python-driven
    engine agree, the voxelization equals the oracle's --; no points at all, non-finite coordinates and batch ids outside [0, batch_size)
    raise ValueError instead of reaching a kernel."""
    rng = np.random.default_rng(4242)
    m = _model(torch.bfloat16)
    cloud = _cloud(rng, "blobs", 3000)
    batch = _batch([cloud])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    keys = ("backbone_feats", "semantic_prediction_logits", "offset_predictions")
    with torch.no_grad():
        ref = m(gb, return_loss=False)
        wide = torch.zeros(len(cloud), 5, device="cuda"); wide[:, 1:4] = gb["coords"]
        for name, over in (("strided coords", dict(coords=wide[:, 1:4])), ("float64 coords", dict(coords=gb["coords"].double())),
                           ("int32 batch ids", dict(batch_ids=gb["batch_ids"].int())), ("host tensors", dict(coords=batch["coords"], batch_ids=batch["batch_ids"]))):
            out = m(dict(gb, **over), return_loss=False)
            for k in keys:
                assert torch.equal(out[k], ref[k]), (name, k)
        nan = torch.cat([gb["coords"][:-1], torch.full((1, 3), float("nan"), device="cuda")])
        for name, over in (("no points", dict(coords=gb["coords"][:0], batch_ids=gb["batch_ids"][:0], input_feats=gb["input_feats"][:0])),
                           ("nan", dict(coords=nan)), ("inf", dict(coords=nan.nan_to_num(nan=float("inf")))), ("-inf", dict(coords=nan.nan_to_num(nan=float("-inf")))),
                           ("batch id = batch_size", dict(batch_ids=gb["batch_ids"] + 1)), ("negative batch id", dict(batch_ids=gb["batch_ids"] - 1))):
            with pytest.raises(ValueError):
                m(dict(gb, **over), return_loss=False)
            torch.cuda.synchronize()
        out = m(gb, return_loss=False)                             # the context is still usable after the rejections
        for k in keys:
            assert torch.equal(out[k], ref[k]), k
    # a batch entry without points
    from treelearn_amd import geometry as G
    ids = np.where(np.arange(len(cloud)) % 3 == 0, 0, 2).astype(np.int64)
    hole = dict(coords=torch.from_numpy(cloud), input_feats=torch.ones(len(cloud), 1), batch_ids=torch.from_numpy(ids), batch_size=3)
    geom = G.build_geometry(hole["coords"].cuda(), hole["batch_ids"].cuda(), 3, 0.1, 7, [500, 500, 1000])
    _, vc, v2p, _ = ov.voxelize(cloud, hole["input_feats"].numpy(), ids, 3, 0.1)
    assert np.array_equal(geom.levels[0].coords.cpu().numpy(), vc) and np.array_equal(geom.v2p.cpu().numpy(), v2p)
    a, b = _run_both(m, hole)
    for k in keys:
        assert torch.equal(a[k], b[k]), k
