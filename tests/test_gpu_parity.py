"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.
Bit-exact for voxel indices / rulebooks; fp32 features within 1e-3 relative (BASELINE.json north_star)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om
from oracle import sparse_ops as osp
from oracle import voxel as ov
from treelearn_amd.synth import CONFIGS, make_batch, make_tile

REL_TOL = 1e-3          # north_star: semantic/offset tensors within 1e-3 relative fp32


def _dev():
    assert torch.cuda.is_available(), "needs the MI355X"
    return torch.device("cuda:0")


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def _geom(coords, bids, B, vs, levels, sshape):
    from treelearn_amd.geometry import build_geometry
    d = _dev()
    return build_geometry(torch.from_numpy(coords).to(d), torch.from_numpy(bids).to(d), B, vs, levels, sshape)


def _check_geometry(coords, bids, B, vs, levels, sshape):
    g = _geom(coords, bids, B, vs, levels, sshape)
    _, vc, v2p, ss = ov.voxelize(coords, np.zeros((len(coords), 1), np.float32), bids, B, vs)
    shape = np.asarray(sshape if sshape is not None else ss, np.int64)
    np.testing.assert_array_equal(g.levels[0].coords.cpu().numpy(), vc)
    np.testing.assert_array_equal(g.v2p.cpu().numpy(), v2p)
    cur = vc
    for li in range(levels):
        lv = g.levels[li]
        assert lv.n == len(cur)
        np.testing.assert_array_equal(lv.coords.cpu().numpy(), cur)
        np.testing.assert_array_equal(lv.nbr.cpu().numpy().T, ov.rulebook_subm(cur))
        assert tuple(lv.shape) == tuple(int(s) for s in shape)
        if li + 1 < levels:
            cc, parent, child, shape = ov.rulebook_down(cur, shape)
            np.testing.assert_array_equal(lv.parent.cpu().numpy(), parent)
            np.testing.assert_array_equal(lv.child.cpu().numpy().T, child)
            inv = np.full((len(cur), 8), -1, np.int32)
            tap = (cur[:, 1] & 1) * 4 + (cur[:, 2] & 1) * 2 + (cur[:, 3] & 1)
            ok = parent >= 0
            inv[np.where(ok)[0], tap[ok]] = parent[ok]
            np.testing.assert_array_equal(lv.inv.cpu().numpy().T, inv)
            cur = cc
    return g


def test_voxel_rulebook_config1_bit_exact():
    t = make_tile(**CONFIGS["config1"], seed=0)
    bids = np.zeros(len(t["points"]), np.int64)
    _check_geometry(t["points"], bids, 1, 0.2, 7, [500, 500, 1000])


def test_voxel_rulebook_batch2_odd_shapes_bit_exact():
    a = make_tile(extent=7, voxel=0.2, n_trees=2, seed=3)["points"]
    b = make_tile(extent=5, voxel=0.2, n_trees=1, seed=4)["points"]
    a = a[a[:, 2] < 9]; b = b[b[:, 2] < 6]
    coords = np.concatenate([a, b]); bids = np.concatenate([np.zeros(len(a), np.int64), np.ones(len(b), np.int64)])
    _check_geometry(coords, bids, 2, 0.2, 3, None)               # spatial_shape None -> max+1, odd dims drop voxels
    _check_geometry(coords, bids, 2, 0.2, 3, [37, 41, 47])


def test_voxel_duplicates_and_golden_g3(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_voxelize.npz"))
    geom = _geom(g["coords"], g["batch_ids"], 2, 0.2, 1, None)
    np.testing.assert_array_equal(geom.levels[0].coords.cpu().numpy().astype(np.float32), g["c0f0_voxel_coords"])
    np.testing.assert_array_equal(geom.v2p.cpu().numpy(), g["c0f0_v2p"])
    assert tuple(geom.levels[0].shape) == tuple(int(s) for s in g["c0f0_spatial_shape"])
    # optional voxel features (use_coords/use_feats): mean of first <=3 points in input order
    from treelearn_amd.geometry import voxel_mean_feats
    pf = torch.from_numpy(np.concatenate([g["coords"], g["input_feats"]], 1)).to(_dev())
    mean = voxel_mean_feats(pf, geom, 3).cpu().numpy()
    ref = g["c1f1_voxel_feats"]                                  # (feat, x, y, z)
    np.testing.assert_allclose(np.concatenate([mean[:, 3:], mean[:, :3]], 1), ref, rtol=1e-6, atol=1e-6)


def test_reach_zero_raises():
    t = make_tile(extent=3, voxel=0.5, n_trees=1, seed=35)["points"]
    t = t[t[:, 2] < 1.0]
    with pytest.raises(ValueError, match="reach zero!!!"):
        _geom(t, np.zeros(len(t), np.int64), 1, 0.5, 4, None)


@pytest.mark.parametrize("cin,cout,K,n_out", [
    (32, 32, 27, 333), (64, 32, 27, 333), (96, 96, 27, 333), (32, 64, 8, 333), (64, 32, 8, 333), (384, 192, 1, 700),
    (224, 224, 27, 333), (4, 32, 27, 333), (16, 8, 27, 333), (24, 40, 8, 333), (4, 16, 27, 5000),
    # > 16384 rows: the 128-row output-stationary MFMA kernels (smaller sizes take the split-tap small-level kernel)
    (32, 32, 27, 17001), (64, 64, 27, 16500), (128, 64, 27, 16400), (96, 96, 8, 16390), (64, 32, 1, 20000), (224, 224, 8, 16385),
    # mid-size levels: the 4-wave small-level kernel with fragment-order weights (more than 512 output blocks)
    (160, 160, 27, 6500), (320, 160, 27, 6000),
    # the heads' output Linears in training: one thread per row (tl_linear_small.hip)
    (32, 2, 1, 30001), (32, 3, 1, 600), (64, 8, 1, 5000)])
def test_conv_fwd_vs_oracle(cin, cout, K, n_out):
    from treelearn_amd import ops
    rng = np.random.default_rng(cin * 1000 + cout + K)
    d = _dev()
    n_in = 700 if n_out < 1000 else n_out + 77
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    k = round(K ** (1 / 3))
    w = (rng.normal(size=(cout, k, k, k, cin)) / np.sqrt(cin * K)).astype(np.float32)
    table = rng.integers(-1, n_in, size=(n_out, K)).astype(np.int32)
    table[rng.uniform(size=table.shape) < 0.5] = -1
    table[:, K // 2][:50] = -1
    if K == 1:
        n_in = n_out; x = x[:n_in]; table = np.arange(n_in, dtype=np.int32)[:, None]
    sc = rng.uniform(0.5, 1.5, cin).astype(np.float32); sh = rng.normal(0, 0.3, cin).astype(np.float32)
    osc = rng.uniform(0.5, 1.5, cout).astype(np.float32); osh = rng.normal(0, 0.3, cout).astype(np.float32)
    res = rng.normal(size=(n_out, cout)).astype(np.float32)
    xin = np.maximum(x * sc + sh, 0)
    ref = osp.conv_table(torch.from_numpy(xin), torch.from_numpy(w), table, n_out).numpy()
    ref = np.maximum((ref + res) * osc + osh, 0)
    T = lambda a: torch.from_numpy(a).to(d)
    wp = ops.pack_weight(T(w), torch.float32)
    tab = None if K == 1 else T(np.ascontiguousarray(table.T))
    out = ops.conv_fwd(T(x), wp, tab, n_out, in_scale=T(sc), in_shift=T(sh), in_relu=True, residual=T(res),
                       out_scale=T(osc), out_shift=T(osh), out_relu=True)
    assert rel_err(out.cpu().numpy(), ref) < 2e-5
    # plain (no prologue / epilogue), strided views
    ref2 = osp.conv_table(torch.from_numpy(x), torch.from_numpy(w), table, n_out).numpy()
    wide_in = torch.zeros((n_in, cin + 32), device=d); wide_in[:, 32:] = T(x)
    wide_out = torch.zeros((n_out, cout + 64), device=d)
    ops.conv_fwd(wide_in[:, 32:], wp, tab, n_out, out=wide_out[:, 64:])
    assert rel_err(wide_out[:, 64:].cpu().numpy(), ref2) < 2e-5
    assert float(wide_out[:, :64].abs().max()) == 0.0


def _batch_from_golden(g, name):
    keys = ["coords", "input_feats", "batch_ids", "semantic_labels", "instance_labels", "masks_inner", "masks_off",
            "masks_sem", "offset_labels", "centers"]
    b = {k: torch.from_numpy(g[f"{name}_in_{k}"]) for k in keys}
    b["batch_size"] = int(g[f"{name}_in_batch_size"])
    return b


@pytest.mark.parametrize("name", ["m3", "m7", "m2b"])
def test_forward_vs_golden(golden_dir, name):
    """Whole model on the HIP path vs the reference module tree run through the dense stand-in."""
    from treelearn_amd.model import TreeLearn
    g = np.load(os.path.join(golden_dir, "g10_forward.npz"))
    cfg = json.loads(str(g[f"{name}_cfg"]))
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=cfg["spatial_shape"], voxel_size=cfg["voxel_size"], **cfg["cfg"])
    model.load_state_dict(om.random_state_dict(cfg["seed"], **cfg["cfg"]), strict=True)
    model = model.cuda().eval()
    batch = _batch_from_golden(g, name)
    with torch.no_grad():
        out = model(batch, return_loss=False)
        loss, ld = model(batch, return_loss=True)
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        assert out[k].is_cuda
        assert rel_err(out[k].cpu().numpy(), g[f"{name}_eval_{k}"]) < REL_TOL, k
    assert float(loss) == pytest.approx(float(g[f"{name}_eval_loss"]), rel=REL_TOL)


def test_forward_default_model_config1_vs_oracle():
    """Default 7-level / 32-channel model (30.1 M params) on the config-1 tile vs the CPU oracle."""
    from treelearn_amd.model import TreeLearn
    t = make_tile(**CONFIGS["config1"], seed=0)
    batch = make_batch([t])
    sd = om.random_state_dict(7, channels=32, num_blocks=7)
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.2)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    with torch.no_grad():
        out = model(batch, return_loss=False)
    ref = om.forward(sd, batch["coords"].numpy(), batch["input_feats"].numpy(), batch["batch_ids"].numpy(), 1,
                     voxel_size=0.2, num_blocks=7, spatial_shape=[500, 500, 1000])
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < REL_TOL, k
    # determinism: same tile twice -> bit-identical (no atomics in the conv path)
    with torch.no_grad():
        out2 = model(batch, return_loss=False)
    for k in out:
        assert torch.equal(out[k], out2[k])


def test_forward_fp32_mid_tile_vs_oracle():
    """12 x 12 m tile at 0.1 m (140 k points; levels 1-3 above the 16 k-row threshold, so the fp32 direct and stream kernels run
    on real rulebooks, the deeper levels on the small-level kernel): fp32 forward within 1e-3 of the CPU oracle."""
    from treelearn_amd.model import TreeLearn
    t = make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.10, seed=0)
    batch = make_batch([t])
    sd = om.random_state_dict(7, channels=32, num_blocks=7)
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    with torch.no_grad():
        out = model(batch, return_loss=False)
    ref = om.forward(sd, batch["coords"].numpy(), batch["input_feats"].numpy(), batch["batch_ids"].numpy(), 1,
                     voxel_size=0.1, num_blocks=7, spatial_shape=[500, 500, 1000])
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < REL_TOL, k


def test_tile_loop_golden_g9(golden_dir):
    """get_pointwise_preds semantics (skip rule, +centers, masks_inner) with the golden's fake model."""
    from treelearn_amd.util.pipeline import get_pointwise_preds
    g = np.load(os.path.join(golden_dir, "g9_tile_loop.npz"))

    class Fake(torch.nn.Module):
        def forward(self, batch, return_loss):
            c = batch["coords"].cuda()
            if float(c[:, 0].mean()) > 900:
                raise RuntimeError("your out spatial shape reach zero!!! (fake)")
            return dict(offset_predictions=c * 0.5 + 1, semantic_prediction_logits=torch.stack([c[:, 0], -c[:, 1]], 1),
                        backbone_feats=c.repeat(1, 11)[:, :32])
    keys = ["coords", "input_feats", "batch_ids", "semantic_labels", "instance_labels", "masks_inner", "masks_off",
            "masks_sem", "offset_labels", "centers"]
    tiles = []
    for i in range(3):
        b = {k: torch.from_numpy(g[f"t{i}_{k}"]) for k in keys}; b["batch_size"] = 1
        tiles.append(b)
    res = get_pointwise_preds(Fake(), tiles, dict(voxel_size=0.2))
    for i, r in enumerate(res):
        np.testing.assert_allclose(r, g[f"out{i}"], rtol=1e-6, atol=1e-6)
    dres = get_pointwise_preds(Fake(), tiles, dict(voxel_size=0.2), keep_on_device=True)          # same results, left on the GPU
    for r, d in zip(res, dres):
        assert d.is_cuda and np.array_equal(d.cpu().numpy(), r)


def test_tile_loop_edge_cases_of_the_result_path():
    """The worker-thread result path (util/pipeline._ResultSink) at its edges, with a fake model: a FIRST tile without a single inner point
    (the result arrays are sized from it), an iterable without a length, every tile skipped by the "reach zero" rule, an empty iterable, and
    a forward that raises something else in the middle of the loop (the error reaches the caller, no thread is left behind)."""
    import threading
    from treelearn_amd.util.pipeline import get_pointwise_preds
    rng = np.random.default_rng(5)

    class Fake(torch.nn.Module):
        def forward(self, batch, return_loss):
            c = batch["coords"].cuda()
            if float(c[0, 0]) > 900:
                raise RuntimeError("your out spatial shape reach zero!!! (fake)")
            if float(c[0, 0]) < -900:
                raise ValueError("something else")
            return dict(offset_predictions=c * 0.5 + 1, semantic_prediction_logits=torch.stack([c[:, 0], -c[:, 1]], 1), backbone_feats=c.repeat(1, 11)[:, :32])

    def tile(n, n_inner, x0=0.0):
        c = torch.from_numpy(rng.uniform(-5, 5, (n, 3)).astype(np.float32)); c[0, 0] = x0
        m = torch.zeros(n, dtype=torch.bool); m[torch.from_numpy(rng.permutation(n)[:n_inner])] = True
        return dict(coords=c, input_feats=torch.from_numpy(rng.normal(size=(n, 1)).astype(np.float32)), batch_ids=torch.zeros(n, dtype=torch.int64),
                    semantic_labels=torch.from_numpy(rng.integers(0, 3, n)), instance_labels=torch.from_numpy(rng.integers(0, 50, n)), masks_inner=m,
                    offset_labels=torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)), centers=torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)),
                    batch_size=1)
    tiles = [tile(500, 0), tile(3000, 700), tile(40, 40), tile(900, 0), tile(7000, 6500), tile(100, 1, x0=950.0), tile(2500, 800), tile(60, 1)]
    want_rows = [(0, 0), (1, 700), (2, 40), (3, 0), (4, 6500), (6, 800), (7, 1)]
    n_threads = threading.active_count()
    for src in (tiles, (t for t in tiles), [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in t.items()} for t in tiles]):
        res, rows = get_pointwise_preds(Fake(), src, dict(voxel_size=0.2), return_tile_rows=True)
        assert rows == want_rows
        keep = [t for i, t in enumerate(tiles) if i != 5]
        np.testing.assert_array_equal(res[4], np.concatenate([(t["coords"] + t["centers"]).numpy()[t["masks_inner"].numpy()] for t in keep]))
        np.testing.assert_array_equal(res[2], np.concatenate([(t["coords"] * 0.5 + 1).numpy()[t["masks_inner"].numpy()] for t in keep]))
        for i, k in ((1, "semantic_labels"), (5, "instance_labels"), (3, "offset_labels"), (7, "input_feats")):
            ref = np.concatenate([t[k].numpy()[t["masks_inner"].numpy()] for t in keep])
            assert res[i].dtype == ref.dtype and res[i].shape == ref.shape
            np.testing.assert_array_equal(res[i], ref)
        assert res[6].shape == (8041, 32) and res[0].shape == (8041, 2)
    class OnHost(Fake):                                                 # a model that hands its results back on the host (on a GPU machine)
        def forward(self, batch, return_loss):
            return {k: v.cpu() for k, v in super().forward(batch, return_loss).items()}
    res2, rows2 = get_pointwise_preds(OnHost(), tiles, dict(voxel_size=0.2), return_tile_rows=True)
    assert rows2 == want_rows
    for a, b_ in zip(res2, res):
        np.testing.assert_array_equal(a, b_)
    res = get_pointwise_preds(Fake(), [tile(64, 10, x0=950.0), tile(64, 3, x0=990.0)], dict(voxel_size=0.2))       # every tile skipped
    assert len(res) == 8 and all(len(r) == 0 for r in res)
    res = get_pointwise_preds(Fake(), [], dict(voxel_size=0.2))
    assert len(res) == 8 and all(len(r) == 0 for r in res)
    with pytest.raises(ValueError, match="something else"):
        get_pointwise_preds(Fake(), [tile(300, 30), tile(300, 30), tile(300, 30, x0=-950.0), tile(300, 30)], dict(voxel_size=0.2))
    assert threading.active_count() == n_threads


def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize("cin,cout,K,n_out", [
    (32, 32, 27, 401), (64, 64, 27, 401), (96, 96, 27, 401), (128, 64, 27, 401), (64, 32, 8, 401), (384, 192, 1, 900), (224, 224, 27, 401),
    (16, 8, 27, 401), (4, 32, 27, 3000),
    (32, 32, 27, 17001), (64, 64, 27, 16500), (96, 96, 27, 16400), (128, 64, 8, 16390), (64, 32, 1, 20000), (224, 224, 8, 16385), (160, 160, 8, 16385)])
def test_conv_fwd_bf16_vs_oracle(cin, cout, K, n_out):
    """bf16 storage + bf16 MFMA, fp32 accumulate: compare with the oracle fed the same bf16-rounded operands."""
    from treelearn_amd import ops
    rng = np.random.default_rng(cin * 7 + cout + K)
    d = _dev()
    n_in = 900 if n_out < 1000 else n_out + 77
    x = _bf16_round(rng.normal(size=(n_in, cin)).astype(np.float32))
    k = round(K ** (1 / 3))
    w = _bf16_round((rng.normal(size=(cout, k, k, k, cin)) / np.sqrt(cin * K)).astype(np.float32))
    table = rng.integers(-1, n_in, size=(n_out, K)).astype(np.int32)
    table[rng.uniform(size=table.shape) < 0.4] = -1
    if K == 1:
        n_in = n_out; x = x[:n_in]; table = np.arange(n_in, dtype=np.int32)[:, None]
    sc = rng.uniform(0.5, 1.5, cin).astype(np.float32); sh = rng.normal(0, 0.3, cin).astype(np.float32)
    res = _bf16_round(rng.normal(size=(n_out, cout)).astype(np.float32))
    xin = _bf16_round(np.maximum(x * sc + sh, 0).astype(np.float32))
    ref = osp.conv_table(torch.from_numpy(xin), torch.from_numpy(w), table, n_out).numpy() + res
    T = lambda a, dt=torch.float32: torch.from_numpy(a).to(d).to(dt)
    wp = ops.pack_weight(T(w), torch.bfloat16)
    tab = None if K == 1 else torch.from_numpy(np.ascontiguousarray(table.T)).to(d)
    out = ops.conv_fwd(T(x, torch.bfloat16), wp, tab, n_out, in_scale=T(sc), in_shift=T(sh), in_relu=True, residual=T(res, torch.bfloat16))
    assert out.dtype == torch.bfloat16
    assert rel_err(out.float().cpu().numpy(), ref) < 8e-3          # one bf16 rounding of the output (2^-8)


@pytest.mark.parametrize("cin,cout,n", [(32, 2, 100003), (32, 3, 4097), (64, 4, 9000)])
def test_head_output_linear_kernels_bf16(cin, cout, n):
    """The heads' output Linears in mixed-precision training (tl_linear_small.hip: one thread per row, forward and weight gradient)
    against float64 on the same bf16-rounded operands; deterministic."""
    from treelearn_amd import ops
    rng = np.random.default_rng(cin + cout)
    d = _dev()
    x = _bf16_round(rng.normal(size=(n, cin)).astype(np.float32)); w = _bf16_round((rng.normal(size=(cout, 1, 1, 1, cin)) / 6).astype(np.float32))
    g = _bf16_round(rng.normal(size=(n, cout)).astype(np.float32))
    T = lambda a, dt=torch.bfloat16: torch.from_numpy(a).to(d).to(dt)
    out = ops.conv_fwd(T(x), ops.pack_weight(T(w, torch.float32), torch.bfloat16), None, n)
    ref = x.astype(np.float64) @ w.reshape(cout, cin).astype(np.float64).T
    assert out.dtype == torch.bfloat16 and rel_err(out.float().cpu().numpy(), ref) < 8e-3
    gw = ops.conv_wgrad(T(x), T(g), None, n, 1)
    gw2 = ops.conv_wgrad(T(x), T(g), None, n, 1)
    assert torch.equal(gw, gw2)
    refw = (g.astype(np.float64).T @ x.astype(np.float64))[None]
    assert rel_err(gw.cpu().numpy(), refw) < 2e-5


@pytest.mark.parametrize("device_tiles", [False, True])
def test_tile_loop_same_results_with_tiles_in_flight(device_tiles, monkeypatch):
    """The production tile loop with 1 / 2 / 3 tiles in flight (compute streams round-robin, read-back on its own stream) over
    tiles of unequal size returns exactly the same arrays and per-tile row counts, both through numpy and left on the device --
    nothing may depend on which stream a tile ran on or on when its buffers were recycled."""
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.util import get_pointwise_preds
    tiles = []
    for s_, ext in enumerate((14.0, 9.0, 16.0, 11.0, 13.0, 8.0, 15.0)):
        t = make_tile(extent=ext, voxel=0.1, n_trees=4 + s_, fill=0.08, seed=20 + s_)
        t["center"] = np.array([3.0 * s_, 0.0, 0.0])
        b = make_batch([t], inner_square_edge_length=6.0)
        tiles.append({k: (v.cuda() if (device_tiles and torch.is_tensor(v)) else v) for k, v in b.items()})
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
    model.load_state_dict(om.random_state_dict(7, channels=32, num_blocks=7)); model = model.cuda().eval()
    ref = None
    for nf in ("3", "1", "2", "3"):          # three in flight first: the fresh model's plan (packed weights) must be ready for every stream
        monkeypatch.setenv("TL_TILES_IN_FLIGHT", nf)
        for keep in (False, True):
            res, rows = get_pointwise_preds(model, tiles, dict(voxel_size=0.1), return_tile_rows=True, keep_on_device=keep)
            res = [r.cpu().numpy() if torch.is_tensor(r) else r for r in res]
            if ref is None:
                ref = (res, rows)
                assert len(rows) == len(tiles) and sum(n for _, n in rows) == len(res[0]) > 0
                continue
            assert rows == ref[1], (nf, keep)
            for a, b_ in zip(res, ref[0]):
                np.testing.assert_array_equal(a, b_, err_msg=f"tiles in flight {nf}, keep_on_device {keep}")
    # the numpy results are assembled by a worker thread from pinned landing buffers (util/pipeline._ResultSink); the plain form (every tile
    # copied home synchronously, one torch.cat at the end) and an iterable without a length (the result arrays grow) give the same arrays,
    # and the labels keep the batch's integer type
    for how in ("plain", "generator", "generator, labels int16 on the device"):
        monkeypatch.setenv("TL_RESULT_SINK", "0" if how == "plain" else "1")
        src = tiles if how == "plain" else (t for t in tiles)
        if how.endswith("device"):
            src = [dict(t, semantic_labels=t["semantic_labels"].to(torch.int16).cuda(), instance_labels=t["instance_labels"].to(torch.int32).cuda()) for t in tiles]
        res, rows = get_pointwise_preds(model, src, dict(voxel_size=0.1), return_tile_rows=True)
        assert rows == ref[1]
        for i, (a, b_) in enumerate(zip(res, ref[0])):
            np.testing.assert_array_equal(a, b_, err_msg=how)
            assert a.dtype == b_.dtype or (how.endswith("device") and i in (1, 5)), (how, i, a.dtype, b_.dtype)
        if how.endswith("device"):
            assert res[1].dtype == np.int16 and res[5].dtype == np.int32
    monkeypatch.setenv("TL_RESULT_SINK", "1")
    res = get_pointwise_preds(model, tiles, dict(voxel_size=0.1))
    assert res[1].dtype == tiles[0]["semantic_labels"].cpu().numpy().dtype and res[5].dtype == tiles[0]["instance_labels"].cpu().numpy().dtype
    host = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in t.items()} for t in tiles]
    inner = np.concatenate([t["semantic_labels"].numpy()[t["masks_inner"].numpy()] for t in host])
    np.testing.assert_array_equal(res[1], inner)                         # (bit patterns through the float columns: exact)
    cc = np.concatenate([(t["coords"] + t["centers"]).numpy()[t["masks_inner"].numpy()] for t in host])
    np.testing.assert_array_equal(res[4], cc)


@pytest.mark.parametrize("n_pts", [1, 31, 32, 33, 4097, 70001])
def test_head_kernels_vs_torch(n_pts):
    """tl_head_mlp (v2p gather + output_layer BN/ReLU + both MLPs, tree_learn.py:93-103, blocks.py:8-26) against plain torch fp32:
    the fp32 kernel to 1e-5, the bf16 C = 32 kernel (hidden layers on bf16 MFMAs) to bf16 accuracy; the optional backbone
    output is the fp32 relu(bn(x)) of the gathered row in both; ragged point counts around the 32-point MFMA tile."""
    from treelearn_amd import ops
    rng = np.random.default_rng(n_pts)
    M, C = max(n_pts // 3, 5), 32
    x = torch.from_numpy(rng.normal(size=(M, C)).astype(np.float32)).cuda()
    v2p = torch.from_numpy(rng.integers(0, M, size=n_pts)).cuda()
    so = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).cuda(); ho = torch.from_numpy(rng.normal(0, 0.2, C).astype(np.float32)).cuda()
    w1 = torch.from_numpy((rng.normal(size=(2, C, C)) / np.sqrt(C)).astype(np.float32)).cuda(); b1 = torch.from_numpy(rng.normal(0, 0.1, (2, C)).astype(np.float32)).cuda()
    w2 = torch.from_numpy((rng.normal(size=(5, C)) / np.sqrt(C)).astype(np.float32)).cuda(); b2 = torch.from_numpy(rng.normal(0, 0.1, 5).astype(np.float32)).cuda()

    def ref(xf):
        f = torch.relu(xf[v2p] * so + ho)
        h0 = torch.relu(f @ w1[0].T + b1[0]); h1 = torch.relu(f @ w1[1].T + b1[1])
        return f, h0 @ w2[:2].T + b2[:2], h1 @ w2[2:].T + b2[2:]

    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1.5e-2)):
        xin = x.to(dt)
        f, lg, of = ref(xin.float())
        bb, lo, off = ops.head_mlp(xin, v2p, so, ho, w1, b1, w2, b2, True)
        assert torch.allclose(bb, f, rtol=1e-6, atol=1e-6)          # one fma + max per element (torch rounds the product first)
        scale = float(torch.cat([lg, of], 1).abs().max()) + 1e-6
        assert float((lo - lg).abs().max()) / scale < tol and float((off - of).abs().max()) / scale < tol, dt
        _, lo2, off2 = ops.head_mlp(xin, v2p, so, ho, w1, b1, w2, b2, False)
        assert torch.equal(lo2, lo) and torch.equal(off2, off)


def test_forward_bf16_close_to_fp32():
    """Throughput mode (bf16 features/weights, fp32 accumulate) stays close to the fp32 parity path."""
    from treelearn_amd.model import TreeLearn
    t = make_tile(**CONFIGS["config1"], seed=0)
    batch = make_batch([t])
    sd = om.random_state_dict(7, channels=32, num_blocks=7)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.2, compute_dtype=dt)
        model.load_state_dict(sd, strict=True)
        model = model.cuda().eval()
        with torch.no_grad():
            outs[dt] = {k: v.float().cpu().numpy() for k, v in model(batch, return_loss=False).items()}
    for k in ("semantic_prediction_logits", "offset_predictions"):
        assert rel_err(outs[torch.bfloat16][k], outs[torch.float32][k]) < 5e-2, k


def test_dbscan_grid_vs_golden(golden_dir):
    """GPU eps-graph components == sklearn DBSCAN(min_samples=2) incl. numbering (golden from the reference)."""
    from treelearn_amd.util.pipeline import group_dbscan, get_instances
    g = np.load(os.path.join(golden_dir, "g5_clustering.npz"))
    np.testing.assert_array_equal(group_dbscan(g["gd_xy"], 0.15, 50, -1, 1), g["gd_pred"])
    cfg = dict(tree_conf_thresh=0.5, tau_vert=0.6, tau_off=4, tau_group=0.15, tau_min=50, use_hdbscan=False)
    for case in "ab":
        pred = get_instances(g[f"{case}_coords"], g[f"{case}_offsets"], g[f"{case}_logits"], cfg, g[f"{case}_vert"], 0, 0, -1, 1)
        np.testing.assert_array_equal(pred, g[f"{case}_dbscan_pred"])


def test_dbscan_grid_vs_sklearn_random():
    from sklearn.cluster import DBSCAN
    from treelearn_amd.cluster import dbscan_min2
    rng = np.random.default_rng(0)
    for n, spread in ((1, 1.0), (2, 0.05), (5000, 3.0), (60000, 12.0)):
        xy = (rng.normal(size=(n, 2)) * spread).astype(np.float32)
        xy[: n // 3] = (rng.integers(-20, 20, size=(n // 3, 2)) * 0.5 + rng.normal(size=(n // 3, 2)) * 0.03).astype(np.float32)
        ref = DBSCAN(eps=0.15, min_samples=2).fit(xy).labels_
        np.testing.assert_array_equal(dbscan_min2(xy, 0.15), ref)


@pytest.mark.parametrize("name", ["m3", "m2b"])
def test_training_step_vs_golden(golden_dir, name):
    """Training-mode forward (batch-statistics BatchNorm) + backward through the HIP convs vs the reference
    module tree run through the dense stand-in: loss, running stats, gradient norms, two full gradients."""
    from treelearn_amd.model import TreeLearn
    g = np.load(os.path.join(golden_dir, "g10_forward.npz"))
    cfg = json.loads(str(g[f"{name}_cfg"]))
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=cfg["spatial_shape"], voxel_size=cfg["voxel_size"], **cfg["cfg"])
    model.load_state_dict(om.random_state_dict(cfg["seed"], **cfg["cfg"]), strict=True)
    model = model.cuda().train()
    batch = _batch_from_golden(g, name)
    # also check the unfused eval path == fused eval path (module-by-module vs engine)
    model.eval()
    with torch.no_grad():
        fused = model(batch, return_loss=False)
    unf = model(batch, return_loss=False)                       # grad enabled -> module-by-module path
    for k in ("semantic_prediction_logits", "offset_predictions"):
        assert rel_err(unf[k].detach().cpu().numpy(), fused[k].cpu().numpy()) < 1e-4
    model.train()
    model.zero_grad()
    loss, ld = model(batch, return_loss=True)
    loss.backward()
    assert float(loss) == pytest.approx(float(g[f"{name}_train_loss"]), rel=REL_TOL)
    assert float(ld["offset_loss"]) == pytest.approx(float(g[f"{name}_train_offset_loss"]), rel=REL_TOL)
    np.testing.assert_allclose(model.output_layer[0].running_mean.cpu().numpy(), g[f"{name}_bn_running_mean_after"], rtol=1e-3, atol=1e-5)
    names = [str(s) for s in g[f"{name}_grad_names"]]
    params = dict(model.named_parameters())
    ours = np.array([float(params[n].grad.norm()) for n in names])
    ref = g[f"{name}_grad_norms"]
    np.testing.assert_allclose(ours, ref, rtol=5e-3, atol=1e-6 * ref.max())
    assert rel_err(model.input_conv[0].weight.grad.cpu().numpy(), g[f"{name}_grad_input_conv"]) < 5e-3
    assert rel_err(model.semantic_linear[3].weight.grad.cpu().numpy(), g[f"{name}_grad_sem3"]) < 5e-3


def _same_partition(a, b):
    """labels equal up to renaming of the non-noise clusters (noise = -1 must coincide)."""
    a = np.asarray(a); b = np.asarray(b)
    if not np.array_equal(a == -1, b == -1):
        return False
    pairs = set(zip(a[a != -1].tolist(), b[b != -1].tolist()))
    return len(pairs) == len(set(p[0] for p in pairs)) == len(set(p[1] for p in pairs))


def test_hdbscan_vs_golden_and_sklearn(golden_dir):
    from sklearn.cluster import HDBSCAN
    from treelearn_amd.cluster import hdbscan
    from treelearn_amd.util.pipeline import get_instances
    g = np.load(os.path.join(golden_dir, "g5_clustering.npz"))
    cfg = dict(tree_conf_thresh=0.5, tau_vert=0.6, tau_off=4, tau_group=0.15, tau_min=50, use_hdbscan=True)
    for case in "ab":
        pred = get_instances(g[f"{case}_coords"], g[f"{case}_offsets"], g[f"{case}_logits"], cfg, g[f"{case}_vert"], 0, 0, -1, 1)
        ref = g[f"{case}_hdbscan_pred"]
        np.testing.assert_array_equal(pred <= 0, ref <= 0)              # non-tree (0) / unassigned (-1) coincide
        np.testing.assert_array_equal(pred == 0, ref == 0)
        assert _same_partition(np.where(pred > 0, pred, -1), np.where(ref > 0, ref, -1))
        np.testing.assert_array_equal(pred, ref)                        # and, on this data, the numbering too
    rng = np.random.default_rng(5)
    for n, k in ((400, 10), (3000, 50), (12000, 50)):
        centers = rng.uniform(-20, 20, size=(9, 2))
        xy = np.concatenate([c + rng.normal(size=(n // 10, 2)) * rng.uniform(0.05, 0.6) for c in centers] + [rng.uniform(-25, 25, size=(n // 10, 2))]).astype(np.float32)
        skl = HDBSCAN(min_cluster_size=k).fit(xy)
        ref = skl.labels_
        ours, (_, _, ew) = hdbscan(xy, k, return_mst=True)
        # Demonstration that any difference is sklearn's tie ordering, not a different tree: every minimum spanning tree of a graph
        # has the same multiset of edge weights, and ours equals sklearn's to the last float64 bit (same mutual-reachability
        # arithmetic) ...
        sw = np.sort(skl._single_linkage_tree_["value"])
        np.testing.assert_array_equal(np.sort(ew), sw)
        # ... while some of those weights are exact ties (an edge's weight is often some point's core distance, shared by all edges
        # into that point); sklearn orders equal-weight edges with numpy's unstable quicksort, so points on a split born at a tied
        # lambda may fall on either side
        ties = int((np.diff(sw) == 0).sum())
        assert ties > 0, (n, k, ties)
        if not np.array_equal(ours, ref):
            print(f"hdbscan n={len(xy)} k={k}: {int((ours != ref).sum())} of {len(xy)} labels differ from sklearn; {ties} of {len(sw)} MST weights are exact ties")
        # Same clusters, >= 99 % identical point assignments.
        assert len(set(ours[ours >= 0])) == len(set(ref[ref >= 0])), (n, k)
        agree = 0
        for c in set(ours.tolist()):
            m = ours == c
            vals, cnts = np.unique(ref[m], return_counts=True)
            agree += cnts.max() if c != -1 else int((ref[m] == -1).sum())
        assert agree / len(ours) >= 0.99, (n, k, agree / len(ours))
    # min_cluster_size beyond the 128 neighbours the register kernels keep (reference util/pipeline.py:184-191 accepts any tau_min):
    # heaps in the workspace; same clusters as sklearn, MST weight multiset bit-equal
    for k in (200, 513):
        skl = HDBSCAN(min_cluster_size=k).fit(xy)
        ours, (_, _, ew) = hdbscan(xy, k, return_mst=True)
        np.testing.assert_allclose(np.sort(ew), np.sort(skl._single_linkage_tree_["value"]), rtol=1e-14, atol=0)     # (one weight of 11 999 is an ulp off at k = 200)
        assert len(set(ours[ours >= 0])) == len(set(skl.labels_[skl.labels_ >= 0])) and (ours == skl.labels_).mean() >= 0.99, k
    with pytest.raises(ValueError, match="exceeds the 4096"):
        hdbscan(np.zeros((5000, 2), np.float32), 4097)
    with pytest.raises(ValueError, match="Prim form"):
        hdbscan(xy, 200, algorithm="prim")


def test_hdbscan_auto_takes_prim_order_on_tie_heavy_input():
    """algorithm="auto" above the grid threshold: where more than 10 % of the MST weights are exact ties (quantised coordinates) the
    tree is re-built in Prim's order, so the default equals the Prim form (= sklearn's labels) there; on tie-poor data it keeps the
    fast grid form."""
    from treelearn_amd import cluster
    from treelearn_amd.cluster import hdbscan
    rng = np.random.default_rng(11)
    xy = (np.round(rng.uniform(0, 30, size=(9000, 2)) * 4) / 4).astype(np.float32)            # 0.25 m lattice with duplicates: ties everywhere
    calls = []
    orig = cluster.hdbscan
    cluster.hdbscan = lambda *a, **k: (calls.append(k.get("algorithm")), orig(*a, **k))[1]
    try:
        auto = hdbscan(xy, 5)
    finally:
        cluster.hdbscan = orig
    np.testing.assert_array_equal(auto, hdbscan(xy, 5, algorithm="prim"))
    c = rng.uniform(0, 60, (12, 2)); smooth = (c[rng.integers(0, 12, 9000)] + rng.normal(0, 0.7, (9000, 2))).astype(np.float32)
    np.testing.assert_array_equal(hdbscan(smooth, 50), hdbscan(smooth, 50, algorithm="grid"))


def test_hdbscan_auto_keeps_the_grid_tree_above_the_prim_cap(monkeypatch):
    """The tie-heavy fallback to the O(n^2) Prim form is bounded (cluster.PRIM_FALLBACK_MAX_POINTS): above it `auto` keeps the quadtree
    form's tree and warns; the result is the grid form's."""
    from treelearn_amd import cluster
    rng = np.random.default_rng(12)
    xy = (np.round(rng.uniform(0, 30, size=(9000, 2)) * 4) / 4).astype(np.float32)
    monkeypatch.setattr(cluster, "PRIM_FALLBACK_MAX_POINTS", 8500)
    with pytest.warns(RuntimeWarning, match="tied tree weights"):
        auto = cluster.hdbscan(xy, 5)
    np.testing.assert_array_equal(auto, cluster.hdbscan(xy, 5, algorithm="grid"))


def _hdb_device_stage(xy, k, grid):
    """(core f64[n], sorted MST weights) straight through the C ABI, Prim form or quadtree/Boruvka form"""
    import ctypes as C
    from treelearn_amd import _hip
    L = _hip.lib(); t = torch.from_numpy(np.ascontiguousarray(xy, np.float32)).cuda(); n = len(xy)
    es = torch.empty(n - 1, dtype=torch.int32, device="cuda"); ed = torch.empty_like(es)
    ew = torch.empty(n - 1, dtype=torch.float64, device="cuda"); core = torch.empty(n, dtype=torch.float64, device="cuda")
    if grid:
        g = _hip.HdbGrid(); pws = torch.empty(int(L.tl_hdbscan_grid_plan_ws_bytes()), dtype=torch.uint8, device="cuda")
        _hip.check(L.tl_hdbscan_grid_plan(_hip.ptr(t), n, C.addressof(g), _hip.ptr(pws), _hip.stream()), "plan")
        ws = torch.empty(int(L.tl_hdbscan_grid_ws_bytes(n, C.addressof(g))), dtype=torch.uint8, device="cuda")
        _hip.check(L.tl_hdbscan_mst_grid(_hip.ptr(t), n, k, C.addressof(g), _hip.ptr(es), _hip.ptr(ed), _hip.ptr(ew), _hip.ptr(core), _hip.ptr(ws), _hip.stream()), "grid")
    else:
        ws = torch.empty(int(L.tl_hdbscan_ws_bytes(n)), dtype=torch.uint8, device="cuda")
        _hip.check(L.tl_hdbscan_mst(_hip.ptr(t), n, k, _hip.ptr(es), _hip.ptr(ed), _hip.ptr(ew), _hip.ptr(core), _hip.ptr(ws), _hip.stream()), "prim")
    torch.cuda.synchronize()
    return core.cpu().numpy(), es.cpu().numpy(), ed.cpu().numpy(), ew.cpu().numpy()


def _hdb_cases():
    rng = np.random.default_rng(11)
    c = rng.uniform(0, 60, (25, 2))
    blobs = (c[rng.integers(0, 25, 30000)] + rng.normal(0, 0.15, (30000, 2))).astype(np.float32)
    blobs[:1500] = rng.uniform(0, 60, (1500, 2))
    quant = (np.round((c[rng.integers(0, 12, 6000)] + rng.normal(0, 0.3, (6000, 2))) * 20) / 20).astype(np.float32)      # 0.05 m lattice: duplicates, tied weights
    return {
        "blobs + scattered noise, 30 k": (blobs, 50, True),
        "quantised blobs": (quant, 50, True),
        "uniform": (rng.uniform(0, 10, (8000, 2)).astype(np.float32), 50, True),
        "line": (np.stack([np.linspace(0, 50, 4000), np.zeros(4000)], 1).astype(np.float32), 20, True),
        "far outliers": (np.concatenate([rng.normal(0, 0.1, (3000, 2)), rng.normal(0, 0.1, (3000, 2)) + 5, [[1e4, 1e4], [-1e4, 3e3]]]).astype(np.float32), 50, True),
        "all identical": (np.full((500, 2), 3.5, np.float32), 5, True),
        "two points": (np.array([[0, 0], [1, 1]], np.float32), 2, True),
        "n == min_samples": (rng.normal(size=(50, 2)).astype(np.float32), 50, True),
        # every edge of a perfect lattice has the same weight: any spanning tree is minimal and the labels are whatever the tie
        # order makes them (sklearn's too); only the tree's weights are comparable
        "perfect lattice": (np.stack(np.meshgrid(np.arange(40), np.arange(40)), -1).reshape(-1, 2).astype(np.float32) * 0.1, 10, False),
    }


@pytest.mark.parametrize("case", list(_hdb_cases()))
def test_hdbscan_grid_form_equals_prim_form(case):
    """tl_hdbscan_mst_grid (quadtree k-NN + Boruvka) against tl_hdbscan_mst (the O(n^2) restatement of sklearn's two stages): core
    distances to the bit, a spanning tree with the same weight multiset (= minimal), each edge's weight equal to the mutual
    reachability of its ends, and -- after tl_hdbscan_prim_order_host -- identical labels including the numbering."""
    from treelearn_amd.cluster import hdbscan
    xy, k, labels_too = _hdb_cases()[case]
    n = len(xy)
    core_p, _, _, w_p = _hdb_device_stage(xy, k, grid=False)
    core_g, s_g, d_g, w_g = _hdb_device_stage(xy, k, grid=True)
    np.testing.assert_array_equal(core_g, core_p)
    np.testing.assert_array_equal(np.sort(w_g), np.sort(w_p))
    X = xy.astype(np.float64)
    dist = np.sqrt(((X[s_g] - X[d_g]) ** 2).sum(1))
    np.testing.assert_array_equal(w_g, np.maximum(np.maximum(core_g[s_g], core_g[d_g]), dist))
    parent = np.arange(n)                                      # the n-1 edges span the points
    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]; a = parent[a]
        return a
    for a, b in zip(s_g.tolist(), d_g.tolist()):
        ra, rb = find(a), find(b)
        assert ra != rb
        parent[ra] = rb
    w2 = _hdb_device_stage(xy, k, grid=True)                   # deterministic: the same edge set on a second run
    assert sorted(zip(s_g.tolist(), d_g.tolist())) == sorted(zip(w2[1].tolist(), w2[2].tolist()))
    if labels_too:
        np.testing.assert_array_equal(hdbscan(xy, k, algorithm="grid"), hdbscan(xy, k, algorithm="prim"))


def test_hdbscan_grid_form_vs_sklearn_and_goldens(golden_dir):
    from sklearn.cluster import HDBSCAN
    from treelearn_amd import cluster
    from treelearn_amd.util.pipeline import get_instances
    for seed in range(3):
        r = np.random.default_rng(seed); c = r.uniform(0, 40, (10, 2))
        xy = (c[r.integers(0, 10, 5000)] + r.normal(0, 0.2, (5000, 2))).astype(np.float32)
        np.testing.assert_array_equal(cluster.hdbscan(xy, 50, algorithm="grid"), HDBSCAN(min_cluster_size=50).fit(xy.astype(np.float64)).labels_)
    g = np.load(os.path.join(golden_dir, "g5_clustering.npz"))
    cfg = dict(tree_conf_thresh=0.5, tau_vert=0.6, tau_off=4, tau_group=0.15, tau_min=50, use_hdbscan=True)
    old = cluster.GRID_MIN_POINTS
    cluster.GRID_MIN_POINTS = 0                                # the reference-generated goldens through the grid form as well
    try:
        for case in "ab":
            pred = get_instances(g[f"{case}_coords"], g[f"{case}_offsets"], g[f"{case}_logits"], cfg, g[f"{case}_vert"], 0, 0, -1, 1)
            np.testing.assert_array_equal(pred, g[f"{case}_hdbscan_pred"])
    finally:
        cluster.GRID_MIN_POINTS = old


def test_hdbscan_grid_form_400k_points_under_half_a_second():
    import time
    from treelearn_amd.cluster import hdbscan
    rng = np.random.default_rng(0); n = 400000; k = n // 2500
    c = rng.uniform(0, 100, (k, 2))
    xy = (c[rng.integers(0, k, n)] + rng.normal(0, 0.15, (n, 2))).astype(np.float32)
    xy[: n // 20] = rng.uniform(0, 100, (n // 20, 2))
    hdbscan(xy[:5000], 50, algorithm="grid")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lab = hdbscan(xy, 50)
    dt = time.perf_counter() - t0
    assert lab.max() + 1 >= k * 0.9 and dt < 0.5, (lab.max() + 1, dt)


def test_training_step_mixed_precision(golden_dir):
    """compute_dtype = bf16 in training = the reference's autocast regime (convs in half precision, BatchNorm / loss fp32,
    tools/training/train.py:35-40): loss and gradients stay within bf16 distance of the fp32 golden."""
    from treelearn_amd.model import TreeLearn
    name = "m3"
    g = np.load(os.path.join(golden_dir, "g10_forward.npz"))
    cfg = json.loads(str(g[f"{name}_cfg"]))
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=cfg["spatial_shape"], voxel_size=cfg["voxel_size"],
                      compute_dtype=torch.bfloat16, **cfg["cfg"])
    model.load_state_dict(om.random_state_dict(cfg["seed"], **cfg["cfg"]), strict=True)
    model = model.cuda().train()
    model.zero_grad()
    loss, _ = model(_batch_from_golden(g, name), return_loss=True)
    loss.backward()
    assert float(loss) == pytest.approx(float(g[f"{name}_train_loss"]), rel=3e-2)
    names = [str(s) for s in g[f"{name}_grad_names"]]
    params = dict(model.named_parameters())
    assert all(params[n].grad is not None and params[n].grad.dtype == torch.float32 for n in names)
    ours = np.array([float(params[n].grad.norm()) for n in names]); ref = g[f"{name}_grad_norms"]
    big = ref > 1e-3 * ref.max()
    assert np.abs(ours[big] / ref[big] - 1).max() < 0.15
    gi = model.input_conv[0].weight.grad.cpu().numpy().ravel(); ri = g[f"{name}_grad_input_conv"].ravel()
    assert float(gi @ ri / (np.linalg.norm(gi) * np.linalg.norm(ri))) > 0.95      # deepest gradient path, ~50 bf16 layers (measured 0.979)


@pytest.mark.parametrize("cin,cout,K,n_out", [(32, 32, 27, 17001), (64, 32, 27, 16500), (32, 64, 8, 16400), (64, 32, 8, 20000), (64, 96, 8, 16390),
                                               (96, 64, 8, 16385), (64, 32, 1, 20000), (64, 64, 27, 16500), (32, 32, 27, 300), (4, 32, 27, 5000), (128, 64, 27, 16400), (96, 96, 27, 16390),
                                               (128, 128, 27, 16401), (192, 96, 27, 16402), (96, 128, 8, 16403), (256, 128, 27, 16404),
                                               (160, 160, 27, 6500), (320, 160, 27, 6000), (192, 192, 27, 1200), (64, 128, 27, 16420)])
def test_conv_bf16_no_prologue_multi_output(cin, cout, K, n_out):
    """Pre-activated form: no gather-side prologue, residual, three output views (raw, bn+relu, bn+relu) --
    exercises the weights-in-LDS direct kernel (level-1 shapes), the stream kernel (fragment-shaped gathers),
    the stream-q kernel (quad gathers + register transposition; Cin % 64 == 0) and the small-level kernel."""
    from treelearn_amd import ops
    rng = np.random.default_rng(cin + 3 * cout + K)
    d = _dev()
    n_in = n_out + 77 if K != 1 else n_out
    x = _bf16_round(rng.normal(size=(n_in, cin)).astype(np.float32))
    k = round(K ** (1 / 3))
    w = _bf16_round((rng.normal(size=(cout, k, k, k, cin)) / np.sqrt(cin * K)).astype(np.float32))
    table = rng.integers(-1, n_in, size=(n_out, K)).astype(np.int32)
    table[rng.uniform(size=table.shape) < 0.6] = -1
    if K == 1:
        table = np.arange(n_in, dtype=np.int32)[:, None]
    res = _bf16_round(rng.normal(size=(n_out, cout)).astype(np.float32))
    s2 = rng.uniform(0.5, 1.5, cout).astype(np.float32); h2 = rng.normal(0, 0.3, cout).astype(np.float32)
    s3 = rng.uniform(0.5, 1.5, cout).astype(np.float32); h3 = rng.normal(0, 0.3, cout).astype(np.float32)
    y = osp.conv_table(torch.from_numpy(x), torch.from_numpy(w), table, n_out).numpy() + res
    T = lambda a, dt=torch.float32: torch.from_numpy(a).to(d).to(dt)
    wp = ops.pack_weight(T(w), torch.bfloat16)
    tab = None if K == 1 else torch.from_numpy(np.ascontiguousarray(table.T)).to(d)
    wide = torch.zeros((n_out, 2 * cout), dtype=torch.bfloat16, device=d)               # out2 lands in the right half of a concat buffer
    o3 = torch.empty((n_out, cout), dtype=torch.bfloat16, device=d)
    out = ops.conv_fwd(T(x, torch.bfloat16), wp, tab, n_out, residual=T(res, torch.bfloat16),
                       out2=(wide[:, cout:], T(s2), T(h2), True), out3=(o3, T(s3), T(h3), True))
    assert rel_err(out.float().cpu().numpy(), y) < 8e-3
    assert rel_err(wide[:, cout:].float().cpu().numpy(), np.maximum(y * s2 + h2, 0)) < 8e-3
    assert rel_err(o3.float().cpu().numpy(), np.maximum(y * s3 + h3, 0)) < 8e-3
    assert float(wide[:, :cout].abs().max()) == 0.0


def test_next_rows_ensemble_and_knn_fill_vs_golden(golden_dir):
    """SURVEY.md §8f #1/#2 against the reference's own outputs (goldens G6/G7)."""
    from treelearn_amd.util.postprocess import assign_remaining_points_nearest_neighbor, ensemble
    g = np.load(os.path.join(golden_dir, "g6_g7_next.npz"))
    res = ensemble(g["e_coords"], g["e_sem"], g["e_seml"], g["e_off"], g["e_offl"], g["e_inst"], g["e_feats"], g["e_inf"])
    for i, r in enumerate(res):
        ref = g[f"e_out{i}"]
        assert r.shape == ref.shape and r.dtype == ref.dtype, i
        if r.dtype == np.int64:
            np.testing.assert_array_equal(r, ref)
        else:
            np.testing.assert_allclose(r, ref, rtol=1e-6, atol=1e-6)
    out = assign_remaining_points_nearest_neighbor(g["k_coords"], g["k_pred"], -1)
    np.testing.assert_array_equal(out, g["k_out"])
    # bigger random case against sklearn directly
    from sklearn.neighbors import KNeighborsClassifier
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(20000, 3)).astype(np.float32) * 3
    pred = rng.integers(1, 30, 20000).astype(np.int64); pred[rng.uniform(size=20000) < 0.4] = -1
    ours = assign_remaining_points_nearest_neighbor(pts, pred, -1)
    knn = KNeighborsClassifier(n_neighbors=5).fit(pts[pred != -1], pred[pred != -1])
    ref = pred.copy(); ref[pred == -1] = knn.predict(pts[pred == -1])
    np.testing.assert_array_equal(ours, ref)


def test_next_rows_device_tiler_vs_golden(golden_dir):
    """PlotTiler (tl_tile_crop + host label bookkeeping) reproduces every array of every tile of the
    reference's tile_generate_and_save -> npz -> TreeDataset -> collate chain (golden G11), bit for bit."""
    import hashlib
    from treelearn_amd.util.tiles import PlotTiler
    g = np.load(os.path.join(golden_dir, "g11_tiles.npz"))
    ie, oe, st, isel = (float(v) for v in g["params"])
    tiler = PlotTiler(g["points"], g["labels"], g["feats"])
    keys = [str(k) for k in g["keys"]]
    n = 0
    for i, b in enumerate(tiler.tiles(ie, oe, st, isel)):
        assert b["batch_size"] == 1 and len(b["coords"]) == int(g["counts"][i]), i
        for j, k in enumerate(keys):
            a = b[k].cpu().numpy()
            assert hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest() == str(g["digests"][i][j]), (i, k, a.dtype)
        n += 1
    assert n == int(g["n_tiles"])
    fast = list(tiler.tiles(ie, oe, st, isel, offset_labels="none"))
    assert len(fast) == n and not bool(fast[0]["masks_off"].any()) and fast[0]["coords"].is_cuda


def test_next_rows_propagate_preds_vs_golden(golden_dir):
    """propagate_preds (k-NN label propagation, util/pipeline.py:300-331) against the reference's own output, k = 5 and k = 4
    (even k: ties between equally frequent labels resolve to the smallest label, negatives included)."""
    from treelearn_amd.util.postprocess import propagate_preds
    g = np.load(os.path.join(golden_dir, "g6_g7_next.npz"))
    for k in (5, 4):
        ours = propagate_preds(g["p_src"], g["p_pred"], g["p_tgt"], k)
        assert ours.dtype == np.int64
        np.testing.assert_array_equal(ours, g[f"p_out{k}"])


def test_next_rows_voxel_downsample_vs_oracle():
    """SURVEY.md 8f #4 (parity unpinned by the reference: open3d is absent): the device down-sample equals the numpy
    restatement bit for bit -- voxel set, in-order double means -> float32 -> 2 decimals, first indices, trace."""
    from oracle import prepare as op
    from treelearn_amd.util.prepare import propagate_to_original, voxelize
    rng = np.random.default_rng(4)
    for n, scale, shift in ((5000, 3.0, 0.0), (20000, 12.0, 431.77)):
        pts = rng.normal(size=(n, 3)) * scale + shift
        pts[: n // 4] = np.round(pts[: n // 4], 1) + 0.05                     # points on voxel boundaries after rounding
        lab = rng.integers(-1, 9, size=(n, 1)).astype(np.float64)
        data = np.hstack([pts, lab])
        ref, first, p2v = op.voxelize(data, 0.1)
        out, trace = voxelize(data, 0.1)
        assert out.shape == ref.shape
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
        np.testing.assert_array_equal(trace["first_idx"].cpu().numpy(), first)
        np.testing.assert_array_equal(trace["point2vox"].cpu().numpy(), p2v)
        preds = rng.integers(0, 50, size=len(ref))
        np.testing.assert_array_equal(propagate_to_original(preds, trace).cpu().numpy(), preds[p2v])


def test_next_rows_voxel_downsample_vs_hand_worked_example():
    from kat_cases import voxel_downsample_by_hand
    from treelearn_amd.util.prepare import voxelize
    data, voxel, expect, first, p2v = voxel_downsample_by_hand()
    out, trace = voxelize(data, voxel)
    np.testing.assert_allclose(out.cpu().numpy(), expect, atol=1e-6)
    np.testing.assert_array_equal(trace["first_idx"].cpu().numpy(), first)
    np.testing.assert_array_equal(trace["point2vox"].cpu().numpy(), p2v)


def test_next_rows_verticality_vs_oracle():
    """Verticality = 1 - |n_z| of the radius-neighbourhood covariance vs scipy cKDTree + numpy eigh (fp64), 1e-5 absolute
    wherever the normal is well conditioned; same NaN set (fewer than 3 neighbours) and the same NaN replacement."""
    from oracle import prepare as op
    from treelearn_amd.synth import make_tile
    from treelearn_amd.util.prepare import compute_features
    t = make_tile(extent=12.0, voxel=0.1, n_trees=5, fill=0.10, seed=8)
    pts = t["points"].astype(np.float64)
    pts = np.vstack([pts, np.array([[40.0, 40.0, 3.0], [40.2, 40.0, 3.0], [-50.0, 7.0, 1.0]])])       # isolated points: < 3 neighbours
    ref, gap = op.verticality(pts, 0.6)
    ours = compute_features(pts, search_radius=0.6).cpu().numpy()[:, 0]
    nan = np.isnan(ref)
    assert 3 <= nan.sum() < 50
    good = ~nan & (gap > 1e-6)
    assert good.mean() > 0.99
    assert np.abs(ours[good] - ref[good]).max() < 1e-5
    filled = op.replace_nan(ref)
    assert np.abs(ours[nan] - filled[nan]).max() < 1e-4 and ours.dtype == np.float32


def test_next_rows_verticality_vs_hand_derived_planes():
    """The HIP verticality feature against answers that need no implementation (tests/kat_cases.verticality_planes): 1 - cos(theta)
    on a plane whose normal is theta off the vertical."""
    from kat_cases import verticality_planes
    from treelearn_amd.util.prepare import compute_features
    for pts, expect in verticality_planes():
        v = compute_features(pts, search_radius=0.6).cpu().numpy()[:, 0]
        assert np.abs(v - expect).max() < 1e-5, (expect, float(np.abs(v - expect).max()))


def test_end_to_end_plot_pipeline_runs_and_is_deterministic():
    """All rows of SURVEY 8 chained on one small raw cloud, everything on the device: down-sample + verticality (8f #4) ->
    tiles (8f #3) -> tile loop (a2..a13) -> ensemble (8f #1) -> grouping (a14..a17) -> k-NN fill (8f #2) -> back to the
    original points.  Random weights: the check is plumbing (shapes, label ranges, every original point labelled) and
    run-to-run determinism of the whole chain."""
    from oracle import model as om
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_tile
    from treelearn_amd.util import get_instances, get_pointwise_preds
    from treelearn_amd.util.postprocess import assign_remaining_points_nearest_neighbor, ensemble
    from treelearn_amd.util.prepare import compute_features, propagate_to_original, voxelize
    from treelearn_amd.util.tiles import PlotTiler
    t = make_tile(extent=14.0, voxel=0.1, n_trees=6, fill=0.10, seed=5)
    rng = np.random.default_rng(0)
    raw = np.vstack([t["points"].astype(np.float64) + rng.normal(0, 0.015, size=t["points"].shape) for _ in range(2)])
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1)
    model.load_state_dict(om.random_state_dict(3, channels=32, num_blocks=7)); model = model.cuda().eval()
    cfg = dict(tree_conf_thresh=0.5, tau_vert=0.0, tau_off=1e9, tau_group=0.3, tau_min=20, use_hdbscan=False)

    def run():
        down, trace = voxelize(raw, 0.1)
        feats = compute_features(down[:, :3], 0.6)
        tiler = PlotTiler(down[:, :3].float(), torch.full((len(down),), -1.0), feats)
        res = get_pointwise_preds(model, tiler.tiles(4.0, 5.0, 0.5, 4.0, offset_labels="none"), dict(voxel_size=0.1))
        res = ensemble(res[4], res[0], res[1], res[2], res[3], res[5], res[6], res[7])
        coords, sem, _, off, _, _, _, infeat = res
        inst = get_instances(coords, off, sem, cfg, infeat[:, -1], 0, 0, -1, 1)
        tree = inst != 0
        if tree.any() and (inst[tree] != -1).any():
            inst[tree] = assign_remaining_points_nearest_neighbor(coords[tree] + off[tree], inst[tree], -1)
        return down, trace, coords, inst

    down, trace, coords, inst = run()
    assert len(coords) > 0.5 * len(down) and inst.dtype == np.int64 and inst.min() >= -1
    # the ensembled inner points are voxelised points: map them back onto the down-sampled rows, then onto the raw cloud
    dn = down[:, :3].cpu().numpy().astype(np.float32)
    key = lambda a: np.round(a * 100).astype(np.int64) @ np.array([1 << 40, 1 << 20, 1], np.int64)        # noqa: E731
    order = np.argsort(key(dn)); pos = np.searchsorted(key(dn)[order], key(coords.astype(np.float32)))
    assert (key(dn)[order][np.minimum(pos, len(dn) - 1)] == key(coords.astype(np.float32))).mean() > 0.999
    vox_pred = np.zeros(len(dn), np.int64); vox_pred[order[np.minimum(pos, len(dn) - 1)]] = inst
    full = propagate_to_original(vox_pred, trace).cpu().numpy()
    assert full.shape == (len(raw),)
    down2, _, coords2, inst2 = run()
    np.testing.assert_array_equal(coords, coords2); np.testing.assert_array_equal(inst, inst2)


@pytest.mark.parametrize("cin,cout,n_out", [(64, 32, 20000), (96, 64, 16500), (128, 96, 17000), (160, 128, 3000)])
def test_inverse_conv_one_hot_form(cin, cout, n_out):
    """SparseInverseConv3d rulebooks have one valid entry per output row: the gather-once path (table_one_hot) must give
    the same result as the general path and as the oracle, including rows with no parent at all."""
    from treelearn_amd import ops
    rng = np.random.default_rng(cin + cout)
    d = _dev()
    n_in = n_out // 5 + 3
    x = _bf16_round(rng.normal(size=(n_in, cin)).astype(np.float32))
    w = _bf16_round((rng.normal(size=(cout, 2, 2, 2, cin)) / np.sqrt(cin)).astype(np.float32))
    table = np.full((n_out, 8), -1, np.int32)
    table[np.arange(n_out), rng.integers(0, 8, n_out)] = rng.integers(0, n_in, n_out)
    table[rng.uniform(size=n_out) < 0.02] = -1                                        # orphans
    y = osp.conv_table(torch.from_numpy(x), torch.from_numpy(w), table, n_out).numpy()
    T = lambda a, dt=torch.float32: torch.from_numpy(a).to(d).to(dt)
    wp = ops.pack_weight(T(w), torch.bfloat16)
    tab = torch.from_numpy(np.ascontiguousarray(table.T)).to(d)
    s2 = rng.uniform(0.5, 1.5, cout).astype(np.float32); h2 = rng.normal(0, 0.3, cout).astype(np.float32)
    o2a = torch.empty((n_out, cout), dtype=torch.bfloat16, device=d); o2b = torch.empty_like(o2a)
    a = ops.conv_fwd(T(x, torch.bfloat16), wp, tab, n_out, out2=(o2a, T(s2), T(h2), True), one_hot=True)
    b = ops.conv_fwd(T(x, torch.bfloat16), wp, tab, n_out, out2=(o2b, T(s2), T(h2), True), one_hot=False)
    assert rel_err(a.float().cpu().numpy(), y) < 8e-3
    assert torch.equal(a, b) and torch.equal(o2a, o2b)


@pytest.mark.parametrize("cin,cout,K,n_out", [(32, 32, 27, 9000), (64, 32, 27, 5000), (4, 32, 27, 7000), (96, 64, 8, 4100), (160, 192, 8, 700),
                                               (64, 32, 1, 6000), (224, 224, 27, 223), (96, 96, 27, 3000), (192, 96, 27, 1500),
                                               (96, 96, 27, 120000), (192, 96, 27, 100100),       # big levels: the dense-over-taps form (tl_wgrad_dense.hip)
                                               (32, 2, 1, 50000), (32, 3, 1, 70001), (64, 4, 1, 3000),     # the heads' output Linears (tl_linear_small.hip)
                                               (32, 32, 1, 100003), (64, 32, 1, 70000), (128, 64, 1, 40001), (192, 96, 1, 33000), (256, 128, 1, 31000),   # tl_wgrad_rows.hip
                                               (4, 32, 27, 50001), (128, 128, 27, 60500), (256, 128, 27, 60100), (32, 32, 27, 70001), (64, 32, 27, 65000), (64, 64, 27, 99999), (128, 64, 27, 61000),
                                               (96, 96, 27, 50000), (192, 96, 27, 40000)])        # multiples of 96 below the dense form's row threshold: 96 x 96 pair-list blocks
def test_conv_wgrad_vs_dense_reference(cin, cout, K, n_out):
    """tl_conv_wgrad (present pairs only, fp32 MFMA, deterministic) vs gather + matmul in float64."""
    from treelearn_amd import ops
    rng = np.random.default_rng(cin * 7 + cout + K)
    d = _dev()
    n_in = n_out + 50 if K != 1 else n_out
    x = rng.normal(size=(n_in, cin)).astype(np.float32); g = rng.normal(size=(n_out, cout)).astype(np.float32)
    table = rng.integers(-1, n_in, size=(n_out, K)).astype(np.int32)
    table[rng.uniform(size=table.shape) < 0.7] = -1
    if K == 1:
        table = np.arange(n_in, dtype=np.int32)[:, None]
    ref = np.zeros((K, cout, cin))
    for k in range(K):
        m = table[:, k] >= 0
        ref[k] = g[m].astype(np.float64).T @ x[table[m, k]].astype(np.float64)
    tab = None if K == 1 else torch.from_numpy(np.ascontiguousarray(table.T)).to(d)
    wide = torch.zeros((n_in, cin + 8), dtype=torch.float32, device=d); wide[:, 8:] = torch.from_numpy(x).to(d)      # column view input
    a = ops.conv_wgrad(wide[:, 8:], torch.from_numpy(g).to(d), tab, n_out, K)
    b = ops.conv_wgrad(wide[:, 8:], torch.from_numpy(g).to(d), tab, n_out, K)
    assert torch.equal(a, b)                                                        # deterministic
    assert rel_err(a.cpu().numpy(), ref) < 2e-5
    # bf16 inputs (mixed-precision training) on the bf16 matrix cores: exact products, fp32 sums -> same as the fp32 kernel fed with the
    # bf16-rounded values up to the summation order; deterministic
    xb = wide[:, 8:].to(torch.bfloat16); gb = torch.from_numpy(g).to(d).to(torch.bfloat16)
    c = ops.conv_wgrad(xb, gb, tab, n_out, K)
    c2 = ops.conv_wgrad(xb, gb, tab, n_out, K)
    e = ops.conv_wgrad(xb.float(), gb.float(), tab, n_out, K)
    assert torch.equal(c, c2)
    if K == 27 and n_out >= 60000:                                                   # the dense-over-taps kernels wait on COUNTED outstanding loads: a wrong
        for _ in range(12):                                                          # count shows as an occasional stale tile, not as a wrong mean
            assert torch.equal(ops.conv_wgrad(xb, gb, tab, n_out, K), c)
    assert rel_err(c.cpu().numpy(), e.cpu().numpy()) < 2e-5
    if K == 27 and n_out >= 60000 and cin >= 32:                                     # dense-over-taps: the register-staged form of the same kernel
        from treelearn_amd import _hip
        _hip.lib().tl_set_tuning(b"wgrad_dma", 0)
        try:
            c3 = ops.conv_wgrad(xb, gb, tab, n_out, K)
        finally:
            _hip.lib().tl_set_tuning(b"wgrad_dma", 1)
        assert rel_err(c3.cpu().numpy(), c.cpu().numpy()) < 2e-5
        if (cin, cout) in ((32, 32), (96, 96), (192, 96), (128, 128), (256, 128)):   # same slot count in both forms -> same summation order
            assert torch.equal(c3, c)
    if cin % 4 == 0:                                                                 # the parameter's own layout [Cout, K, Cin] straight from the reduction
        assert torch.equal(ops.conv_wgrad(xb, gb, tab, n_out, K, ref_layout=True), c.permute(1, 0, 2).contiguous())
        assert torch.equal(ops.conv_wgrad(wide[:, 8:], torch.from_numpy(g).to(d), tab, n_out, K, ref_layout=True), a.permute(1, 0, 2).contiguous())


def test_compact_rulebook_equals_table():
    """Column form of the level-1 rulebook (tl_rulebook_compact, 9 bases + presence mask): the direct kernel fed with it
    must reproduce the table-fed result bit for bit on a real geometry (incl. tile-boundary rows and a ragged last tile)."""
    from treelearn_amd import ops
    from treelearn_amd.geometry import build_geometry
    from treelearn_amd.synth import make_tile
    t = make_tile(extent=16.0, voxel=0.1, n_trees=8, fill=0.10, seed=2)
    pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
    g = build_geometry(pts, bid, 1, 0.1, 7, [500, 500, 1000])
    lv = g.levels[0]
    ct = getattr(lv.nbr, "_tl_compact", None)
    assert ct is not None and ct.shape == (10, lv.n)
    nbr = lv.nbr.cpu().numpy(); c = ct.cpu().numpy()
    mask = c[9].astype(np.uint32)
    for k in range(27):
        col = k // 3
        below = (mask >> np.uint32(3 * col)) & np.uint32((1 << (k % 3)) - 1)
        dec = np.where((mask >> np.uint32(k)) & 1, c[col] + np.array([bin(int(v)).count("1") for v in np.unique(below)])[np.searchsorted(np.unique(below), below)], -1)
        np.testing.assert_array_equal(dec, nbr[k])
    for cin in (32, 64):
        x = torch.randn(lv.n, cin, device="cuda").to(torch.bfloat16)
        w = ops.pack_weight(torch.randn(32, 3, 3, 3, cin, device="cuda") * 0.05, torch.bfloat16)
        res = torch.randn(lv.n, 32, device="cuda").to(torch.bfloat16)
        a = ops.conv_fwd(x, w, lv.nbr, lv.n, residual=res)
        plain = lv.nbr.clone()                                   # same table without the attached column form
        b = ops.conv_fwd(x, w, plain, lv.n, residual=res)
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_column_form_rulebook_in_the_mid_level_gather_kernels(dtype):
    """Levels 2 and 3 of a real tile (>= 65 536 rows: the geometry attaches the column form there too): the stream-q / stream kernels fed
    with the 40-B column form must give the table-fed result bit for bit -- 64 -> 64, 128 -> 64 (two channel slices), 96 -> 96, residual
    and a second view; fp32 (the stream kernel's exact mode) as well."""
    from treelearn_amd import ops
    from treelearn_amd.geometry import COMPACT_MIN_ROWS, build_geometry
    from treelearn_amd.synth import make_tile
    t = make_tile(extent=30.0, voxel=0.1, n_trees=30, fill=0.10, seed=3)
    pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
    g = build_geometry(pts, bid, 1, 0.1, 7, [500, 500, 1000])
    served = 0
    for li, shapes in ((1, [(64, 64), (128, 64)]), (2, [(96, 96)])):
        lv = g.levels[li]
        assert lv.n >= COMPACT_MIN_ROWS and getattr(lv.nbr, "_tl_compact", None) is not None, (li, lv.n)
        plain = lv.nbr.clone()                                   # the same table without the attached column form
        for cin, cout in shapes:
            x = torch.randn(lv.n, cin, device="cuda").to(dtype)
            w = ops.pack_weight(torch.randn(cout, 3, 3, 3, cin, device="cuda") * 0.05, dtype)
            res = torch.randn(lv.n, cout, device="cuda").to(dtype)
            sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
            outs = []
            for tab in (lv.nbr, plain):
                o2 = torch.empty(lv.n, cout, device="cuda", dtype=dtype)
                outs.append((ops.conv_fwd(x, w, tab, lv.n, residual=res, out2=(o2, sc, sh, True)), o2))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (li, cin, cout)
            served += 1
    assert served == 3


def test_input_conv_of_all_ones_by_presence_mask_table():
    """tl_conv_args.in_all_ones (the reference's default use_feats = False, use_coords = False feeds ones, tree_learn.py:129-167): the
    27-entry table indexed by the rulebook's presence mask equals the gather kernel on an all-ones input (fp32 sums in another order,
    one bf16 rounding) and float64, with the BatchNorm + ReLU second view."""
    from treelearn_amd import ops
    from treelearn_amd.geometry import build_geometry
    from treelearn_amd.synth import make_tile
    t = make_tile(extent=16.0, voxel=0.1, n_trees=8, fill=0.10, seed=4)
    pts = torch.from_numpy(t["points"]).cuda(); bid = torch.zeros(len(pts), dtype=torch.int64, device="cuda")
    lv = build_geometry(pts, bid, 1, 0.1, 7, [500, 500, 1000]).levels[0]
    assert getattr(lv.nbr, "_tl_compact", None) is not None
    x = torch.ones(lv.n, 4, device="cuda", dtype=torch.bfloat16)
    wr = torch.randn(32, 3, 3, 3, 4, device="cuda") * 0.2
    w = ops.pack_weight(wr, torch.bfloat16)
    sc = torch.rand(32, device="cuda") + 0.5; sh = torch.randn(32, device="cuda") * 0.2
    o2a = torch.empty(lv.n, 32, device="cuda", dtype=torch.bfloat16); o2b = torch.empty_like(o2a)
    a = ops.conv_fwd(x, w, lv.nbr, lv.n, out2=(o2a, sc, sh, True), all_ones=True)
    b = ops.conv_fwd(x, w, lv.nbr, lv.n, out2=(o2b, sc, sh, True))
    present = (lv.nbr >= 0).double().T                                       # [n, 27]
    ref = present @ w.double().sum(-1)                                       # sum over taps of sum_c W[k][co][c]  -> [n, 32]
    assert float((a.double() - ref).abs().max()) <= 8e-3 * float(ref.abs().max())
    assert float((a.float() - b.float()).abs().max()) <= 8e-3 * float(ref.abs().max())
    ref2 = torch.relu(ref * sc.double() + sh.double())
    assert float((o2a.double() - ref2).abs().max()) <= 8e-3 * float(ref2.abs().max())


@pytest.mark.parametrize("extent,shape", [(6.0, None), (21.0, [500, 500, 1000]), (9.3, None)])
def test_pyramid_calls_equal_per_level_calls(extent, shape):
    """tl_pyramid_build + tl_rulebooks_build (what build_geometry uses: deep levels built by one workgroup / shared launches, one
    backing block) against the per-level entry points of the same library, array by array, incl. odd extents whose last
    cells are dropped by the k2s2 output shape (parent = -1 rows)."""
    from treelearn_amd import _hip
    from treelearn_amd.geometry import build_geometry, level_shapes, _nwords
    from treelearn_amd.synth import make_tile
    L = _hip.lib(); st = _hip.stream()
    t = make_tile(extent=extent, voxel=0.1, n_trees=5, fill=0.08, seed=11)
    pts = torch.from_numpy(t["points"]).cuda(); N = len(pts)
    bid = torch.zeros(N, dtype=torch.int64, device="cuda")
    nl = 7 if extent > 7 else 4
    g = build_geometry(pts, bid, 1, 0.1, nl, shape)
    i32 = lambda *sh: torch.empty(sh, dtype=torch.int32, device="cuda")             # noqa: E731
    shapes = level_shapes(shape if shape is not None else g.levels[0].dims[1:], nl)
    bms, pfs, ns = [], [], []
    for li, lv in enumerate(g.levels):
        nw = _nwords(lv.dims)
        bm = torch.empty(nw, dtype=torch.int64, device="cuda"); pf = i32(nw); cnt = i32(1); ws = i32(int(L.tl_scan_ws_words(nw)))
        if li == 0:
            _hip.check(L.tl_bitmap_from_points(_hip.ptr(g.pcoords), N, _hip.dims4(lv.dims), _hip.ptr(bm), st), "bitmap")
        else:
            _hip.check(L.tl_bitmap_down(_hip.ptr(bms[-1]), _hip.dims4(g.levels[li - 1].dims), _hip.dims3(shapes[li]), _hip.ptr(bm),
                                        _hip.dims4(lv.dims), st), "down")
        _hip.check(L.tl_bitmap_scan(_hip.ptr(bm), nw, _hip.ptr(pf), _hip.ptr(cnt), _hip.ptr(ws), st), "scan")
        bms.append(bm); pfs.append(pf); ns.append(int(cnt.item()))
        assert torch.equal(bm, lv.bitmap) and torch.equal(pf, lv.prefix) and ns[-1] == lv.n, f"level {li}"
    coords = []
    for li, lv in enumerate(g.levels):
        c = i32(lv.n, 4); nbr = i32(27, lv.n); ct = i32(10, lv.n)
        _hip.check(L.tl_expand_coords(_hip.ptr(bms[li]), _hip.ptr(pfs[li]), _hip.dims4(lv.dims), _hip.ptr(c), st), "expand")
        _hip.check(L.tl_rulebook_subm(_hip.ptr(c), lv.n, _hip.ptr(bms[li]), _hip.ptr(pfs[li]), _hip.dims4(lv.dims), _hip.ptr(nbr),
                                      _hip.ptr(ct), st), "subm")
        coords.append(c)
        assert torch.equal(c, lv.coords) and torch.equal(nbr, lv.nbr), f"level {li}"
        ct2 = i32(10, lv.n)
        _hip.check(L.tl_rulebook_compact(_hip.ptr(nbr), lv.n, _hip.ptr(ct2), st), "compact")
        assert torch.equal(ct, ct2)
        if getattr(lv.nbr, "_tl_compact", None) is not None:
            assert torch.equal(lv.nbr._tl_compact, ct)
    dropped = 0
    for li in range(nl - 1):
        f, c = g.levels[li], g.levels[li + 1]
        child = i32(8, c.n); parent = i32(f.n); inv = i32(8, f.n)
        _hip.check(L.tl_rulebook_down(_hip.ptr(coords[li + 1]), c.n, _hip.ptr(bms[li]), _hip.ptr(pfs[li]), _hip.dims4(f.dims), f.n,
                                      _hip.ptr(child), _hip.ptr(parent), _hip.ptr(inv), st), "rb down")
        assert torch.equal(child, f.child) and torch.equal(parent, f.parent) and torch.equal(inv, f.inv), f"level {li}"
        dropped += int((parent < 0).sum())
    if shape is None and any(d % 2 for d in g.levels[0].shape):
        assert dropped > 0                                         # an odd extent drops the cells at its last index (parent = -1)
    v2p = torch.empty(N, dtype=torch.int64, device="cuda")
    _hip.check(L.tl_point_rank(_hip.ptr(g.pcoords), N, _hip.ptr(bms[0]), _hip.ptr(pfs[0]), _hip.dims4(g.levels[0].dims), _hip.ptr(v2p), st), "rank")
    assert torch.equal(v2p, g.v2p)


def test_kernel_families_agree_on_a_real_tile():
    """Size-independent check at realistic scale (24 x 24 m tile, 0.7 M points, all seven levels populated): the bf16 forward
    must not depend on WHICH kernel family serves a layer.  Default dispatch (direct / stream-q / stream / small, column-form
    rulebook, gather-once inverse convs) vs the fallbacks forced through tl_set_tuning (no direct, no stream-q, no stream ->
    tile kernel; 4-wave small kernel with row-major weights; plain tables): same outputs up to bf16 re-association."""
    from oracle import model as om
    from treelearn_amd import _hip
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile
    b = make_batch([make_tile(extent=24.0, voxel=0.1, n_trees=20, fill=0.10, seed=4)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
    model.load_state_dict(om.random_state_dict(11, channels=32, num_blocks=7)); model = model.cuda().eval()
    L = _hip.lib()

    def run(**tuning):
        for k, v in tuning.items():
            _hip.check(L.tl_set_tuning(k.encode(), v), k)
        try:
            with torch.no_grad():
                o = model(gb, return_loss=False)
            return {k: v.float().cpu().numpy() for k, v in o.items()}
        finally:
            for k in tuning:
                _hip.check(L.tl_set_tuning(k.encode(), {"small_mode": 0}.get(k, 1)), k)

    ref = run()
    os.environ["TL_NO_COMPACT"] = "1"
    try:
        variants = [run(streamq=0), run(direct=0), run(direct=0, stream=0), run(small_mode=3)]
    finally:
        os.environ.pop("TL_NO_COMPACT")
    for i, v in enumerate(variants):
        for k in ("semantic_prediction_logits", "offset_predictions", "backbone_feats"):
            assert rel_err(v[k], ref[k]) < 5e-2, (i, k, rel_err(v[k], ref[k]))     # random weights amplify bf16 re-association 2-4 % (max over 0.66 M points)
    # and the bf16 result stays within bf16 distance of the exact-fp32 mode on the same tile
    m32 = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1)
    m32.load_state_dict(model.state_dict()); m32 = m32.cuda().eval()
    with torch.no_grad():
        o32 = m32(gb, return_loss=False)
    for k in ("semantic_prediction_logits", "offset_predictions"):
        e32 = o32[k].float().cpu().numpy()
        assert rel_err(ref[k], e32) < 6e-2, k
        for i, v in enumerate(variants):
            assert rel_err(v[k], e32) < 6e-2, (i, k)


@pytest.mark.parametrize("seed", [101, 102, 103, 104])
def test_forward_fuzz_small_tiles_vs_oracle(seed):
    """Ragged inputs: random small tiles (odd point counts, thin slabs, a batch of two very unequal elements, voxel sizes 0.1-0.3)
    through the default model in fp32 against the oracle; a tile that collapses must raise the same "reach zero!!!" error in both."""
    from treelearn_amd.model import TreeLearn
    rng = np.random.default_rng(seed)
    voxel = float(rng.choice([0.1, 0.2, 0.3]))
    tiles = []
    for _ in range(int(rng.integers(1, 3))):
        t = make_tile(extent=float(rng.uniform(4, 9)), voxel=voxel, n_trees=int(rng.integers(1, 4)), fill=0.1, seed=int(rng.integers(1 << 30)))
        keep = rng.uniform(size=len(t["points"])) < rng.uniform(0.2, 1.0)
        keep[: 5] = True
        tiles.append({k: (v[keep] if k != "center" else v) for k, v in t.items()})
    batch = make_batch(tiles)
    sd = om.random_state_dict(seed, channels=32, num_blocks=7)
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=voxel)
    model.load_state_dict(sd, strict=True); model = model.cuda().eval()
    args = (sd, batch["coords"].numpy(), batch["input_feats"].numpy(), batch["batch_ids"].numpy(), len(tiles))
    try:
        ref = om.forward(*args, voxel_size=voxel, num_blocks=7, spatial_shape=[500, 500, 1000])
    except ValueError as e:
        assert "reach zero!!!" in str(e)
        with pytest.raises(ValueError, match="reach zero!!!"), torch.no_grad():
            model(batch, return_loss=False)
        return
    with torch.no_grad():
        out = model(batch, return_loss=False)
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        assert np.isfinite(out[k].cpu().numpy()).all()
        assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < REL_TOL, (k, len(batch["coords"]))


def test_handwritten_known_answers_through_the_hip_path():
    """The hand-written spconv known-answer vectors (tests/golden/kat_spconv_handwritten.json; pencil arithmetic, no repo code)
    through the real HIP chain: voxel hashing -> rulebooks -> tl_conv_fwd, fp32 exact and bf16 within its rounding."""
    from kat_cases import build_weight, case_points, load_cases
    from treelearn_amd import ops
    d = _dev()
    for case in load_cases():
        pts, bids = case_points(case)
        B = int(bids.max()) + 1
        g = _geom(pts, bids, B, 1.0, 1 if case["kind"] == "subm" else 2, list(case["shape"]))
        w = torch.from_numpy(build_weight(case)).to(d)
        lv = g.levels[0]
        for dt, tol in ((torch.float32, 0.0), (torch.bfloat16, 8e-3)):
            wp = ops.pack_weight(w, dt)
            if case["kind"] == "subm":
                out = ops.conv_fwd(torch.tensor(case["feats"], device=d).to(dt), wp, lv.nbr, lv.n)
            elif case["kind"] == "down":
                np.testing.assert_array_equal(g.levels[1].coords.cpu().numpy(), np.asarray(case["expect_coords"]), err_msg=case["name"])
                out = ops.conv_fwd(torch.tensor(case["feats"], device=d).to(dt), wp, lv.child, g.levels[1].n)
            else:
                out = ops.conv_fwd(torch.tensor(case["coarse_feats"], device=d).to(dt), wp, lv.inv, lv.n, one_hot=True)
            got = out.float().cpu().numpy(); want = np.asarray(case["expect"], np.float32)
            if tol == 0.0:
                np.testing.assert_array_equal(got, want, err_msg=case["name"])
            else:
                assert np.abs(got - want).max() <= tol * np.abs(want).max(), (case["name"], got, want)


@pytest.mark.parametrize("cin,cout,n_out,kind", [(64, 64, 70001, "shifted"), (64, 64, 3000, "random"), (128, 64, 66000, "mixed"), (96, 96, 66013, "shifted"),
                                                 (192, 96, 5000, "mixed"), (128, 128, 66100, "shifted"), (256, 128, 2100, "random"), (32, 32, 67000, "mixed"),
                                                 (64, 32, 1025, "shifted")])
def test_conv_window_kernel_vs_oracle(cin, cout, n_out, kind):
    """tl_conv_win (dz taps of a column served from one LDS-staged row window) against the oracle: 'shifted' tables keep every
    neighbour near its output row (the in-window path, as on real rulebooks), 'random' tables put every neighbour outside the
    window (the global-memory slow path), 'mixed' interleaves both and leaves whole (row, group) blocks absent; ragged row counts,
    residual and three output views; same result from the 4-wave / 256-row and the 8-wave / 512-row form."""
    from treelearn_amd import _hip as _h
    if _h.lib().tl_set_tuning(b"win", 0) != 0:
        pytest.skip("the window conv kernel is in the developer build only (python -m treelearn_amd.build --dev)")
    from treelearn_amd import _hip, ops
    rng = np.random.default_rng(cin + 3 * cout + n_out)
    d = _dev()
    n_in = n_out + 300
    x = _bf16_round(rng.normal(size=(n_in, cin)).astype(np.float32))
    w = _bf16_round((rng.normal(size=(cout, 3, 3, 3, cin)) / np.sqrt(cin * 27)).astype(np.float32))
    rows = np.arange(n_out)[:, None]
    shifted = rows + rng.integers(-40, 160, size=(1, 27)) + rng.integers(-2, 3, size=(n_out, 27))
    shifted = np.clip(shifted, 0, n_in - 1)
    rnd = rng.integers(0, n_in, size=(n_out, 27))
    if kind == "shifted":
        table = shifted
    elif kind == "random":
        table = rnd
    else:
        table = np.where(rng.uniform(size=(n_out, 27)) < 0.03, rnd, shifted)
        table[(np.arange(n_out) // 700) % 3 == 1, 9:18] = -1
    table = table.astype(np.int32)
    table[rng.uniform(size=table.shape) < 0.4] = -1
    res = _bf16_round(rng.normal(size=(n_out, cout)).astype(np.float32))
    s2 = rng.uniform(0.5, 1.5, cout).astype(np.float32); h2 = rng.normal(0, 0.3, cout).astype(np.float32)
    y = osp.conv_table(torch.from_numpy(x), torch.from_numpy(w), table, n_out).numpy() + res
    T = lambda a, dt=torch.float32: torch.from_numpy(a).to(d).to(dt)
    wp = ops.pack_weight(T(w), torch.bfloat16)
    tab = torch.from_numpy(np.ascontiguousarray(table.T)).to(d)
    L = _hip.lib()
    outs = {}
    try:
        _hip.check(L.tl_set_tuning(b"win", 2), "win"); _hip.check(L.tl_set_tuning(b"win_min_rows", 0), "win_min_rows")
        for wr in (0, 512):
            _hip.check(L.tl_set_tuning(b"win_rows", wr), "win_rows")
            wide = torch.zeros((n_out, 2 * cout), dtype=torch.bfloat16, device=d)
            o3 = torch.empty((n_out, cout), dtype=torch.bfloat16, device=d)
            out = ops.conv_fwd(T(x, torch.bfloat16), wp, tab, n_out, residual=T(res, torch.bfloat16),
                               out2=(wide[:, cout:], T(s2), T(h2), True), out3=(o3, None, None, False))
            assert rel_err(out.float().cpu().numpy(), y) < 8e-3, (wr, rel_err(out.float().cpu().numpy(), y))
            assert rel_err(wide[:, cout:].float().cpu().numpy(), np.maximum(y * s2 + h2, 0)) < 8e-3
            assert torch.equal(o3, out) and float(wide[:, :cout].abs().max()) == 0.0
            outs[wr] = out
        assert torch.equal(outs[0], outs[512])                   # tile / window size changes where rows come from, never the arithmetic
        _hip.check(L.tl_set_tuning(b"win", 0), "win")
        ref = ops.conv_fwd(T(x, torch.bfloat16), wp, tab, n_out, residual=T(res, torch.bfloat16))
        assert rel_err(outs[0].float().cpu().numpy(), ref.float().cpu().numpy()) < 8e-3
    finally:
        _hip.check(L.tl_set_tuning(b"win", 1 if _hip.WIN_KERNEL else 0), "win"); _hip.check(L.tl_set_tuning(b"win_min_rows", 65536), "win_min_rows")
        _hip.check(L.tl_set_tuning(b"win_rows", 0), "win_rows")


@pytest.mark.parametrize("n,C,relu", [(5, 32, True), (1000, 32, True), (70001, 64, True), (33333, 448, True), (4097, 96, False), (257, 224, True)])
def test_bn_train_kernels_vs_torch(n, C, relu):
    """tl_bn_train_stats / tl_affine_relu / tl_bn_train_bwd (the training-mode BatchNorm1d + ReLU pairs of blocks.py:55-70) against
    torch.nn.BatchNorm1d + ReLU in float64: outputs, running statistics, and all three gradients; bit-reproducible run to run."""
    from treelearn_amd.autograd import bn_relu_train
    d = _dev()
    gen = torch.Generator(device="cuda"); gen.manual_seed(n + C)
    x = (torch.randn((n, C), device=d, generator=gen) * 2.0 + torch.randn(C, device=d, generator=gen)).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C, eps=1e-4, momentum=0.1).to(d).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, device=d, generator=gen) + 0.5); bn.bias.copy_(torch.randn(C, device=d, generator=gen) * 0.3)
        bn.running_mean.copy_(torch.randn(C, device=d, generator=gen)); bn.running_var.copy_(torch.rand(C, device=d, generator=gen) + 0.5)
    ref = torch.nn.BatchNorm1d(C, eps=1e-4, momentum=0.1).to(d).double().train()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    gy = torch.randn((n, C), device=d, generator=gen)
    y = bn_relu_train(x, bn, relu)
    y.backward(gy)
    xr = x.detach().double().requires_grad_(True)
    # ReLU as a fixed mask taken from OUR output: the kink is a discontinuity of the gradient, and an fp32 y within rounding of
    # zero may fall on the other side than the fp64 reference's (about one element in 10^7 -- enough to move a gradient sum)
    yl = ref(xr)
    if relu:
        assert float((torch.relu(yl) - yl * (y.detach() > 0)).abs().max()) < 1e-5 * float(yl.abs().max())     # masks differ only at the kink
        yr = yl * (y.detach() > 0).double()
    else:
        yr = yl
    yr.backward(gy.double())
    tol = lambda a: 2e-5 * max(float(a.abs().max()), 1e-6)                          # noqa: E731
    assert float((y.double() - yr).abs().max()) < tol(yr)
    assert float((x.grad.double() - xr.grad).abs().max()) < 5 * tol(xr.grad)
    assert float((bn.weight.grad.double() - ref.weight.grad).abs().max()) < 5 * tol(ref.weight.grad)
    assert float((bn.bias.grad.double() - ref.bias.grad).abs().max()) < 5 * tol(ref.bias.grad)
    assert float((bn.running_mean.double() - ref.running_mean).abs().max()) < 1e-5
    assert float((bn.running_var.double() - ref.running_var).abs().max()) < 1e-4 * float(ref.running_var.max())
    assert int(bn.num_batches_tracked) == 1
    x2 = x.detach().clone().requires_grad_(True)
    y2 = bn_relu_train(x2, bn, relu); y2.backward(gy)
    assert torch.equal(y, y2) and torch.equal(x.grad, x2.grad)


def test_forward_unusual_width_falls_back_to_generic_kernels():
    """channels = 48 (not a multiple of 32, no fused-head instantiation): the reference accepts any width, so the eval forward
    must too -- generic conv kernel + the head fallback -- and still match the oracle."""
    from treelearn_amd.model import TreeLearn
    cfg = dict(channels=48, num_blocks=3)
    t = make_tile(extent=6, voxel=0.2, n_trees=2, seed=5)
    batch = make_batch([t])
    sd = om.random_state_dict(19, **cfg)
    model = TreeLearn(use_feats=False, use_coords=False, voxel_size=0.2, **cfg)
    model.load_state_dict(sd, strict=True); model = model.cuda().eval()
    with torch.no_grad():
        out = model(batch, return_loss=False)
    ref = om.forward(sd, batch["coords"].numpy(), batch["input_feats"].numpy(), batch["batch_ids"].numpy(), 1, voxel_size=0.2, num_blocks=3)
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < REL_TOL, k


@pytest.mark.parametrize("kind,nr,nq,k", [("uniform", 50_000, 20_000, 5), ("clustered", 200_000, 30_000, 5), ("ties", 30_000, 10_000, 3), ("flat", 40_000, 5_000, 1),
                                           ("clustered", 6_000, 2_000, 8)])
def test_knn_vote_grid_equals_brute_force(kind, nr, nq, k):
    """tl_knn_vote_grid (cell grid, ring search) returns exactly what the brute-force tl_knn_vote returns: same k nearest by
    (distance, reference index), same vote -- on uniform points, on trunk-like dense clusters with queries far outside the box, on
    duplicated points (distance ties between different labels) and on a degenerate flat cloud."""
    from treelearn_amd.util.postprocess import knn_vote
    rng = np.random.default_rng(nr + nq + k)
    if kind == "uniform":
        ref = rng.uniform(-20, 20, size=(nr, 3)); qry = rng.uniform(-22, 22, size=(nq, 3))
    elif kind == "clustered":
        c = rng.uniform(-30, 30, size=(40, 3)); c[:, 2] *= 0.1
        ref = c[rng.integers(0, 40, nr)] + rng.normal(size=(nr, 3)) * rng.choice([0.02, 0.1, 0.5], size=(nr, 1))
        qry = np.concatenate([c[rng.integers(0, 40, nq - 500)] + rng.normal(size=(nq - 500, 3)) * 0.3, rng.uniform(-80, 80, size=(500, 3))])
    elif kind == "ties":
        base = np.round(rng.uniform(-5, 5, size=(nr // 3, 3)), 1)
        ref = np.concatenate([base, base, base]); qry = np.round(rng.uniform(-5, 5, size=(nq, 3)), 1)
    else:
        ref = rng.uniform(-20, 20, size=(nr, 3)); ref[:, 2] = 1.25; qry = rng.uniform(-20, 20, size=(nq, 3)); qry[:, 2] = 1.25
    lab = rng.integers(1, 30, size=len(ref))
    d = _dev()
    R = torch.from_numpy(ref.astype(np.float32)).to(d); Lb = torch.from_numpy(lab.astype(np.int64)).to(d); Q = torch.from_numpy(qry.astype(np.float32)).to(d)
    a = knn_vote(R, Lb, Q, k, force="brute")
    b = knn_vote(R, Lb, Q, k, force="grid")
    assert torch.equal(a, b), int((a != b).sum())


def test_point_to_voxel_shim_runs_the_reference_voxelize_golden_g3(golden_dir):
    """spconv_compat.PointToVoxel under the reference's own `voxelize` post-processing (tree_learn/model/tree_learn.py:129-167, restated
    here line by line) on the inputs of golden G3: every output of the reference function is reproduced, for the four use_coords /
    use_feats settings."""
    from treelearn_amd.spconv_compat import PointToVoxel
    g = np.load(os.path.join(golden_dir, "g3_voxelize.npz"))
    dev = _dev()
    feats = torch.from_numpy(np.concatenate([g["coords"], g["input_feats"]], 1)).to(dev)
    batch_ids = torch.from_numpy(g["batch_ids"]).to(dev)

    def voxelize(feats, batch_ids, batch_size, voxel_size, use_coords, use_feats, max_num_points_per_voxel, epsilon=1):
        voxel_coords, voxel_feats, v2p_maps = [], [], []
        total = 0
        for i in range(batch_size):
            one = feats[batch_ids == i]
            lo = torch.min(one[:, :3], dim=0).values
            hi = torch.max(one[:, :3], dim=0).values + epsilon
            vx = PointToVoxel(vsize_xyz=[voxel_size] * 3, coors_range_xyz=lo.tolist() + hi.tolist(), num_point_features=feats.shape[1],
                              max_num_voxels=len(feats), max_num_points_per_voxel=max_num_points_per_voxel, device=feats.device)
            vf, vc, num, v2p = vx.generate_voxel_with_id(one)
            assert vf.dtype == torch.float32 and vc.dtype == torch.int32 and num.dtype == torch.int32 and v2p.dtype == torch.int64
            assert int((v2p == -1).sum()) == 0 and int(num.max()) <= max_num_points_per_voxel and int(num.min()) >= 1
            vc = vc.float()
            vc[:, [0, 2]] = vc[:, [2, 0]]                                   # zyx -> xyz
            vc = torch.cat((torch.ones((len(vc), 1), device=feats.device) * i, vc), dim=1)
            zero_rows = torch.sum(vf == 0, dim=2) == vf.shape[2]
            vf[zero_rows] = float("nan")
            vf = torch.nanmean(vf, dim=1)
            if not use_coords:
                vf[:, :3] = torch.ones_like(vf[:, :3])
            if not use_feats:
                vf[:, 3:] = torch.ones_like(vf[:, 3:])
            vf = torch.hstack([vf[:, 3:], vf[:, :3]])
            voxel_coords.append(vc); voxel_feats.append(vf); v2p_maps.append(v2p + total)
            total += len(vc)
        vc = torch.cat(voxel_coords); vf = torch.cat(voxel_feats); v2p = torch.cat(v2p_maps)
        return vf, vc, v2p, (vc.max(dim=0).values + 1)[1:]

    for uc in (False, True):
        for uf in (False, True):
            vf, vc, v2p, ss = voxelize(feats, batch_ids, 2, 0.2, uc, uf, 3)
            tag = f"c{int(uc)}f{int(uf)}"
            np.testing.assert_array_equal(vc.cpu().numpy(), g[f"{tag}_voxel_coords"])
            np.testing.assert_array_equal(v2p.cpu().numpy(), g[f"{tag}_v2p"])
            np.testing.assert_array_equal(ss.cpu().numpy(), g[f"{tag}_spatial_shape"])
            np.testing.assert_allclose(vf.cpu().numpy(), g[f"{tag}_voxel_feats"], rtol=1e-6, atol=1e-6)
    # out-of-range points and the voxel capacity
    vx = PointToVoxel([0.2] * 3, [0.0, 0.0, 0.0, 1.0, 1.0, 1.0], 4, 3, 2, device=dev)
    pts = torch.tensor([[0.05, 0.05, 0.05, 1.0], [0.06, 0.05, 0.05, 2.0], [0.07, 0.05, 0.05, 3.0], [0.5, 0.5, 0.5, 4.0], [1.5, 0.5, 0.5, 5.0],
                        [-0.1, 0.5, 0.5, 6.0], [0.9, 0.9, 0.9, 7.0], [0.05, 0.9, 0.9, 8.0], [0.05, 0.5, 0.9, 9.0]], device=dev)
    vf, vc, num, ids = vx.generate_voxel_with_id(pts)
    assert ids.tolist() == [0, 0, 0, -1, -1, -1, -1, 2, 1]          # ascending (x, y, z): (0,0,0), (0,2,4), (0,4,4) kept; (2,2,2), (4,4,4) over capacity
    assert num.tolist() == [2, 1, 1] and vc.tolist() == [[0, 0, 0], [4, 2, 0], [4, 4, 0]]
    assert vf[0, :, 3].tolist() == [1.0, 2.0] and vf[1, :, 3].tolist() == [9.0, 0.0]


def test_hdbscan_prim_fallback_threshold_is_a_documented_knob():
    """Tie-heavy input ABOVE the Prim fall-back cap (50 000 points by default: the O(n^2) form takes minutes beyond): `auto` keeps the quadtree
    form's tree and warns; `prim_fallback_max` (or TL_HDBSCAN_PRIM_MAX) moves the cap, and with the cap raised the result IS the Prim form's.
    The kept-tree labels are the same clusters with >= 97 % identical point assignments on a 0.25 m lattice (INTEGRATION.md states this)."""
    import warnings
    from treelearn_amd.cluster import hdbscan
    rng = np.random.default_rng(3)
    n = 52_000
    c = rng.uniform(0, 120, (40, 2))
    xy = (np.round((c[rng.integers(0, 40, n)] + rng.normal(0, 1.5, (n, 2))) * 4) / 4).astype(np.float32)       # lattice + duplicates: ties everywhere
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        kept = hdbscan(xy, 50)
    assert any("tied tree weights" in str(x.message) for x in w), "the kept-tree path must say so"
    prim = hdbscan(xy, 50, algorithm="prim")
    np.testing.assert_array_equal(hdbscan(xy, 50, prim_fallback_max=60_000), prim)
    assert len(set(kept[kept >= 0])) == len(set(prim[prim >= 0]))
    agree = 0
    for cl in set(kept.tolist()):
        m = kept == cl
        vals, cnts = np.unique(prim[m], return_counts=True)
        agree += cnts.max() if cl != -1 else int((prim[m] == -1).sum())
    print(f"hdbscan kept-tree vs Prim form on {n} tied points: {agree / n:.4f} identical assignments")
    assert agree / n >= 0.97


# ------------------------------------------------------------------------------------------------ round 6: the forms tl_forward uses of two geometry steps
@pytest.mark.parametrize("n,spread", [(1, 1.0), (37, 3.0), (5000, 9.0), (300_000, 30.0)])
def test_point_coords_single_tile_form_equals_the_atomic_form(n, spread):
    """tl_voxel_point_coords_one (no same-address atomics: per-workgroup partial minima, per-workgroup extents folded by the caller) against
    tl_voxel_point_coords with B = 1: the same per-point voxel coordinates bit for bit, the same extent and error flag -- incl. negative
    coordinates, points far from the origin, and a point whose batch id is not 0 (flagged by both)."""
    import ctypes
    from treelearn_amd import _hip
    L = _hip.lib(); st = _hip.stream()
    gen = torch.Generator(device="cuda"); gen.manual_seed(n)
    xyz = ((torch.rand((n, 3), device="cuda", generator=gen) - 0.5) * spread + torch.tensor([-4321.0, 98765.0, 12.5], device="cuda")).contiguous()
    for bad in (False, True):
        bid = torch.zeros(n, dtype=torch.int64, device="cuda")
        if bad:
            bid[n // 2] = 1
        pc_a = torch.empty((n, 4), dtype=torch.int32, device="cuda"); maxc = torch.empty(4, dtype=torch.int32, device="cuda")
        mm = torch.empty(6, dtype=torch.int32, device="cuda")
        _hip.check(L.tl_voxel_point_coords(_hip.ptr(xyz), _hip.ptr(bid), n, 1, 0.1, _hip.ptr(mm), _hip.ptr(pc_a), _hip.ptr(maxc), st), "atomic form")
        pc_b = torch.empty((n, 4), dtype=torch.int32, device="cuda")
        parts = torch.full((1024, 4), -7, dtype=torch.int32, device="cuda"); ws = torch.empty(6 * 256, dtype=torch.int32, device="cuda")
        n_parts = ctypes.c_int32(0)
        _hip.check(L.tl_voxel_point_coords_one(_hip.ptr(xyz), _hip.ptr(bid), n, 0.1, _hip.ptr(ws), _hip.ptr(pc_b), _hip.ptr(parts), ctypes.byref(n_parts), st),
                   "single-tile form")
        torch.cuda.synchronize()
        assert 1 <= n_parts.value <= 1024
        folded = parts[:n_parts.value].max(0).values
        assert bool((parts[n_parts.value:] == -7).all())                    # rows beyond n_parts are not written
        assert torch.equal(folded[:3], maxc[:3]) and int(folded[3]) == int(maxc[3]) == int(bad)
        if not bad:
            assert torch.equal(pc_a, pc_b)
            assert int(pc_b[:, 1:].min()) == 0                                # the minimum point sits in voxel 0 of every axis


def test_rulebooks_build_packed_inverse_table_and_optional_parent():
    """tl_level.inv_packed / parent = NULL (what tl_forward passes for level 1): the packed table decodes to exactly the one-hot table of the
    default call, the other tables are unchanged, and a conv through table_one_hot = 2 equals the conv through the one-hot table bit for bit
    (16-bit and bf16x3); a shape the gather-once kernel does not serve is refused rather than misread."""
    import ctypes
    from treelearn_amd import _hip, ops
    from treelearn_amd.geometry import build_geometry, _nwords
    from treelearn_amd.synth import make_tile
    L = _hip.lib(); st = _hip.stream()
    t = make_tile(extent=13.0, voxel=0.1, n_trees=5, fill=0.1, seed=21)
    pts = torch.from_numpy(t["points"]).cuda(); N = len(pts)
    g = build_geometry(pts, torch.zeros(N, dtype=torch.int64, device="cuda"), 1, 0.1, 3, [500, 500, 1000])
    f, c = g.levels[0], g.levels[1]
    assert f.n > 20_000                                                     # (above the small-level threshold: the direct kernel serves the conv)
    i32 = lambda *sh: torch.full(sh, 12345, dtype=torch.int32, device="cuda")   # noqa: E731
    arr = (_hip.Level * 2)()
    keep = []
    for li, lv in enumerate((f, c)):
        a = arr[li]
        a.dims[:] = lv.dims; a.n = lv.n
        a.bitmap = lv.bitmap.data_ptr(); a.prefix = lv.prefix.data_ptr()
        co, nb = i32(lv.n, 4), i32(27, lv.n); keep += [co, nb]
        a.coords = co.data_ptr(); a.nbr = nb.data_ptr()
    child, invp = i32(8, c.n), i32(f.n)
    arr[0].child = child.data_ptr(); arr[0].parent = None; arr[0].inv = None; arr[0].inv_packed = invp.data_ptr()
    _hip.check(L.tl_rulebooks_build(arr, 2, None, 0, None, 0, None, st), "tl_rulebooks_build")
    torch.cuda.synchronize()
    assert torch.equal(child, f.child) and torch.equal(keep[0], f.coords) and torch.equal(keep[1], f.nbr)
    dec = torch.full((8, f.n), -1, dtype=torch.int32, device="cuda")
    rows = torch.nonzero(invp >= 0).squeeze(1)
    dec[(invp[rows] & 7).long(), rows] = invp[rows] >> 3
    assert torch.equal(dec, f.inv)
    assert bool(((invp >= 0) == (f.parent >= 0)).all())
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    w_ref = torch.randn((32, 2, 2, 2, 64), device="cuda", generator=gen) * 0.1
    for dt, x3 in ((torch.bfloat16, False), (torch.float16, False), (torch.float32, True)):
        ops.PACK_X3 = x3
        try:
            w = ops.pack_weight(w_ref, dt)
        finally:
            ops.PACK_X3 = False
        x = torch.randn((c.n, 64), device="cuda", generator=gen).to(dt)
        ref = ops.conv_fwd(x, w, f.inv, f.n, one_hot=True)
        out = torch.empty_like(ref)
        a = _hip.ConvArgs()
        a.in_ = x.data_ptr(); a.in_ld = 64; a.weight = w.data_ptr(); a.weight_frag = _hip.ptr(getattr(w, "_tl_frag", None)); a.weight_x3 = _hip.ptr(getattr(w, "_tl_x3", None))
        a.table = invp.data_ptr(); a.table_one_hot = 2; a.n_out = f.n; a.n_in = c.n; a.K = 8; a.Cin = 64; a.Cout = 32; a.dtype = _hip.dtype_code(dt)
        a.out = out.data_ptr(); a.out_ld = 32
        _hip.check(L.tl_conv_fwd(ctypes.byref(a), st), "packed one-hot conv")
        assert torch.equal(out, ref), dt
    # ... and the exact fp32 kernels (no split-bf16 weights) do not know the packed form
    w32 = ops.pack_weight(w_ref, torch.float32)
    x32 = torch.randn((c.n, 64), device="cuda", generator=gen)
    out32 = torch.empty((f.n, 32), device="cuda")
    a = _hip.ConvArgs()
    a.in_ = x32.data_ptr(); a.in_ld = 64; a.weight = w32.data_ptr(); a.table = invp.data_ptr(); a.table_one_hot = 2
    a.n_out = f.n; a.n_in = c.n; a.K = 8; a.Cin = 64; a.Cout = 32; a.dtype = _hip.TL_F32; a.out = out32.data_ptr(); a.out_ld = 32
    assert L.tl_conv_fwd(ctypes.byref(a), st) == _hip.TL_ERR_UNSUPPORTED


def test_postprocessing_edge_cases():
    """The callers behind the tile loop at the sizes a sparse plot produces: grouping with 0 / 1 / 2 / 3 candidate points (DBSCAN with
    min_samples 2 = components of the eps-graph, util/pipeline.py:173-180), `get_instances` when no point passes the masks, the k-NN fill
    with nothing to fill and with fewer assigned points than neighbours (sklearn's ValueError, as in the reference, util/pipeline.py:287-296),
    label propagation onto an empty target."""
    from treelearn_amd.util.pipeline import get_instances, get_instances_device, group_dbscan
    from treelearn_amd.util.postprocess import assign_remaining_points_nearest_neighbor, propagate_preds
    pts = np.array([[0.0, 0.0], [0.1, 0.0], [5.0, 5.0]], np.float32)
    assert group_dbscan(pts[:0], 0.5, 1, 0, 1).tolist() == []
    assert group_dbscan(pts[:1], 0.5, 1, 0, 1).tolist() == [0]                      # a lone point is noise -> not assigned
    assert group_dbscan(pts[:2], 0.5, 2, 0, 1).tolist() == [1, 1]
    assert group_dbscan(pts[:2], 0.5, 3, 0, 1).tolist() == [0, 0]                   # a pair below tau_min
    assert group_dbscan(pts, 0.5, 1, 0, 1).tolist() == [1, 1, 0]
    rng = np.random.default_rng(8)
    g = dict(tree_conf_thresh=0.5, tau_vert=0.6, tau_off=2.0, tau_min=3, tau_group=0.15, use_hdbscan=False)
    for n in (0, 1, 40):
        c = rng.uniform(0, 10, (n, 3)).astype(np.float32); off = rng.normal(size=(n, 3)).astype(np.float32)
        logits = rng.normal(size=(n, 2)).astype(np.float32) - np.array([50, 0], np.float32); vert = rng.uniform(0, 1, n).astype(np.float32)
        host = get_instances(c, off, logits, g, vert, 0, -1, 0, 1)                  # class 0 = tree never reaches the confidence threshold
        assert host.dtype == np.int64 and host.tolist() == [-1] * n
        T = lambda a: torch.from_numpy(a).cuda()                                      # noqa: E731
        assert get_instances_device(T(c), T(off), T(logits), g, T(vert), 0, -1, 0, 1).cpu().tolist() == [-1] * n
    q = rng.uniform(0, 10, (6, 3)).astype(np.float32)
    pred = np.array([1, 1, 2, 3, 3, 3])
    np.testing.assert_array_equal(assign_remaining_points_nearest_neighbor(q, pred, -1), pred)          # nothing to fill
    with pytest.raises(ValueError, match="n_neighbors <= n_samples_fit"):
        assign_remaining_points_nearest_neighbor(q, np.array([1, 1, 2, -1, -1, -1]), -1)                # 3 assigned points, 5 neighbours
    assert assign_remaining_points_nearest_neighbor(q, np.array([1, 1, 2, -1, -1, -1]), -1, n_neighbors=3).tolist()[:3] == [1, 1, 2]
    out = propagate_preds(q, pred, q[:0], 5)
    assert out.shape == (0,) and out.dtype == np.int64
    with pytest.raises(ValueError):
        propagate_preds(q[:2], pred[:2], q, 5)
