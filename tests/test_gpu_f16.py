"""float16 inference path (TL_F16: BASELINE config 5 says "fp16"; the reference trains / evaluates under fp16 autocast,
tools/training/train.py:32): the conv kernel families, compiled a second time with IEEE-half conversions (csrc/tl_half.h), against the
oracle fed the same fp16-rounded operands; the fused engine on the config-5 stress tile with overflow counts."""
import numpy as np
import pytest
import torch

from oracle import sparse_ops as osp
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict

pytestmark = pytest.mark.gpu


def _f16_round(a):
    return torch.from_numpy(a).to(torch.float16).float().numpy()


def rel_err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(np.asarray(b, np.float64)).max(), 1e-30))


# direct (weights in LDS), the 4-channel input conv, stream-q, stream, small-level, down / inverse shapes, 1x1
@pytest.mark.parametrize("cin,cout,K,n_out,one_hot", [
    (32, 32, 27, 17001, False), (64, 32, 27, 16500, False), (4, 32, 27, 20000, False), (64, 64, 27, 16500, False), (128, 64, 27, 16400, False),
    (96, 96, 27, 16400, False), (192, 96, 27, 16402, False), (128, 128, 27, 16401, False), (256, 128, 27, 16404, False), (32, 64, 8, 16400, False),
    (64, 32, 8, 20000, True), (96, 64, 8, 16385, True), (64, 32, 1, 20000, False), (160, 160, 27, 6500, False), (224, 224, 27, 401, False),
    (320, 160, 27, 3000, False)])
def test_conv_fwd_f16_vs_oracle(cin, cout, K, n_out, one_hot):
    """fp16 storage + f16 MFMA, fp32 accumulate, residual + BatchNorm/ReLU second view: against the oracle on the same fp16-rounded
    operands (one fp16 rounding of the output: 2^-11)."""
    from treelearn_amd import ops
    rng = np.random.default_rng(cin * 7 + cout + K)
    d = torch.device("cuda")
    n_in = n_out + 77 if K > 1 else n_out
    x = _f16_round(rng.normal(size=(n_in, cin)).astype(np.float32))
    k = round(K ** (1 / 3))
    w = _f16_round((rng.normal(size=(cout, k, k, k, cin)) / np.sqrt(cin * K)).astype(np.float32))
    if one_hot:
        table = np.full((n_out, K), -1, np.int32)
        table[np.arange(n_out), rng.integers(0, K, n_out)] = rng.integers(0, n_in, n_out)
    else:
        table = rng.integers(-1, n_in, size=(n_out, K)).astype(np.int32)
        table[rng.uniform(size=table.shape) < 0.4] = -1
    if K == 1:
        table = np.arange(n_in, dtype=np.int32)[:, None]
    res = _f16_round(rng.normal(size=(n_out, cout)).astype(np.float32))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32); sh = rng.normal(0, 0.3, cout).astype(np.float32)
    ref = osp.conv_table(torch.from_numpy(x), torch.from_numpy(w), table, n_out).numpy() + res
    T = lambda a, dt=torch.float16: torch.from_numpy(a).to(d).to(dt)                     # noqa: E731
    wp = ops.pack_weight(T(w, torch.float32), torch.float16)
    tab = None if K == 1 else torch.from_numpy(np.ascontiguousarray(table.T)).to(d)
    act = torch.empty((n_out, cout), dtype=torch.float16, device=d)
    out = ops.conv_fwd(T(x), wp, tab, n_out, residual=T(res), out2=(act, T(sc, torch.float32), T(sh, torch.float32), True), one_hot=one_hot)
    assert out.dtype == torch.float16
    assert rel_err(out.float().cpu().numpy(), ref) < 1.5e-3
    assert rel_err(act.float().cpu().numpy(), np.maximum(ref * sc + sh, 0)) < 1.5e-3
    # and the bf16 kernels are untouched by the second compilation: same call in bf16 against its own rounding
    outb = ops.conv_fwd(T(x, torch.bfloat16), ops.pack_weight(T(w, torch.float32), torch.bfloat16), tab, n_out, residual=T(res, torch.bfloat16), one_hot=one_hot)
    assert outb.dtype == torch.bfloat16 and rel_err(outb.float().cpu().numpy(), ref) < 2e-2


def _trained_like(voxel, sshape, batch, seed=11):
    """Synthetic weights with every BatchNorm's running statistics SET to the statistics of this tile (one training-mode forward with
    momentum 1: running = batch statistics, what a converged net has seen of such data).  A random-init net with arbitrary running
    statistics saturates and reaches 1e5 in places -- and so does one whose statistics were only blended towards the data (two passes at
    the default momentum 0.1 leave 81 % of the arbitrary initial values: 845 of 95 M outputs overflow fp16 on the 19 M-point tile)."""
    from treelearn_amd.model import TreeLearn
    m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=sshape, voxel_size=voxel, compute_dtype=torch.bfloat16)
    m.load_state_dict(random_state_dict(seed, channels=32, num_blocks=7), strict=True)
    m = m.cuda().train()
    bns = [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm1d)]
    for mod in bns:
        mod.momentum = 1.0
    with torch.no_grad():
        m(batch, return_loss=False)                           # module-by-module path, batch statistics -> the running statistics
    for mod in bns:
        mod.momentum = 0.1
    return m.eval()


def test_config5_stress_tile_forward_fp16(tile5_batch):
    """BASELINE config 5 as worded and at its stated size (config5_20m: 19.0 M points, 17.9 M voxels -- the tile of bench.py's config5
    block, shared with tests/test_gpu_configs.py through the session fixture): the 0.05 m tile in fp16, on trained-like weights.  Every output finite (overflow count 0),
    decisions as in bf16 / closer to fp32 than bf16 is; on RANDOM-INIT running statistics the same forward overflows fp16 (counted), which
    is why bf16 stays the default 16-bit mode."""
    from treelearn_amd.model import TreeLearn
    batch = tile5_batch
    assert batch["coords"].shape[0] > 18_500_000
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    mb = _trained_like(0.05, None, gb)
    outs = {}
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=None, voxel_size=0.05, compute_dtype=dt)
        m.load_state_dict(mb.state_dict(), strict=True)
        m = m.cuda().eval()
        with torch.no_grad():
            o = m(gb, return_loss=False)
        outs[dt] = {k: o[k].float() for k in ("semantic_prediction_logits", "offset_predictions")}
        del m
    o32, ob, oh = outs[torch.float32], outs[torch.bfloat16], outs[torch.float16]
    bad = sum(int((~torch.isfinite(v)).sum()) for v in oh.values())
    print(f"config 5 fp16, trained-like weights: {bad} non-finite outputs of {sum(v.numel() for v in oh.values())}")
    assert bad == 0
    for k in oh:
        eh = float((oh[k] - o32[k]).abs().max() / o32[k].abs().max()); eb = float((ob[k] - o32[k]).abs().max() / o32[k].abs().max())
        print(f"  {k}: max-norm error vs fp32: fp16 {eh:.2e}, bf16 {eb:.2e}")
        assert eh < 2e-2 and eh <= eb * 1.2
    flips = float((oh["semantic_prediction_logits"].argmax(1) != o32["semantic_prediction_logits"].argmax(1)).float().mean())
    assert flips < 0.01, flips
    # random-init running statistics: the overflow the reference needs its GradScaler / trained BatchNorms for
    m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=None, voxel_size=0.05, compute_dtype=torch.float16)
    m.load_state_dict(random_state_dict(11, channels=32, num_blocks=7), strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        o = m(gb, return_loss=False)
    bad_r = sum(int((~torch.isfinite(o[k])).sum()) for k in ("semantic_prediction_logits", "offset_predictions"))
    print(f"config 5 fp16, random-init running statistics: {bad_r} non-finite outputs")


def test_autocast_float16_eval_runs_the_f16_kernels():
    """Inside `torch.autocast("cuda", dtype=torch.float16)` an eval-mode forward of a default (fp32) model takes the float16 plan."""
    from treelearn_amd.model import TreeLearn
    batch = make_batch([make_tile(extent=10.0, voxel=0.1, n_trees=4, fill=0.10, seed=2)])
    m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, channels=32, num_blocks=4)
    m.load_state_dict(random_state_dict(3, channels=32, num_blocks=4), strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        ref = m(batch, return_loss=False)
        with torch.autocast("cuda", dtype=torch.float16):
            o = m(batch, return_loss=False)
    assert m._plan.dtype == torch.float16
    for k in ("semantic_prediction_logits", "offset_predictions"):
        assert bool(torch.isfinite(o[k]).all())
        assert rel_err(o[k].float().cpu().numpy(), ref[k].float().cpu().numpy()) < 3e-2, k


@pytest.mark.parametrize("cin,cout,K", [(32, 32, 27), (64, 64, 27), (96, 96, 27), (32, 64, 8), (64, 32, 1), (32, 3, 1), (4, 32, 27)])
def test_f16_weight_gradient_vs_float64(cin, cout, K):
    """tl_conv_wgrad on IEEE-half operands (the _f16 compilation of the weight-gradient units, csrc/tl_f16_train.h) against float64 on the same
    rounded inputs, and against the bf16 kernels' accuracy class."""
    import torch
    from treelearn_amd import ops
    from treelearn_amd.geometry import build_geometry
    from treelearn_amd.synth import make_tile
    t = make_tile(extent=12.0, voxel=0.1, n_trees=5, fill=0.1, seed=2)
    xyz = torch.from_numpy(t["points"]).cuda()
    g = build_geometry(xyz, torch.zeros(len(xyz), dtype=torch.int64, device="cuda"), 1, 0.1, 3, [500, 500, 1000])
    lv = g.levels[0]
    if K == 27:
        table, n_out = lv.nbr, lv.n
    elif K == 8:
        table, n_out = lv.child, g.levels[1].n
    else:
        table, n_out = None, lv.n
    gen = torch.Generator(device="cuda"); gen.manual_seed(cin + 7 * cout + K)
    x = torch.randn((lv.n, cin), device="cuda", generator=gen).half()
    go = (torch.randn((n_out, cout), device="cuda", generator=gen) * 0.1).half()
    gw = ops.conv_wgrad(x, go, table, n_out, K)
    assert gw.dtype == torch.float32 and tuple(gw.shape) == (K, cout, cin)
    xd, gd = x.double(), go.double()
    ref = torch.zeros((K, cout, cin), dtype=torch.float64, device="cuda")
    for k in range(K):
        if table is None:
            ref[k] = gd.T @ xd
        else:
            idx = table[k].long(); m = idx >= 0
            ref[k] = gd[m].T @ xd[idx[m]]
    err = float((gw.double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err


def test_f16_batchnorm_train_kernels_vs_torch():
    """tl_bn_train_stats / tl_bn_train_bwd on float16 rows against torch's float64 BatchNorm + ReLU on the same rounded inputs."""
    import torch
    from treelearn_amd import ops
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    n, C = 50_000, 64
    x = (torch.randn((n, C), device="cuda", generator=gen) * 1.7 + 0.3).half()
    dy = torch.randn((n, C), device="cuda", generator=gen).half()
    gamma = torch.rand(C, device="cuda", generator=gen) + 0.5; beta = torch.randn(C, device="cuda", generator=gen) * 0.2
    st = ops.bn_train_stats(x, gamma, beta, 1e-4, 0.1)
    y = ops.affine_relu(x, st[2], st[3], True)
    dx, dgamma, dbeta = ops.bn_train_bwd(x, dy, st, True)
    assert y.dtype == torch.float16 and dx.dtype == torch.float16
    xd = x.double().requires_grad_(True)
    yd = torch.relu(torch.nn.functional.batch_norm(xd, None, None, gamma.double(), beta.double(), True, 0.1, 1e-4))
    gd = gamma.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
    yd2 = torch.relu(torch.nn.functional.batch_norm(xd, None, None, gd, bd, True, 0.1, 1e-4))
    yd2.backward(dy.double())
    assert float((y.double() - yd).abs().max()) < 4e-3                            # one half-precision rounding of values up to ~4
    assert float((st[0].double() - x.double().mean(0)).abs().max()) < 1e-5
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())        # noqa: E731
    assert rel(dgamma, gd.grad) < 2e-3 and rel(dbeta, bd.grad) < 2e-3             # (the ReLU mask is decided on the fp32 affine of the rounded x)
    assert rel(dx, xd.grad) < 5e-3


def test_f16_training_step_close_to_fp32_step():
    """One training step (loss + every gradient) under a float16 autocast against the fp32 step of the same model: the loss within 2e-2 and
    every large gradient tensor at cosine >= 0.98 -- the accuracy class of the bf16 regime (which has three bits less) or better."""
    import torch
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    cfg = dict(channels=32, num_blocks=4)
    b = make_batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.10, seed=s) for s in (1, 2)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}

    def step(dt):
        m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, **cfg)
        m.load_state_dict(random_state_dict(5, **cfg), strict=True)
        m = m.cuda().train()
        if dt is None:
            loss, _ = m(gb, return_loss=True)
        else:
            with torch.autocast("cuda", dtype=dt):
                loss, _ = m(gb, return_loss=True)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {n: p.grad.detach().double().flatten() for n, p in m.named_parameters() if p.grad is not None}

    l32, g32 = step(None)
    res = {}
    for dt in (torch.float16, torch.bfloat16):
        l, g = step(dt)
        assert abs(l - l32) <= 2e-2 * abs(l32), (dt, l, l32)
        nmax = max(float(v.norm()) for v in g32.values())
        worst = min(float((g[k] * g32[k]).sum() / (g[k].norm() * g32[k].norm())) for k in g32 if float(g32[k].norm()) > 1e-3 * nmax)
        res[dt] = worst
        assert all(bool(torch.isfinite(v).all()) for v in g.values()), dt
    print("worst gradient cosine vs the fp32 step:", {str(k): round(v, 5) for k, v in res.items()})
    assert res[torch.float16] >= 0.98 and res[torch.float16] >= res[torch.bfloat16] - 5e-3


def test_16bit_training_step_with_voxel_features_on_the_block_local_level():
    """Mixed-precision training keeps level 1 in the block-local row order; with `use_coords` / `use_feats` the voxel-mean features come back
    in that order too and the input conv is a real 4 -> 32 conv (not the ones table).  One step in float16 and bf16 against the fp32 step
    (canonical order) of the same model on a tile large enough for the block-local path."""
    import torch
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    b = make_batch([make_tile(extent=14.0, voxel=0.1, n_trees=8, fill=0.10, seed=s) for s in (3, 4)])
    g = torch.Generator().manual_seed(0)
    b["input_feats"] = torch.randn(b["coords"].shape[0], 1, generator=g)
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}

    def step(dt, vf=True):
        m = TreeLearn(use_feats=vf, use_coords=vf, spatial_shape=[500, 500, 1000], voxel_size=0.1)
        m.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
        m = m.cuda().train()
        if dt is None:
            loss, _ = m(gb, return_loss=True)
        else:
            with torch.autocast("cuda", dtype=dt):
                loss, _ = m(gb, return_loss=True)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {n: p.grad.detach().double().flatten() for n, p in m.named_parameters() if p.grad is not None}, m

    l32, g32, _ = step(None)
    for dt in (torch.float16, torch.bfloat16):
        from treelearn_amd import autograd as ag
        ag.RELU_MASK_SINK = {}                                  # (switches the `_last_geom` test hook on)
        try:
            l, gr, m = step(dt)
        finally:
            ag.RELU_MASK_SINK = None
        assert m._last_geom.blocked, "the tile is large enough for the block-local level-1 path"
        assert abs(l - l32) <= 3e-2 * abs(l32), (dt, l, l32)
        nmax = max(float(v.norm()) for v in g32.values())
        cos = lambda a, b_: min((float((a[k] * b_[k]).sum() / (a[k].norm() * b_[k].norm())), k) for k in b_ if float(b_[k].norm()) > 1e-3 * nmax)   # noqa: E731
        worst = cos(gr, g32)
        # the yardstick: the same step of the same net on all-ones voxel features (the reference's default), against ITS fp32 step
        if dt == torch.float16:
            l32o, g32o, _ = step(None, False)
        lo, go, _ = step(dt, False)
        nmax_o = max(float(v.norm()) for v in g32o.values())
        base = min((float((go[k] * g32o[k]).sum() / (go[k].norm() * g32o[k].norm())), k) for k in g32o if float(g32o[k].norm()) > 1e-3 * nmax_o)
        print(dt, "loss", l, "fp32", l32, "worst gradient cosine", worst, "| all-ones features:", base)
        assert worst[0] >= base[0] - 0.05, (dt, worst, base)
