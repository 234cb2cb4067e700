"""Training-mode conv epilogues (tl_conv_args.epi_mode) and the fused BatchNorm -> ReLU -> conv autograd node against the separate
passes / float64 (reference: the `norm_fn(C), nn.ReLU(), conv` triples of tree_learn/model/blocks.py:55-70,102-123 in train() mode,
stepped by tools/training/train.py:30-44)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda")


def _table(rng, n_out, n_in, K, density=0.4):
    t = rng.integers(0, n_in, size=(K, n_out)).astype(np.int32)
    t[rng.uniform(size=t.shape) > density] = -1
    return t


# (Cin, Cout, K, rows): the direct kernel (whole weight tensor in LDS), the 4-channel input conv, stream-q, stream, the one-hot stream form
SHAPES = [(32, 32, 27, 40001), (64, 32, 27, 30000), (4, 32, 27, 50017), (64, 64, 27, 33000), (128, 64, 27, 20000), (96, 96, 27, 17000),
          (192, 96, 27, 17000), (32, 64, 8, 30000), (64, 32, 1, 25000), (64, 96, 8, 18000), (128, 128, 27, 17001)]


@pytest.mark.parametrize("cin,cout,K,n", SHAPES)
def test_conv_epilogue_statistics(cin, cout, K, n):
    """epi="stats": same output bits as the plain launch; the partial sums add up to the column sums of the STORED result."""
    from treelearn_amd import ops
    rng = np.random.default_rng(cin + cout + K)
    d = _dev()
    n_in = n + 33 if K > 1 else n
    x = torch.from_numpy(rng.normal(size=(n_in, cin)).astype(np.float32)).to(d).bfloat16()
    w = ops.pack_weight(torch.from_numpy((rng.normal(size=(cout, K, cin)) / np.sqrt(cin * K * 0.4)).astype(np.float32).reshape(cout, K, 1, 1, cin)).to(d), torch.bfloat16)
    tab = None if K == 1 else torch.from_numpy(_table(rng, n, n_in, K)).to(d)
    res = torch.from_numpy(rng.normal(size=(n, cout)).astype(np.float32)).to(d).bfloat16()
    for residual in (None, res):
        plain = ops.conv_fwd(x, w, tab, n, residual=residual)
        r = ops.conv_fwd(x, w, tab, n, residual=residual, epi="stats")
        assert r is not None, "this shape is expected on a kernel family with the training epilogue"
        out, parts, nparts = r
        assert torch.equal(out, plain)
        assert 0 < nparts <= parts.shape[0]
        s = parts[:nparts].sum(0).cpu().numpy()
        y = out.double().cpu().numpy()
        np.testing.assert_allclose(s[0], y.sum(0), rtol=1e-6, atol=1e-6 * np.abs(y).sum(0).max())
        np.testing.assert_allclose(s[1], (y * y).sum(0), rtol=1e-6)
        r2 = ops.conv_fwd(x, w, tab, n, residual=residual, epi="stats")
        assert torch.equal(r2[1][:nparts], parts[:nparts])                       # deterministic


@pytest.mark.parametrize("cin,cout,K,n", SHAPES[:2] + SHAPES[3:7] + SHAPES[7:9])
@pytest.mark.parametrize("relu", [True, False])
def test_conv_epilogue_bn_backward(cin, cout, K, n, relu):
    """epi=("bn_bwd", x, st, relu): the views receive dy masked by the ReLU of the BatchNorm in front of the layer, the partial sums are
    dbeta / dgamma; tl_bn_train_bwd_from_parts then equals tl_bn_train_bwd on the unmasked dy."""
    from treelearn_amd import ops
    rng = np.random.default_rng(cin * 3 + cout + K)
    d = _dev()
    n_in = n + 33 if K > 1 else n
    gy = torch.from_numpy(rng.normal(size=(n_in, cin)).astype(np.float32)).to(d).bfloat16()         # "grad_out" rows gathered by the transposed conv
    w = ops.pack_weight(torch.from_numpy((rng.normal(size=(cout, K, cin)) / np.sqrt(cin * K * 0.4)).astype(np.float32).reshape(cout, K, 1, 1, cin)).to(d), torch.bfloat16)
    tab = None if K == 1 else torch.from_numpy(_table(rng, n, n_in, K)).to(d)
    xb = torch.from_numpy((rng.normal(size=(n, cout)) * 2 + 0.5).astype(np.float32)).to(d).bfloat16()   # the BatchNorm input of the layer
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(d); beta = torch.from_numpy(rng.normal(0, 0.3, cout).astype(np.float32)).to(d)
    st = ops.bn_train_stats(xb, gamma, beta, 1e-4, 0.1)
    dy = ops.conv_fwd(gy, w, tab, n)
    r = ops.conv_fwd(gy, w, tab, n, epi=("bn_bwd", xb, st, relu))
    assert r is not None
    g, parts, nparts = r
    keep = ((xb.double() * st[2].double() + st[3].double()) > 0) if relu else torch.ones_like(xb, dtype=torch.bool)
    assert torch.equal(g, torch.where(keep, dy, torch.zeros_like(dy)))
    add = torch.from_numpy(rng.normal(size=(n, cout)).astype(np.float32)).to(d).bfloat16()
    ref = ops.bn_train_bwd(xb, dy, st, relu, dx_add=add)
    got = ops.bn_train_bwd_from_parts(xb, g, st, parts, nparts, dx_add=add)
    assert got is not None
    for a, b, tol in ((got[1], ref[1], 2e-5), (got[2], ref[2], 2e-5)):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-6
    err = float((got[0].float() - ref[0].float()).abs().max()) / float(ref[0].float().abs().max())
    assert err < 1e-2                                                                       # one bf16 rounding of dx


def test_fused_training_nodes_equal_separate_passes():
    """One training step of a 4-level model on two 14 m tiles (the big levels take the epilogue-fused kernels, the small ones the separate
    passes) with the fusion on and off (autograd.FUSE_BN).  fp32: same loss, running statistics and gradients up to summation order.
    bf16: the two paths differ by rounding flips that the small deep levels amplify (a channel whose batch mean is several standard
    deviations sees a bf16 ulp as per cents of its normalised value), so the check is that the fused gradients are as close to the fp32
    gradients as the separate-pass gradients are -- a bias in the fused path would show as a larger distance."""
    from treelearn_amd import autograd as ag
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    cfg = dict(channels=32, num_blocks=4)
    batch = make_batch([make_tile(extent=14.0, voxel=0.1, n_trees=8, fill=0.10, seed=s) for s in (3, 4)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        for fuse in (True, False):
            ag.FUSE_BN = fuse
            try:
                model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, compute_dtype=dtype, **cfg)
                model.load_state_dict(random_state_dict(5, **cfg), strict=True)
                model = model.cuda().train()
                loss, _ = model(gb, return_loss=True)
                loss.backward()
                res[dtype, fuse] = (float(loss.detach()), {n: p.grad.detach().float().cpu().numpy().ravel() for n, p in model.named_parameters()},
                                    {n: b.detach().float().cpu().numpy() for n, b in model.named_buffers() if "running" in n})
            finally:
                ag.FUSE_BN = True

    def cosd(a, b):
        return 1 - float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    # a Linear bias in front of a BatchNorm has a zero gradient in exact arithmetic: what is computed there is rounding noise
    names = [n for n in res[torch.float32, False][1] if n not in ("semantic_linear.0.bias", "offset_linear.0.bias")]
    (la, ga, ba), (lb, gb_, bb) = res[torch.float32, True], res[torch.float32, False]
    assert la == pytest.approx(lb, rel=1e-5)
    for n in bb:
        np.testing.assert_allclose(ba[n], bb[n], rtol=1e-4, atol=1e-6)
    for n in names:
        assert np.isfinite(ga[n]).all()
        if np.linalg.norm(gb_[n]) > 0:
            assert cosd(ga[n], gb_[n]) < 1e-4, n
            assert abs(np.linalg.norm(ga[n]) / np.linalg.norm(gb_[n]) - 1) < 2e-3, n
    (lf, gf, bf), (ls, gs, bs) = res[torch.bfloat16, True], res[torch.bfloat16, False]
    assert lf == pytest.approx(ls, rel=2e-3) and lf == pytest.approx(lb, rel=2e-2)
    for n in bs:
        np.testing.assert_allclose(bf[n], bs[n], rtol=1e-3, atol=3e-3 * max(1e-3, float(np.abs(bs[n]).max())))
    df = np.array([cosd(gf[n], gb_[n]) for n in names if np.linalg.norm(gb_[n]) > 0])
    ds = np.array([cosd(gs[n], gb_[n]) for n in names if np.linalg.norm(gb_[n]) > 0])
    assert np.isfinite(df).all() and df.mean() < 1.25 * ds.mean() + 1e-3, (df.mean(), ds.mean())
    assert df.max() < 1.5 * ds.max() + 1e-2, (df.max(), ds.max())

def test_side_stream_join_modes_give_identical_gradients(monkeypatch):
    """The weight gradients run on a side stream.  Default: ONE join at the end of the backward pass (autograd final callback; the
    tensors the side stream reads are kept alive until then); TL_WGRAD_JOIN=layer: the main stream waits after every layer;
    TL_WGRAD_STREAM=0: everything on the main stream.  Same kernels, same order of summation: the three must agree BIT FOR BIT, step after
    step, also when gradients are accumulated into an existing `.grad` (where the deferred join must not be used: AccumulateGrad adds right
    after the node returns)."""
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    cfg = dict(channels=32, num_blocks=4)
    batch = make_batch([make_tile(extent=16.0, voxel=0.1, n_trees=10, fill=0.10, seed=s) for s in (5, 6)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}

    def run(mode):
        monkeypatch.delenv("TL_WGRAD_JOIN", raising=False); monkeypatch.delenv("TL_WGRAD_STREAM", raising=False)
        if mode == "layer": monkeypatch.setenv("TL_WGRAD_JOIN", "layer")
        if mode == "serial": monkeypatch.setenv("TL_WGRAD_STREAM", "0")
        model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, compute_dtype=torch.bfloat16, **cfg)
        model.load_state_dict(random_state_dict(5, **cfg), strict=True)
        model = model.cuda().train()
        opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=1e-3)
        out = []
        for step in range(3):
            opt.zero_grad()
            loss, _ = model(gb, return_loss=True)
            loss.backward()
            if step == 1:                                    # accumulate a second backward into the existing .grad
                loss2, _ = model(gb, return_loss=True)
                loss2.backward()
            out.append({n: p.grad.detach().clone() for n, p in model.named_parameters()})
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
        torch.cuda.synchronize()
        return out

    ref = run("serial")
    for mode in ("deferred", "layer"):
        got = run(mode)
        for step, (a, b) in enumerate(zip(ref, got)):
            bad = [n for n in a if not torch.equal(a[n], b[n])]
            assert not bad, (mode, step, bad[:5])


def test_side_stream_join_respects_hooks_and_double_backward():
    """The deferred join is only legal when nothing reads a weight gradient on the main stream before backward() returns
    (treelearn_amd/backward.py `_may_defer_join`).  A tensor hook and a post-accumulate-grad hook on conv weights copy what they see ON THE
    MAIN STREAM while the weight-gradient kernel may still be running on the side stream; what they saw must be the final gradient."""
    from treelearn_amd import backward as B
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    cfg = dict(channels=32, num_blocks=4)
    batch = make_batch([make_tile(extent=16.0, voxel=0.1, n_trees=10, fill=0.10, seed=s) for s in (5, 6)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, compute_dtype=torch.bfloat16, **cfg)
    model.load_state_dict(random_state_dict(5, **cfg), strict=True)
    model = model.cuda().train()
    convs = [(n, p) for n, p in model.named_parameters() if p.dim() == 5]
    seen = {}
    (n1, p1), (n2, p2) = convs[1], convs[5]
    assert B._may_defer_join(p1) is False or True                           # (callable outside a backward pass)
    h1 = p1.register_hook(lambda g, n=n1: seen.__setitem__(n, g.detach().clone()))
    h2 = p2.register_post_accumulate_grad_hook(lambda p, n=n2: seen.__setitem__(n, p.grad.detach().clone()))
    loss, _ = model(gb, return_loss=True)
    loss.backward()
    torch.cuda.synchronize()
    assert set(seen) == {n1, n2}
    assert torch.equal(seen[n1], p1.grad) and torch.equal(seen[n2], p2.grad)
    h1.remove(); h2.remove()
    with torch.no_grad():
        assert B._may_defer_join(convs[2][1].detach().clone().requires_grad_(True))     # plain leaf, no hooks, grad mode off: may defer
    assert not B._may_defer_join(p1)                                        # `.grad` is set now: AccumulateGrad adds at once


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gather_rows_and_scatter_add_vs_torch(dtype):
    """features[v2p] (tl_gather_rows) and its gradient (tl_scatter_add_rows over the stable argsort): duplicates, voxels without a point,
    a negative index (counts from the end, as torch indexing does); bit-reproducible."""
    from treelearn_amd.autograd import gather_rows
    d = _dev()
    g = torch.Generator(device="cpu").manual_seed(3)
    n, N, C = 50000, 140001, 32
    idx = torch.randint(0, n - 100, (N,), generator=g)
    idx[:5000] = torch.arange(5000) // 3                                          # up to three points per voxel
    idx[7] = -1
    idx = idx.to(d)
    x = torch.randn(n, C, generator=g).to(d).to(dtype).requires_grad_(True)
    w = torch.randn(N, C, generator=g).to(d).to(dtype)
    cache = {}
    y = gather_rows(x, idx, cache)
    assert torch.equal(y.detach(), x.detach()[idx])
    (y.float() * w.float()).sum().backward()
    g1 = x.grad.clone(); x.grad = None
    y2 = gather_rows(x, idx, cache); (y2.float() * w.float()).sum().backward()
    assert torch.equal(g1, x.grad) and "v2p_sort" in cache
    ref = torch.zeros(n, C, dtype=torch.float64, device=d).index_add_(0, torch.where(idx < 0, idx + n, idx), w.double())
    err = float((g1.double() - ref).abs().max() / ref.abs().max())
    assert err < (1e-2 if dtype == torch.bfloat16 else 1e-6)


def test_reference_training_step_body_under_autocast():
    """The literal step body of reference tools/training/train.py:30-44 -- zero_grad, `torch.cuda.amp.autocast(enabled=config.fp16)`
    around the forward, the loss `.item()` reads, `scaler.scale(loss).backward()`, clip_grad_norm_, `scaler.step`, `scaler.update` --
    around an unmodified TreeLearn (compute_dtype left at its fp32 default), 20 steps: the autocast region selects the float16 kernels, the
    GradScaler scales / unscales and settles after at most a few halvings, the optimizer steps and the loss falls."""
    from collections import defaultdict
    from treelearn_amd import spconv_compat
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    cfg = dict(channels=32, num_blocks=4)
    batch = make_batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.10, seed=s) for s in (1, 2)])
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, **cfg)
    model.load_state_dict(random_state_dict(5, **cfg), strict=True)
    model = model.cuda()
    optimizer = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-3)
    scaler = torch.cuda.amp.GradScaler(enabled=True)
    seen = []
    hook = model.output_layer.register_forward_hook(lambda m, i, o: seen.append(o.features.dtype))       # the backbone's last activations
    before = model.unet.blocks[0].conv_branch[2].weight.detach().clone()
    losses_dict = defaultdict(list)
    model.train()
    skipped, scales = 0, []
    for _ in range(20):
        optimizer.zero_grad()
        with torch.cuda.amp.autocast(enabled=True):
            loss, loss_dict = model(batch, return_loss=True)
            for key, value in loss_dict.items():
                losses_dict[key].append(value.detach().cpu().item())
        scaler.scale(loss).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, norm_type=2)
        s0 = scaler.get_scale()
        scaler.step(optimizer)
        scaler.update()
        skipped += int(scaler.get_scale() < s0)
        scales.append(scaler.get_scale())
    hook.remove()
    # torch.cuda.amp.autocast's default dtype is float16 -- the reference's regime (tools/training/train.py:32): the region runs the IEEE-half
    # kernels, forward, input / weight gradients and BatchNorm passes alike (round 5; before: bf16 kernels under a float16 autocast)
    assert seen and all(d == torch.float16 for d in seen), seen
    assert spconv_compat.SparseConvolution.amp_dtype is None                        # and left no state behind
    assert all(np.isfinite(v).all() for v in losses_dict.values())
    print(f"fp16 autocast + GradScaler, 20 steps: {skipped} skipped, final scale {scales[-1]:.0f}, "
          f"loss {losses_dict['semantic_loss'][0] + losses_dict['offset_loss'][0]:.4f} -> {losses_dict['semantic_loss'][-1] + losses_dict['offset_loss'][-1]:.4f}")
    assert skipped <= 8 and scales[-1] >= 64.0                                      # the scaler settles (a few halvings at most), steps are taken
    assert not torch.equal(before, model.unet.blocks[0].conv_branch[2].weight.detach())
    assert losses_dict["semantic_loss"][-1] + losses_dict["offset_loss"][-1] < losses_dict["semantic_loss"][0] + losses_dict["offset_loss"][0]
    # the same step without autocast runs fp32 and lands near the same loss
    model2 = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, **cfg)
    model2.load_state_dict(random_state_dict(5, **cfg), strict=True)
    model2 = model2.cuda().train()
    l32, _ = model2(batch, return_loss=True)
    assert losses_dict["semantic_loss"][0] + losses_dict["offset_loss"][0] == pytest.approx(float(l32.detach()), rel=3e-2)


def test_fused_optimizer_updates_reach_the_kernels():
    """torch's single-kernel optimizers (`fused=True`) update the parameters WITHOUT bumping `_version`; the kernel-layout copies of the conv
    weights must expire anyway (autograd._pack_epoch), or training silently keeps running on the weights of step 0.  Three steps with the
    for-each and the fused AdamW from the same start: same losses (the two optimizers agree to ~1e-7 per step)."""
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    cfg = dict(channels=32, num_blocks=4)
    batch = make_batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.10, seed=s) for s in (1, 2)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    traj = {}
    for fused in (False, True):
        model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, compute_dtype=torch.bfloat16, **cfg)
        model.load_state_dict(random_state_dict(5, **cfg), strict=True)
        model = model.cuda().train()
        opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=1e-3, fused=fused)
        w = model.unet.blocks[0].conv_branch[2].weight
        v0 = w._version
        losses = []
        for _ in range(4):
            opt.zero_grad()
            loss, ld = model(gb, return_loss=True)
            losses.append(float(loss.detach()))
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
        traj[fused] = losses
        if fused and w._version != v0:
            print("note: this torch build bumps _version in the fused step; the test then only checks equality of the trajectories")
    print("for-each", traj[False], "fused", traj[True])
    assert traj[False][1] < traj[False][0]                                          # the steps do something
    for a, b in zip(traj[False], traj[True]):
        assert abs(a - b) <= 2e-3 * abs(a), (traj[False], traj[True])


def test_layer_used_directly_sees_an_update_that_leaves_the_version_alone():
    """The same guarantee without the model: a SubMConv3d used on its own, its weight changed between two forward / backward rounds through
    `.data` (no version bump, what a fused optimizer does): the second forward must use the new weights."""
    from treelearn_amd import geometry as G, spconv_compat as spconv
    from treelearn_amd.synth import make_batch, make_tile
    b = make_batch([make_tile(extent=8.0, voxel=0.1, n_trees=3, fill=0.1, seed=4)])
    geom = G.build_geometry(b["coords"].cuda().float(), b["batch_ids"].cuda().long(), 1, 0.1, 2, [128, 128, 512])
    lv = geom.levels[0]
    conv = spconv.SubMConv3d(32, 32, kernel_size=3, padding=1, bias=False, indice_key="subm1").cuda()
    feats = torch.randn(lv.n, 32, device="cuda", requires_grad=True)
    def run():
        x = spconv.SparseConvTensor(feats, lv.coords, list(lv.shape), 1, geometry=geom, level=0)
        return conv(x).features
    y0 = run()
    y0.sum().backward()                                                             # a backward pass went through the layer: the epoch advances
    v = conv.weight._version
    conv.weight.data.mul_(2.0)                                                      # in place, version untouched
    assert conv.weight._version == v
    y1 = run()
    assert torch.allclose(y1, 2.0 * y0, rtol=1e-5, atol=1e-6), float((y1 - 2.0 * y0).abs().max())


def test_eval_plan_is_dropped_by_a_training_forward_in_eval_mode():
    """Fine-tuning with the model kept in .eval() (frozen BatchNorm statistics) never calls train(), which is what normally drops the folded /
    packed eval plan: a grad-enabled forward must drop it too, or the next no_grad forward answers with the weights from before the step."""
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    cfg = dict(channels=32, num_blocks=4)
    batch = make_batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.10, seed=1)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, **cfg)
    model.load_state_dict(random_state_dict(5, **cfg), strict=True)
    model = model.cuda().eval()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    with torch.no_grad():
        before = model(gb, return_loss=False)["semantic_prediction_logits"].clone()
    assert model._plan is not None
    loss, _ = model(gb, return_loss=True)                       # eval mode, grad enabled: the module-by-module path with running statistics
    loss.backward(); opt.step()
    with torch.no_grad():
        after = model(gb, return_loss=False)["semantic_prediction_logits"]
        fresh = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, **cfg)
        fresh.load_state_dict(model.state_dict(), strict=True)
        ref = fresh.cuda().eval()(gb, return_loss=False)["semantic_prediction_logits"]
    assert not torch.equal(before, after)
    assert torch.equal(after, ref)                              # the answer of a plan built from the updated weights


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fixed_modules_fine_tuning_step(dtype):
    """The reference's `fixed_modules` (tree_learn.py:66-72, train() override :105-112): the named modules get no gradients and their
    BatchNorms stay in eval mode while the rest trains.  One training step with input_conv / unet / output_layer fixed: their parameters and
    running statistics are untouched, and the heads' loss and gradients equal a plain-torch float64 computation on the backbone features
    the eval forward gives (the fixed backbone IS the eval backbone)."""
    import copy
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    from treelearn_amd.util.train import point_wise_loss_impl
    cfg = dict(channels=32, num_blocks=4)
    fixed = ['input_conv', 'unet', 'output_layer']
    batch = make_batch([make_tile(extent=12.0, voxel=0.1, n_trees=6, fill=0.10, seed=s) for s in (1, 2)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    model = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[256, 256, 512], voxel_size=0.1, fixed_modules=fixed, compute_dtype=dtype, **cfg)
    model.load_state_dict(random_state_dict(5, **cfg), strict=True)
    model = model.cuda()
    model.eval()
    with torch.no_grad():
        feats = model(gb, return_loss=False)["backbone_feats"].double()          # eval backbone (running statistics everywhere)
    model.train()
    for name in fixed:
        assert all(not m.training for m in getattr(model, name).modules() if isinstance(m, torch.nn.BatchNorm1d))
    assert all(m.training for m in model.semantic_linear.modules() if isinstance(m, torch.nn.BatchNorm1d))
    stats0 = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    heads = {n: copy.deepcopy(torch.nn.Sequential(*[m for m in getattr(model, n)])).double() for n in ("semantic_linear", "offset_linear")}
    model.zero_grad()
    loss, _ = model(gb, return_loss=True)
    loss.backward()
    for name, p in model.named_parameters():
        if name.split(".")[0] in fixed:
            assert p.grad is None, name
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
    for k, v in model.state_dict().items():
        if k in stats0:
            same = torch.equal(v, stats0[k])
            assert same == (k.split(".")[0] in fixed), k                             # fixed: untouched; heads: updated
    # plain torch, float64, on the eval backbone's features
    logits = heads["semantic_linear"](feats); offs = heads["offset_linear"](feats)
    sem, off = point_wise_loss_impl(logits, offs, gb["masks_sem"], gb["masks_off"], gb["semantic_labels"], gb["offset_labels"].double())
    from treelearn_amd.model.net import LOSS_MULTIPLIER_SEMANTIC
    ref_loss = sem * LOSS_MULTIPLIER_SEMANTIC + off
    ref_loss.backward()
    tol = 2e-4 if dtype == torch.float32 else 6e-2
    assert float(loss.detach()) == pytest.approx(float(ref_loss.detach()), rel=tol)
    for n in ("semantic_linear", "offset_linear"):
        for (pn, p), (_, q) in zip(getattr(model, n).named_parameters(), heads[n].named_parameters()):
            b = q.grad
            if float(b.abs().max()) < 1e-9:
                continue                                                             # a Linear bias in front of a BatchNorm: zero gradient
            e = float((p.grad.double() - b).abs().max() / b.abs().max())
            assert e < tol, (n, pn, e)
