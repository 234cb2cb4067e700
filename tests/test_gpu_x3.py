"""The split-bf16 ("bf16x3") contraction of fp32 convs (tl_conv_args.weight_x3; csrc/tl_conv_internal.h mma16_x3) against the exact fp32
kernels and the float64 oracle: fp32 storage, every product formed as alo.bhi + ahi.blo + ahi.bhi on the bf16 matrix cores.  The mode
serves the reference's fp32 inference (tree_learn/util/pipeline.py:86) inside the 1e-3 gate at a fraction of the fp32-MFMA time."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sparse_ops as osp
from treelearn_amd.synth import make_batch, make_tile


def _geom(extent=12.0, levels=3, seed=1):
    from treelearn_amd.geometry import build_geometry
    t = make_tile(extent=extent, voxel=0.1, n_trees=5, fill=0.1, seed=seed)
    xyz = torch.from_numpy(t["points"]).cuda()
    return build_geometry(xyz, torch.zeros(len(xyz), dtype=torch.int64, device="cuda"), 1, 0.1, levels, [500, 500, 1000])


def _packs(w):
    from treelearn_amd import ops
    exact = ops.pack_weight(w, torch.float32)
    ops.PACK_X3 = True
    try:
        x3 = ops.pack_weight(w, torch.float32)
    finally:
        ops.PACK_X3 = False
    assert getattr(x3, "_tl_x3", None) is not None and getattr(exact, "_tl_x3", None) is None
    return exact, x3


@pytest.mark.parametrize("cin,cout,kind,level", [(32, 32, "subm", 0), (64, 64, "subm", 0), (128, 64, "subm", 0), (96, 96, "subm", 0), (192, 96, "subm", 0), (128, 128, "subm", 0),
                                                 (64, 32, "subm", 0), (32, 64, "down", 0), (64, 96, "down", 0), (96, 128, "down", 1),
                                                 (96, 64, "inverse", 0), (128, 96, "inverse", 0), (128, 64, "1x1", 0), (64, 32, "inverse", 0)])
def test_x3_conv_vs_exact_fp32_and_float64(cin, cout, kind, level):
    from treelearn_amd import ops
    g = _geom()
    lv = g.levels[level]
    if kind == "inverse":                                  # SparseInverseConv3d: the coarse level's rows through the one-hot table
        table, n_out, n_in = lv.inv, lv.n, g.levels[level + 1].n
    elif kind == "1x1":
        table, n_out, n_in = None, lv.n, lv.n
    else:
        table, n_out = (lv.nbr, lv.n) if kind == "subm" else (lv.child, g.levels[level + 1].n)
        n_in = lv.n
    assert n_out > 16384                                   # the large-level kernels (direct / stream), not the small-level one
    K = table.shape[0] if table is not None else 1
    k = {27: 3, 8: 2, 1: 1}[K]
    gen = torch.Generator(device="cuda"); gen.manual_seed(cin * 1000 + cout)
    w = torch.randn((cout, k, k, k, cin), device="cuda", generator=gen) / (cin * K) ** 0.5
    x = torch.randn((n_in, cin), device="cuda", generator=gen)
    res = torch.randn((n_out, cout), device="cuda", generator=gen)
    oh = dict(one_hot=True) if kind == "inverse" else {}
    sc = torch.rand(cout, device="cuda", generator=gen) + 0.5; sh = torch.randn(cout, device="cuda", generator=gen)
    w_exact, w_x3 = _packs(w)
    y_exact = ops.conv_fwd(x, w_exact, table, n_out, residual=res, out_scale=sc, out_shift=sh, out_relu=True, **oh)
    y2 = torch.empty_like(y_exact)
    y_x3 = ops.conv_fwd(x, w_x3, table, n_out, residual=res, out_scale=sc, out_shift=sh, out_relu=True, out2=(y2, None, None, False), **oh)
    assert not torch.equal(y_exact, y_x3), "the split-bf16 kernel did not run (results are bit-identical to the exact fp32 kernel)"
    scale = float(y_exact.abs().max())
    assert float((y_exact - y_x3).abs().max()) / scale < 1e-4
    # float64 reference on a sample of rows: the x3 result is as close to it as 2^-15-per-product allows
    rows = torch.randperm(n_out, device="cuda", generator=gen)[:2048].sort().values
    sub = table[:, rows].T.contiguous().cpu().numpy() if table is not None else rows.cpu().numpy()[:, None].astype(np.int32)
    ref = osp.conv_table(x.double().cpu(), w.double().cpu(), sub).numpy() + res[rows].double().cpu().numpy()
    raw = y2[rows].double().cpu().numpy()
    assert np.abs(raw - ref).max() / np.abs(ref).max() < 5e-5
    act = np.maximum(ref * sc.double().cpu().numpy() + sh.double().cpu().numpy(), 0)
    assert np.abs(y_x3[rows].double().cpu().numpy() - act).max() / max(np.abs(act).max(), 1e-9) < 5e-5


def test_x3_forward_small_model_vs_exact_and_executor():
    """A whole forward in the parity-fast mode: close to the exact fp32 forward, identical through tl_forward and the Python-driven engine."""
    import os
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import random_state_dict
    b = make_batch([make_tile(extent=16.0, voxel=0.1, n_trees=10, fill=0.1, seed=3)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    outs = {}
    for mode in (torch.float32, "bf16x3"):
        m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=mode)
        m.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); m = m.cuda().eval()
        with torch.no_grad():
            outs[mode] = m(gb, return_loss=False)
            if mode == "bf16x3":
                os.environ["TL_EXEC"] = "0"
                try:
                    py = m(gb, return_loss=False)
                finally:
                    del os.environ["TL_EXEC"]
                for k in py:
                    assert torch.equal(py[k], outs[mode][k])
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        a, c = outs[torch.float32][k], outs["bf16x3"][k]
        assert float((a - c).abs().max()) / float(a.abs().max()) < 1e-3
    assert not torch.equal(outs[torch.float32]["backbone_feats"], outs["bf16x3"]["backbone_feats"])


def test_x3_wide_conv_runs_as_two_half_width_launches():
    """256 -> 128 (the decoder's 2C -> C conv of level 4): no split-bf16 instantiation that wide, so tl_pack_weight_x3 stores two half-width
    convs and tl_conv_fwd chains two 128 -> 128 launches through `out` (csrc/tl_conv.hip)."""
    from treelearn_amd import ops
    g = _geom()
    lv = g.levels[0]
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    w = torch.randn((128, 3, 3, 3, 256), device="cuda", generator=gen) / (256 * 27) ** 0.5
    x = torch.randn((lv.n, 256), device="cuda", generator=gen)
    sc = torch.rand(128, device="cuda", generator=gen) + 0.5; sh = torch.randn(128, device="cuda", generator=gen)
    w_exact, w_x3 = _packs(w)
    y_exact = ops.conv_fwd(x, w_exact, lv.nbr, lv.n, out_scale=sc, out_shift=sh, out_relu=True)
    y_x3 = ops.conv_fwd(x, w_x3, lv.nbr, lv.n, out_scale=sc, out_shift=sh, out_relu=True)
    assert not torch.equal(y_exact, y_x3)
    assert float((y_exact - y_x3).abs().max()) / float(y_exact.abs().max()) < 1e-4
    # with a residual or a second view the chained form does not apply: the exact kernel runs on the plain weights
    res = torch.randn((lv.n, 128), device="cuda", generator=gen)
    assert torch.equal(ops.conv_fwd(x, w_x3, lv.nbr, lv.n, residual=res), ops.conv_fwd(x, w_exact, lv.nbr, lv.n, residual=res))


@pytest.mark.parametrize("cin,cout", [(160, 160), (192, 192), (224, 224), (96, 64)])
def test_x3_small_level_kernel(cin, cout):
    """The small-level kernel (<= 16 384 rows: levels 5-7) in the split-bf16 form against its exact fp32 self and float64."""
    from treelearn_amd import ops
    g = _geom(extent=8.0, levels=3)
    lv = g.levels[2]                                       # a few thousand rows
    assert lv.n <= 16384
    gen = torch.Generator(device="cuda"); gen.manual_seed(cin + cout)
    w = torch.randn((cout, 3, 3, 3, cin), device="cuda", generator=gen) / (cin * 27) ** 0.5
    x = torch.randn((lv.n, cin), device="cuda", generator=gen)
    res = torch.randn((lv.n, cout), device="cuda", generator=gen)
    w_exact, w_x3 = _packs(w)
    y_exact = ops.conv_fwd(x, w_exact, lv.nbr, lv.n, residual=res)
    y_x3 = ops.conv_fwd(x, w_x3, lv.nbr, lv.n, residual=res)
    assert not torch.equal(y_exact, y_x3)
    assert float((y_exact - y_x3).abs().max()) / float(y_exact.abs().max()) < 1e-4
    sub = lv.nbr.T.contiguous().cpu().numpy()
    ref = osp.conv_table(x.double().cpu(), w.double().cpu(), sub).numpy() + res.double().cpu().numpy()
    assert np.abs(y_x3.double().cpu().numpy() - ref).max() / np.abs(ref).max() < 5e-5


# ------------------------------------------------------------------------------------------------ round 6: the staged-unit kernel for fp32 rows (level 1)
def _blk_geoms(extent, seeds):
    from treelearn_amd.geometry import build_geometry
    b = make_batch([make_tile(extent=extent, voxel=0.1, n_trees=6, fill=0.1, seed=s) for s in seeds])
    c, bi = b["coords"].cuda(), b["batch_ids"].cuda()
    can = build_geometry(c, bi, len(seeds), 0.1, 7, [500, 500, 1000])
    blk = build_geometry(c, bi, len(seeds), 0.1, 7, [500, 500, 1000], blocked=True)
    return can, blk


@pytest.mark.parametrize("residual,views,wide,prologue", [(False, 1, False, False), (True, 1, False, False), (True, 2, False, False), (False, 2, True, False),
                                                          (False, 1, True, True), (True, 1, False, True), (True, 2, True, True)])
def test_x3_blk_conv_vs_exact_fp32_and_float64(residual, views, wide, prologue):
    """tl_conv_blk_x3.hip (fp32 rows in the block-local order, staged once, split into hi | lo bf16 halves once, two 16-channel launches)
    against the exact fp32 kernel on the canonical table -- every epilogue / prologue form the level-1 dataflow of the engine uses (raw or
    activated view, residual, second view, column views of the 64-wide concat buffer, BatchNorm + ReLU at staging) -- within 1e-4 of it and
    within 5e-5 of a float64 evaluation on sampled rows."""
    from treelearn_amd import ops
    can, blk = _blk_geoms(14.0, [3, 4])
    n = can.levels[0].n
    assert n > 100_000
    gen = torch.Generator(device="cuda"); gen.manual_seed(17 + views + 2 * residual + 4 * wide)
    w = torch.randn((32, 3, 3, 3, 32), device="cuda", generator=gen) / (32 * 27) ** 0.5
    w_exact, w_x3 = _packs(w)
    r = blk.levels[0].nbr
    perm, o2n = r.perm.long(), r.o2n.long()
    ld = 64 if wide else 32
    xb = torch.randn((n, ld), device="cuda", generator=gen)
    x = xb[:, ld - 32:]
    res = torch.randn((n, 32), device="cuda", generator=gen) if residual else None
    sc = [torch.rand(32, device="cuda", generator=gen) + 0.5 for _ in range(3)]; sh = [torch.randn(32, device="cuda", generator=gen) * 0.3 for _ in range(3)]
    pro = dict(in_scale=sc[2], in_shift=sh[2], in_relu=True) if prologue else {}

    def run(wp, table, xin, rin):
        bufs = [torch.zeros(n, ld, device="cuda") for _ in range(views)]
        tgt = [b_[:, ld - 32:] for b_ in bufs]
        kw = dict(pro)
        if views == 2:
            kw["out2"] = (tgt[1], sc[1], sh[1], True)
        ops.conv_fwd(xin, wp, table, n, out=tgt[0], residual=rin, out_scale=sc[0] if views == 1 else None, out_shift=sh[0] if views == 1 else None,
                     out_relu=views == 1, **kw)
        return [t.clone() for t in tgt]

    if prologue:                                         # the exact kernels take the activated tensor (their producers write it as a second view)
        x_can = torch.relu(x * sc[2] + sh[2])
        exact = run.__wrapped__ if False else None
        bufs = [torch.zeros(n, 32, device="cuda") for _ in range(views)]
        kw = {}
        if views == 2:
            kw["out2"] = (bufs[1], sc[1], sh[1], True)
        ops.conv_fwd(x_can.contiguous(), w_exact, can.levels[0].nbr, n, out=bufs[0], residual=res, out_scale=sc[0] if views == 1 else None,
                     out_shift=sh[0] if views == 1 else None, out_relu=views == 1, **kw)
        a = bufs
    else:
        a = run(w_exact, can.levels[0].nbr, x, res)
        x_can = x
    xnew = torch.zeros(n, ld, device="cuda"); xnew[:, ld - 32:] = x[perm]
    before = ops.BLK_LAUNCHES
    b = [t[o2n] for t in run(w_x3, r, xnew[:, ld - 32:], res[perm].contiguous() if residual else None)]
    assert ops.BLK_LAUNCHES == before + 1
    for p_, q_ in zip(a, b):
        assert not torch.equal(p_, q_), "the split-bf16 staged kernel did not run"
        assert float((p_ - q_).abs().max()) / float(p_.abs().max()) < 1e-4
    rows = torch.randperm(n, device="cuda", generator=gen)[:2048].sort().values
    sub = can.levels[0].nbr[:, rows].T.contiguous().cpu().numpy()
    ref = osp.conv_table(x_can.double().cpu(), w.double().cpu(), sub).numpy() + (res[rows].double().cpu().numpy() if residual else 0.0)
    if views == 1:
        ref = np.maximum(ref * sc[0].double().cpu().numpy() + sh[0].double().cpu().numpy(), 0)
    assert np.abs(b[0][rows].double().cpu().numpy() - ref).max() / np.abs(ref).max() < 5e-5


def test_x3_forward_takes_the_block_local_level_and_stays_inside_the_gate():
    """The parity-fast forward runs level 1 in the block-local order (geometry without a canonical level-1 table, staged x3 kernel): same
    results through tl_forward and the Python-driven engine, within 2e-4 of the exact fp32 forward; TL_BLK_X3=0 restores the gather form."""
    import os
    from treelearn_amd import ops
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import random_state_dict
    b = make_batch([make_tile(extent=16.0, voxel=0.1, n_trees=10, fill=0.1, seed=3), make_tile(extent=11.0, voxel=0.1, n_trees=4, fill=0.1, seed=5)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    sd = random_state_dict(7, channels=32, num_blocks=7)

    def build(mode):
        m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=mode)
        m.load_state_dict(sd)
        return m.cuda().eval()

    with torch.no_grad():
        exact = build(torch.float32)(gb, return_loss=False)
        m3 = build("bf16x3")
        o_c = m3(gb, return_loss=False)
        ex = m3._plan._exec
        assert ex is not None and ex.last["blocked"], "bf16x3 must run level 1 in the block-local order"
        os.environ["TL_EXEC"] = "0"
        try:
            n0 = ops.BLK_LAUNCHES
            o_p = m3(gb, return_loss=False)
            assert ops.BLK_LAUNCHES - n0 >= 9
        finally:
            del os.environ["TL_EXEC"]
        os.environ["TL_BLK_X3"] = "0"
        try:
            o_g = build("bf16x3")(gb, return_loss=False)
        finally:
            del os.environ["TL_BLK_X3"]
    for k in ("backbone_feats", "semantic_prediction_logits", "offset_predictions"):
        assert torch.equal(o_c[k], o_p[k]), k
        s_ = float(exact[k].abs().max())
        assert float((o_c[k] - exact[k]).abs().max()) / s_ < 2e-4, k
        assert float((o_g[k] - exact[k]).abs().max()) / s_ < 2e-4, k
        assert not torch.equal(o_g[k], o_c[k]), k
