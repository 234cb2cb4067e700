"""GPU tests of the block-local level-1 path (csrc/tl_blk.hip + csrc/tl_conv_blk.hip; include/treelearn_hip.h `tl_blk`, tl_conv_args.blk_*):

geometry  order / units / halo lists / local rulebooks / presence masks bit-exact against the numpy restatement (oracle/blk.py), the staged
          form decodes to the canonical rulebook (which the oracle pins, tests/test_gpu_configs.py), down / inverse tables and v2p are the
          canonical ones carried into the new order; forced deep splitting (small halo bound); batches of tiles
conv      the staged-unit kernel is torch.equal to the gather kernel (tl_conv_direct) on the same rulebook: plain, residual, two / three
          views with BatchNorm affine + ReLU, column views of wider buffers, float16; on a mid-size tile and on THE config-2 tile
model     the fused forward with level 1 in block-local order against the canonical-order forward and against the CPU oracle
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import blk as ob
from oracle import voxel as ov
from treelearn_amd.synth import CONFIGS, make_batch, make_tile, random_state_dict


def _batch(extent, seeds, n_trees=8):
    return make_batch([make_tile(extent=extent, voxel=0.1, n_trees=n_trees, fill=0.1, seed=s) for s in seeds])


def _geoms(batch, vs=0.1, sshape=(500, 500, 1000), ref_table=False, **kw):
    from treelearn_amd.geometry import build_geometry
    c, b = batch["coords"].cuda(), batch["batch_ids"].cuda()
    can = build_geometry(c, b, int(batch["batch_size"]), vs, 7, list(sshape))
    blk = build_geometry(c, b, int(batch["batch_size"]), vs, 7, list(sshape), blocked=True, ref_table=ref_table, **kw)
    return can, blk


def _check_blocked_geometry(can, blk, halo_max):
    from treelearn_amd.geometry import BlockedRulebook
    assert blk.blocked and isinstance(blk.levels[0].nbr, BlockedRulebook)
    l0, r = can.levels[0], blk.levels[0].nbr
    n = l0.n
    coords = l0.coords.cpu().numpy(); nbr = l0.nbr.cpu().numpy()
    o = ob.build_fast(coords, nbr, halo_max)
    np.testing.assert_array_equal(r.perm.cpu().numpy(), o["perm"])
    np.testing.assert_array_equal(r.o2n.cpu().numpy(), o["o2n"])
    np.testing.assert_array_equal(r.coords_new.cpu().numpy(), o["coords_new"])
    np.testing.assert_array_equal(r.pmask.cpu().numpy(), o["pmask"])
    nu = int(r.counter[0]); assert int(r.counter[1]) == 0
    assert nu == len(o["units"])
    unit = r.unit[:nu].cpu().numpy()
    nch = (n + 63) // 64
    exp = np.array([[u[0], u[1], len(u[2]), 0] for u in o["units"]], np.int32)
    np.testing.assert_array_equal(unit, exp)                   # regular units: unit c = chunk c; appended pieces ascending by first row
    halo = r.halo.cpu().numpy()
    for lo, cnt, h in o["units"]:
        H = len(h); H16 = (H + 15) // 16 * 16
        got = halo[32 * lo:32 * lo + H16]
        assert np.array_equal(got[:H], h) and (got[H:] == -1).all(), (lo, cnt)
    np.testing.assert_array_equal(r.lrb.cpu().numpy().view(np.uint32), o["lrb"])
    # the staged form decodes to the canonical table carried into the new order
    np.testing.assert_array_equal(r.to_table().cpu().numpy().astype(np.int64), o["nn"])
    # tables that hold / are indexed by level-1 rows, and v2p
    o2n = torch.from_numpy(o["o2n"]).cuda(); perm = torch.from_numpy(o["perm"]).cuda()
    ch = l0.child.long()
    assert torch.equal(blk.levels[0].child.long(), torch.where(ch >= 0, o2n[ch.clamp(min=0)], ch))
    assert torch.equal(blk.levels[0].parent, l0.parent[perm])
    assert torch.equal(blk.levels[0].inv, l0.inv[:, perm])
    assert torch.equal(blk.v2p, o2n[can.v2p])
    for a, b in zip(can.levels[1:], blk.levels[1:]):                                          # the other levels are untouched
        assert torch.equal(a.coords, b.coords) and torch.equal(a.nbr, b.nbr)
        if a.child is not None:
            assert torch.equal(a.child, b.child) and torch.equal(a.inv, b.inv) and torch.equal(a.parent, b.parent)
    return o


@pytest.mark.parametrize("halo_max", [126, 40, 26])
def test_blocked_geometry_vs_oracle(halo_max):
    """A 14 m tile (150 k voxels): everything bit-exact; halo bounds 40 and 26 force two to five levels of halving."""
    import treelearn_amd.geometry as G
    batch = _batch(14.0, [3])
    old = G.BLK_HALO_MAX
    G.BLK_HALO_MAX = halo_max
    try:
        can, blk = _geoms(batch)
    finally:
        G.BLK_HALO_MAX = old
    o = _check_blocked_geometry(can, blk, halo_max)
    assert max(len(u[2]) for u in o["units"]) <= halo_max
    if halo_max < 126:
        assert len(o["units"]) > (can.levels[0].n + 63) // 64


def test_blocked_geometry_batch_of_unequal_tiles_and_ref_table():
    """Three tiles of different size in one batch (blocks never span batch elements); with ref_table the canonical level-1 table rides along."""
    batch = make_batch([make_tile(extent=e, voxel=0.1, n_trees=4, fill=0.1, seed=s) for e, s in ((9.0, 1), (13.0, 2), (6.0, 5))])
    can, blk = _geoms(batch, ref_table=True)
    _check_blocked_geometry(can, blk, 126)
    assert torch.equal(blk.levels[0].nbr_ref, can.levels[0].nbr) and torch.equal(blk.levels[0].coords, can.levels[0].coords)
    b = blk.levels[0].nbr.coords_new[:, 0]
    assert bool((b[1:] >= b[:-1]).all())


def test_blocked_geometry_small_tile_switches_off():
    batch = _batch(2.5, [1], n_trees=1)
    can, blk = _geoms(batch)
    assert can.levels[0].n < 16384
    assert not blk.blocked and torch.equal(blk.levels[0].nbr, can.levels[0].nbr) and torch.equal(blk.v2p, can.v2p)
    can, blk = _geoms(batch, blk_min_rows=1)                  # forced: works at any size
    _check_blocked_geometry(can, blk, 126)


def _conv_pair(can, blk, dtype, residual, views, wide, seed=0):
    """The same conv on the canonical table (gather kernel) and on the block-local form; returns both results in canonical order."""
    from treelearn_amd import ops
    n = can.levels[0].n
    dev = can.v2p.device
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.randn(n, 32, generator=g) * 0.7).to(dtype).to(dev)
    res = (torch.randn(n, 32, generator=g)).to(dtype).to(dev) if residual else None
    w = ops.pack_weight((torch.randn(32, 3, 3, 3, 32, generator=g) * 0.08).to(dev), dtype)
    aff = [((torch.rand(32, generator=g) + 0.5).to(dev), (torch.randn(32, generator=g) * 0.3).to(dev)) for _ in range(3)]
    r = blk.levels[0].nbr
    perm = r.perm.long()

    def run(table, xin, rin):
        ld = 64 if wide else 32
        outs = []
        bufs = [torch.zeros(n, ld, dtype=dtype, device=dev) for _ in range(views)]
        tgt = [b[:, ld - 32:] for b in bufs]                                   # column views of wider buffers when `wide`
        xin_ = xin
        if wide:
            xb = torch.zeros(n, 64, dtype=dtype, device=dev); xb[:, 32:] = xin; xin_ = xb[:, 32:]
        kw = {}
        if views >= 2:
            kw["out2"] = (tgt[1], aff[1][0], aff[1][1], True)
        if views >= 3:
            kw["out3"] = (tgt[2], None, None, True)
        ops.conv_fwd(xin_, w, table, n, out=tgt[0], residual=rin, out_scale=aff[0][0] if views == 1 else None,
                     out_shift=aff[0][1] if views == 1 else None, out_relu=views == 1, **kw)
        return [t.clone() for t in tgt]

    a = run(can.levels[0].nbr, x, res)
    b = run(r, x[perm].contiguous(), res[perm].contiguous() if residual else None)
    o2n = r.o2n.long()
    return a, [t[o2n] for t in b]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("residual,views,wide", [(False, 1, False), (True, 1, False), (True, 2, False), (True, 3, True), (False, 2, True), (False, 3, False)])
def test_blk_conv_equals_gather_kernel(dtype, residual, views, wide):
    """Bit-identical to tl_conv_direct on a 14 m tile, every epilogue form of the inference engine."""
    batch = _batch(14.0, [3])
    can, blk = _geoms(batch)
    a, b = _conv_pair(can, blk, dtype, residual, views, wide)
    for i, (p, q) in enumerate(zip(a, b)):
        assert torch.equal(p, q), f"view {i}: {int((p != q).any(dim=1).sum())} rows differ, max |diff| {float((p.float() - q.float()).abs().max())}"
    assert float(a[0].float().abs().max()) > 0.1


def test_blk_conv_deep_split_units():
    """Units of every size (halo bound 26 -> pieces down to a few rows): same results."""
    import treelearn_amd.geometry as G
    batch = _batch(10.0, [4])
    old = G.BLK_HALO_MAX
    G.BLK_HALO_MAX = 26
    try:
        can, blk = _geoms(batch, blk_min_rows=1)
    finally:
        G.BLK_HALO_MAX = old
    a, b = _conv_pair(can, blk, torch.bfloat16, True, 2, False)
    for p, q in zip(a, b):
        assert torch.equal(p, q)


@pytest.fixture(scope="module")
def tile2(tile2_batch):
    return tile2_batch


def test_config2_blocked_geometry_and_conv_on_the_full_tile(tile2):
    """THE config-2 tile (1.85 M voxels): blocked geometry against the oracle, the level-1 conv torch.equal to the gather kernel."""
    can, blk = _geoms(tile2)
    _check_blocked_geometry(can, blk, 126)
    a, b = _conv_pair(can, blk, torch.bfloat16, True, 2, False)
    for p, q in zip(a, b):
        assert torch.equal(p, q)


def _model(dtype, use_feats=False, seed=7, settle_on=None):
    """`settle_on`: a batch on which the BatchNorm running statistics are re-estimated first (train-mode passes in bf16): a random-init net
    with arbitrary running statistics reaches 1e5 in places, beyond float16's range (tests/test_gpu_f16.py `_trained_like`)."""
    from treelearn_amd.model import TreeLearn
    m = TreeLearn(use_feats=use_feats, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1,
                  compute_dtype=torch.bfloat16 if settle_on is not None else dtype)
    m.load_state_dict(random_state_dict(seed, channels=32, num_blocks=7), strict=True)
    m = m.cuda()
    if settle_on is not None:
        m.train()
        with torch.no_grad():
            for _ in range(2):
                m(settle_on, return_loss=False)
        m.compute_dtype = dtype
    return m.eval()


def _fwd(m, batch, blocked):
    old = os.environ.get("TL_BLK")
    os.environ["TL_BLK"] = "1" if blocked else "0"
    try:
        with torch.no_grad():
            out = m(batch, return_loss=False)
    finally:
        if old is None:
            os.environ.pop("TL_BLK")
        else:
            os.environ["TL_BLK"] = old
    return {k: v.float().cpu() for k, v in out.items()}


@pytest.mark.parametrize("use_feats", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_forward_blocked_vs_canonical_order(dtype, use_feats, monkeypatch):
    """Same tile, level 1 in block-local vs canonical order, with the two-view producers of the canonical path (TL_BLK_PRO=0).  Everything
    but the 64 -> 32 decoder conv (run as two input-channel halves) is bit-identical, so the outputs agree to 16-bit rounding of one layer;
    rows come back in point order either way."""
    from treelearn_amd import ops
    monkeypatch.setenv("TL_BLK_PRO", "0")
    batch = _batch(16.0, [5])
    m = _model(dtype, use_feats, settle_on=batch if dtype == torch.float16 else None)
    ops.PROFILE = []
    a = _fwd(m, batch, True)
    fam = [type(meta["table"]).__name__ for _, _, meta in ops.PROFILE]
    ops.PROFILE = None
    assert fam.count("BlockedRulebook") >= 9                     # the level-1 convs ran on the staged-unit kernel
    b = _fwd(m, batch, False)
    for k in a:
        ref = b[k].abs().max()
        assert torch.isfinite(a[k]).all()
        assert float((a[k] - b[k]).abs().max() / ref) < 2e-2, k
        assert float((a[k] - b[k]).abs().mean() / b[k].abs().mean()) < 2e-3, k


@pytest.mark.parametrize("dtype,use_feats,seeds", [(torch.bfloat16, False, [5]), (torch.float16, False, [5]), (torch.bfloat16, True, [5]),
                                                   (torch.bfloat16, False, [6, 7])])
def test_forward_prologue_at_staging_vs_fp32(dtype, use_feats, seeds, monkeypatch):
    """The default level-1 form stores each conv result once (rounded to 16 bits) and applies BatchNorm + ReLU when the consumer stages the
    row -- the reference's own order of operations (conv output tensor, then BatchNorm1d + ReLU on it, tree_learn/model/blocks.py:57-70) --
    where the two-view form rounds the activated value from the fp32 accumulator.  Neither is `the' 16-bit result, so both are held
    against the fp32 forward: the staged form is no further from it than the canonical-order 16-bit forward (factor 1.25)."""
    from treelearn_amd import ops
    batch = _batch(16.0 if len(seeds) == 1 else 12.0, seeds)           # (the last case: a batch of two tiles)
    m = _model(torch.bfloat16, use_feats, settle_on=batch)
    m.compute_dtype = torch.float32
    ref = _fwd(m, batch, False)
    m.compute_dtype = dtype
    can = _fwd(m, batch, False)
    monkeypatch.setenv("TL_BLK_PRO", "1")
    ops.PROFILE = []
    stg = _fwd(m, batch, True)
    pro = [bool(meta.get("in_scale")) for _, _, meta in ops.PROFILE if type(meta["table"]).__name__ == "BlockedRulebook"]
    ops.PROFILE = None
    assert len(pro) >= 9 and sum(pro) >= 5                       # conv1 of the four blocks + both halves of the 64 -> 32 conv
    for k in ref:
        e_can = float((can[k] - ref[k]).abs().mean() / ref[k].abs().mean())
        e_stg = float((stg[k] - ref[k]).abs().mean() / ref[k].abs().mean())
        m_can = float((can[k] - ref[k]).abs().max() / ref[k].abs().max())
        m_stg = float((stg[k] - ref[k]).abs().max() / ref[k].abs().max())
        assert torch.isfinite(stg[k]).all()
        assert e_stg < 1.25 * e_can + 1e-4, (k, e_stg, e_can)
        assert m_stg < 1.5 * m_can + 1e-3, (k, m_stg, m_can)


def test_forward_blocked_vs_oracle_bf16():
    """The block-local bf16 forward against the fp32 CPU oracle forward (the bf16 tolerance of the canonical path: 5e-2)."""
    from oracle import model as om
    batch = _batch(12.0, [2], n_trees=5)
    m = _model(torch.bfloat16)
    a = _fwd(m, batch, True)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    o = om.forward(sd, batch["coords"].numpy(), batch["input_feats"].numpy(), batch["batch_ids"].numpy(), 1, voxel_size=0.1, num_blocks=7,
                   spatial_shape=[500, 500, 1000])
    for k in ("semantic_prediction_logits", "offset_predictions"):
        ref = o[k].double().numpy()
        err = np.abs(a[k].numpy() - ref).max() / np.abs(ref).max()
        assert err < 5e-2, (k, err)


# =============================================================================================== training on the block-local level
def test_blk_conv_training_epilogues_equal_gather_kernel():
    """The staged-unit kernel's training epilogues (tl_conv_args.epi_mode) against the gather kernel's on the same rulebook: TL_EPI_STATS
    (with a residual) and TL_EPI_BN_BWD give the same stored tensor bit for bit and the same finished reductions (per-lane fp32 partial sums
    here, per-tile there: 1e-5)."""
    from treelearn_amd import ops
    batch = _batch(14.0, [3])
    can, blk = _geoms(batch)
    n = can.levels[0].n
    dev = can.v2p.device
    r = blk.levels[0].nbr
    perm, o2n = r.perm.long(), r.o2n.long()
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.randn(n, 32, generator=g) * 0.7).bfloat16().to(dev)
    res = torch.randn(n, 32, generator=g).bfloat16().to(dev)
    w = ops.pack_weight((torch.randn(32, 3, 3, 3, 32, generator=g) * 0.08).to(dev), torch.bfloat16)
    xb = torch.randn(n, 32, generator=g).bfloat16().to(dev)                               # the BatchNorm input of the BN_BWD mode
    st = torch.stack([torch.randn(32, generator=g) * 0.1, torch.rand(32, generator=g) + 0.5, torch.rand(32, generator=g) + 0.5,
                      torch.randn(32, generator=g) * 0.2]).to(dev).contiguous()            # mean, rstd, scale, shift
    for epi_c, epi_b, kw_c, kw_b in (("stats", "stats", dict(residual=res), dict(residual=res[perm].contiguous())),
                                     (("bn_bwd", xb, st, True), ("bn_bwd", xb[perm].contiguous(), st, True), {}, {})):
        a = ops.conv_fwd(x, w, can.levels[0].nbr, n, epi=epi_c, **kw_c)
        b = ops.conv_fwd(x[perm].contiguous(), w, r, n, epi=epi_b, **kw_b)
        assert a is not None and b is not None
        assert torch.equal(a[0], b[0][o2n]), epi_c if isinstance(epi_c, str) else epi_c[0]
        sa = a[1][:a[2]].sum(0); sb = b[1][:b[2]].sum(0)                                    # [2, 32] fp64 totals
        assert float((sa - sb).abs().max() / sa.abs().max()) < 1e-5


def _train_grads(blocked, gb, dtype=torch.bfloat16):
    from treelearn_amd.model import TreeLearn
    old = os.environ.get("TL_BLK_TRAIN")
    os.environ["TL_BLK_TRAIN"] = "1" if blocked else "0"
    try:
        m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=dtype)
        m.load_state_dict(random_state_dict(7, channels=32, num_blocks=7), strict=True)
        m = m.cuda().train()
        loss, ld = m(gb, return_loss=True)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {n: p.grad.detach().double().flatten() for n, p in m.named_parameters()}, \
            m.output_layer[0].running_mean.detach().clone()
    finally:
        if old is None:
            os.environ.pop("TL_BLK_TRAIN")
        else:
            os.environ["TL_BLK_TRAIN"] = old


def test_training_step_on_the_block_local_level():
    """A mixed-precision training step (two 16 m crops) with level 1 in the block-local order against the canonical-order step: same
    loss (forward results are identical row for row; BatchNorm sums differ in summation order only), every gradient tensor at cosine
    >= 0.999 of the canonical step's, running statistics equal to 1e-5."""
    from treelearn_amd import ops
    batch = make_batch([make_tile(extent=16.0, voxel=0.1, n_trees=10, fill=0.10, seed=s) for s in (5, 6)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    n0 = ops.BLK_LAUNCHES
    lb, gbk, rmb = _train_grads(True, gb)
    assert ops.BLK_LAUNCHES - n0 >= 16, ops.BLK_LAUNCHES - n0                              # level-1 forward and input-gradient convs
    n0 = ops.BLK_LAUNCHES
    lc, gc, rmc = _train_grads(False, gb)
    assert ops.BLK_LAUNCHES == n0
    assert lb == pytest.approx(lc, rel=1e-3)
    assert float((rmb - rmc).abs().max()) < 1e-5 * float(rmc.abs().max()) + 1e-6
    nmax = max(float(v.norm()) for v in gc.values())
    worst = (1.0, None)
    for nme, b in gc.items():
        if float(b.norm()) <= 1e-5 * nmax:
            continue
        a = gbk[nme]
        assert bool(torch.isfinite(a).all()), nme
        cos = float(a @ b / (a.norm() * b.norm()))
        worst = min(worst, (cos, nme))
        assert abs(float(a.norm() / b.norm()) - 1) < 0.05, nme
    print("blocked vs canonical training step: worst cosine", worst)
    assert worst[0] >= 0.999, worst


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("residual", [False, True])
def test_blk_conv_prologue_at_staging(dtype, residual):
    """tl_conv_fwd's gather-side prologue relu(x * in_scale + in_shift) on the staged-unit kernel (applied once per staged row in LDS) equals
    the same conv over a materialised activated tensor (tl_affine_relu: the same fmaf / max / round), bit for bit; also through column views
    of a wider buffer, as the skip concat's halves come."""
    from treelearn_amd import ops
    batch = _batch(14.0, [3])
    _, blk = _geoms(batch)
    r = blk.levels[0].nbr
    n = r.n
    dev = r.unit.device
    g = torch.Generator(device="cpu").manual_seed(2)
    wide = torch.zeros(n, 64, dtype=dtype, device=dev)
    wide[:, 32:] = (torch.randn(n, 32, generator=g) * 0.7).to(dtype).to(dev)
    x = wide[:, 32:]
    res = torch.randn(n, 32, generator=g).to(dtype).to(dev) if residual else None
    w = ops.pack_weight((torch.randn(32, 3, 3, 3, 32, generator=g) * 0.08).to(dev), dtype)
    sc, sh = (torch.rand(32, generator=g) + 0.5).to(dev), (torch.randn(32, generator=g) * 0.3).to(dev)
    osc, osh = (torch.rand(32, generator=g) + 0.5).to(dev), (torch.randn(32, generator=g) * 0.3).to(dev)
    act = ops.affine_relu(x.contiguous(), sc, sh, True)
    a = ops.conv_fwd(act, w, r, n, residual=res, out_scale=osc, out_shift=osh, out_relu=True)
    b = ops.conv_fwd(x, w, r, n, in_scale=sc, in_shift=sh, in_relu=True, residual=res, out_scale=osc, out_shift=osh, out_relu=True)
    assert torch.equal(a, b), int((a != b).any(dim=1).sum())
    assert float(a.float().abs().max()) > 0.1


# =============================================================================================== inverse conv walking the coarse rows
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_inverse_conv_scatter_form_equals_gather_form(dtype):
    """tl_conv_args.table_scatter (csrc/tl_conv_up.hip): the inverse convs of levels 1-4 computed per COARSE row and scattered to the
    children against the gather forms on the same rulebooks (canonical and block-local level 1) -- one product per output element, same
    contraction order: bit-identical; one and two views (affine + ReLU), column views of the skip-concat buffer."""
    from treelearn_amd import ops
    batch = _batch(16.0, [5])
    can, blk = _geoms(batch)
    g = torch.Generator(device="cpu").manual_seed(4)
    widths = [32, 64, 96, 128, 160]
    served = 0
    for geom in (can, blk):
        for li in range(4):
            lv, nxt = geom.levels[li], geom.levels[li + 1]
            ci, co = widths[li + 1], widths[li]
            dev = lv.inv.device
            x = (torch.randn(nxt.n, ci, generator=g) * 0.5).to(dtype).to(dev)
            w = ops.pack_weight((torch.randn(co, 2, 2, 2, ci, generator=g) * 0.1).to(dev), dtype)
            sc, sh = (torch.rand(co, generator=g) + 0.5).to(dev), (torch.randn(co, generator=g) * 0.3).to(dev)
            for two in (False, True):
                outs = []
                for scatter in (None, lv.child):
                    cat = torch.zeros(lv.n, 2 * co, dtype=dtype, device=dev)
                    act = torch.zeros(lv.n, 2 * co, dtype=dtype, device=dev)
                    ops.PROFILE = None
                    ops.conv_fwd(x, w, lv.inv, lv.n, out=cat[:, co:], one_hot=True, scatter=scatter,
                                 out2=(act[:, co:], sc, sh, True) if two else None)
                    outs.append((cat, act))
                assert torch.equal(outs[0][0], outs[1][0]), (li, two)
                assert torch.equal(outs[0][1], outs[1][1]), (li, two)
                assert float(outs[1][0].float().abs().max()) > 0.1
                served += 1
    assert served == 16


def test_no_ones_table_knob_with_a_blocked_geometry(monkeypatch):
    """TL_NO_ONES_TABLE=1 is documented as a knob that leaves the output unchanged: with the default flags (all-ones input, block-local
    level 1, no canonical table built) the forward must run and give the same result (advisor, round 4)."""
    from treelearn_amd.model import TreeLearn
    from treelearn_amd.synth import make_batch, make_tile, random_state_dict
    b = make_batch([make_tile(extent=14.0, voxel=0.1, n_trees=8, fill=0.1, seed=4)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    m = TreeLearn(use_feats=False, use_coords=False, spatial_shape=[500, 500, 1000], voxel_size=0.1, compute_dtype=torch.bfloat16)
    m.load_state_dict(random_state_dict(7, channels=32, num_blocks=7)); m = m.cuda().eval()
    with torch.no_grad():
        ref = m(gb, return_loss=False)
        monkeypatch.setenv("TL_NO_ONES_TABLE", "1")
        out = m(gb, return_loss=False)                      # (the knob also keeps the forward on the Python-driven engine)
        bb, v2p = m.forward_backbone(coords=gb["coords"], input_feats=gb["input_feats"], batch_ids=gb["batch_ids"], batch_size=1)
        out2 = m.forward_head(bb, v2p)
    for k in ("semantic_prediction_logits", "offset_predictions", "backbone_feats"):
        assert torch.equal(ref[k], out[k]) and torch.equal(ref[k], out2[k])


def test_blk_training_conv_with_the_batchnorm_at_staging():
    """Training forward of a block-local level-1 layer with its BatchNorm + ReLU applied when the conv stages its rows (k_conv_blk<.., TR, PRO>;
    autograd._staged_bn_conv): the stored tensor and the statistics partial sums equal the apply pass (tl_affine_relu) + the plain training
    launch bit for bit, with and without a residual; the 64 -> 32 conv as two staged halves equals the two launches on the activated tensor."""
    from treelearn_amd import ops
    batch = _batch(14.0, [3])
    _, blk = _geoms(batch)
    r = blk.levels[0].nbr
    n = r.n
    dev = r.unit.device
    g = torch.Generator(device="cpu").manual_seed(2)
    x = (torch.randn(n, 64, generator=g) * 0.7).bfloat16().to(dev)
    res = torch.randn(n, 32, generator=g).bfloat16().to(dev)
    sc = (torch.rand(64, generator=g) + 0.5).to(dev); sh = (torch.randn(64, generator=g) * 0.3).to(dev)
    w = ops.pack_weight((torch.randn(32, 3, 3, 3, 32, generator=g) * 0.08).to(dev), torch.bfloat16)
    w2 = ops.pack_weight((torch.randn(32, 3, 3, 3, 32, generator=g) * 0.08).to(dev), torch.bfloat16)
    x0, x1 = x[:, :32], x[:, 32:]                                       # column views of a wider matrix, as the skip concat's halves are
    a0 = ops.affine_relu(x0, sc[:32], sh[:32], True); a1 = ops.affine_relu(x1, sc[32:], sh[32:], True)
    for kw in ({}, dict(residual=res)):
        ref = ops.conv_fwd(a0, w, r, n, epi="stats", **kw)
        got = ops.conv_fwd(x0, w, r, n, in_scale=sc[:32], in_shift=sh[:32], in_relu=True, epi="stats", **kw)
        assert ref is not None and got is not None
        assert torch.equal(ref[0], got[0]) and ref[2] == got[2] and torch.equal(ref[1][:ref[2]], got[1][:got[2]])
    part_ref = ops.conv_fwd(a0, w, r, n)
    ref = ops.conv_fwd(a1, w2, r, n, residual=part_ref, epi="stats")
    part = ops.conv_fwd(x0, w, r, n, in_scale=sc[:32], in_shift=sh[:32], in_relu=True)
    got = ops.conv_fwd(x1, w2, r, n, residual=part, in_scale=sc[32:], in_shift=sh[32:], in_relu=True, epi="stats")
    assert torch.equal(part, part_ref) and torch.equal(ref[0], got[0]) and torch.equal(ref[1][:ref[2]], got[1][:got[2]])


def test_training_step_staged_batchnorm_equals_the_apply_pass(monkeypatch):
    """A whole mixed-precision step with the level-1 BatchNorms applied at staging (opt-in, TL_BLK_TRAIN_PRO=1) against the default step with
    the apply passes: the 32 -> 32 layers are bit-identical (test above); the 64 -> 32 decoder conv as two staged halves rounds its half sums
    once more, which every gradient downstream of it feels at the level of one bf16 rounding -- loss within 1e-3, every gradient at cosine
    >= 0.995 (measured: 0.9989 at worst, the level-2 stride-2 conv's weight)."""
    from treelearn_amd import autograd
    b = make_batch([make_tile(extent=14.0, voxel=0.1, n_trees=8, fill=0.1, seed=s) for s in (5, 6)])
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    monkeypatch.setattr(autograd, "STAGE_TRAIN", True)
    l1, g1, _ = _train_grads(True, gb)
    monkeypatch.setattr(autograd, "STAGE_TRAIN", False)
    l0, g0, _ = _train_grads(True, gb)
    assert abs(l1 - l0) <= 1e-3 * abs(l0)
    nmax = max(float(v.norm()) for v in g0.values())
    for k in g0:
        a, c = g0[k], g1[k]
        if float(a.norm()) > 1e-3 * nmax:                   # (a Linear bias in front of a BatchNorm has a zero gradient: pure rounding noise)
            assert float((a * c).sum() / (a.norm() * c.norm())) >= 0.995, k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["tile", "deep split", "batch of two", "64 -> 32"])
def test_blk_weight_gradient_equals_the_plain_table_kernel(dtype, case, monkeypatch):
    """tl_conv_wgrad_blk (level 1 of the training step: units staged once, 27 taps contracted against the staged rows) against tl_conv_wgrad over
    the SAME rulebook as a plain table (tl_blk.nn) and against float64: same 16-bit operands, fp32 accumulation in a different order -> 2e-6 of
    the largest entry.  Units of every size (halo bound 26), a batch of two tiles, the 64 -> 32 decoder conv as its two input halves, both
    output layouts."""
    import treelearn_amd.geometry as G
    from treelearn_amd import ops
    batch = _batch(14.0, [3, 5] if case == "batch of two" else [3])
    old = G.BLK_HALO_MAX
    if case == "deep split":
        G.BLK_HALO_MAX = 26
    try:
        _, blk = _geoms(batch, blk_min_rows=1, nn_table=True)
    finally:
        G.BLK_HALO_MAX = old
    r = blk.levels[0].nbr
    n, ci = r.n, (64 if case == "64 -> 32" else 32)
    gen = torch.Generator(device="cpu").manual_seed(11)
    x = (torch.randn(n, ci, generator=gen) * 0.7).to(dtype).cuda()
    g = (torch.randn(n, 32, generator=gen) * 0.3).to(dtype).cuda()
    for ref_layout in (False, True):
        monkeypatch.setattr(ops, "WGRAD_BLK", True)
        a = ops._conv_wgrad_blk(x, g, r, ref_layout) if ci == 64 else ops.conv_wgrad(x, g, r, n, 27, ref_layout=ref_layout)     # (64 -> 32 is not routed there by default)
        monkeypatch.setattr(ops, "WGRAD_BLK", False)
        b = ops.conv_wgrad(x, g, r, n, 27, ref_layout=ref_layout)
        assert a.shape == b.shape and a._tl_ref_layout == b._tl_ref_layout == ref_layout
        scale = float(b.abs().max())
        assert scale > 1.0 and float((a - b).abs().max()) < 2e-6 * scale * 27, (case, ref_layout, float((a - b).abs().max()), scale)
    # float64 on the host for a few taps (a is in the reference layout [Cout][K][Cin] here)
    tab = r.nn_table.long().cpu()
    xd, gd = x.double().cpu(), g.double().cpu()
    for k in (0, 13, 26):
        t = tab[k]; ok = t >= 0
        ref = gd[ok].T @ xd[t[ok]]
        assert float((a[:, k, :].double().cpu() - ref).abs().max()) < 1e-5 * float(ref.abs().max()) + 1e-4, k


def test_blk_weight_gradient_entry_point_refuses_what_it_does_not_cover():
    """tl_conv_wgrad_blk answers TL_ERR_UNSUPPORTED (not a wrong gradient) for fp32 rows, other widths and unaligned rows -- ops.conv_wgrad then
    takes tl_conv_wgrad over the plain table, and says so when the geometry was built without it."""
    from treelearn_amd import _hip, ops
    _, blk = _geoms(_batch(10.0, [2]), blk_min_rows=1, nn_table=True)
    r = blk.levels[0].nbr
    n = r.n
    L = _hip.lib()
    ws = torch.empty(int(L.tl_conv_wgrad_blk_ws_floats()), dtype=torch.float32, device="cuda")
    gw = torch.empty(27 * 32 * 32, dtype=torch.float32, device="cuda")

    def call(x, g, dt, ci=32, co=32):
        return L.tl_conv_wgrad_blk(_hip.ptr(x), x.stride(0), _hip.ptr(g), g.stride(0), dt, _hip.ptr(r.unit), _hip.ptr(r.counter), _hip.ptr(r.halo), _hip.ptr(r.lrb), n, ci, co,
                                   _hip.ptr(gw), 0, _hip.ptr(ws), _hip.stream())
    xb = torch.randn(n, 40, device="cuda").bfloat16(); gb = torch.randn(n, 32, device="cuda").bfloat16()
    assert call(xb[:, :32], gb, _hip.dtype_code(torch.bfloat16)) == _hip.TL_OK
    assert call(xb.float()[:, :32].contiguous(), gb.float(), _hip.dtype_code(torch.float32)) == _hip.TL_ERR_UNSUPPORTED
    assert call(xb[:, :32], gb, _hip.dtype_code(torch.bfloat16), ci=16) == _hip.TL_ERR_UNSUPPORTED
    assert call(xb[:, 4:36], gb, _hip.dtype_code(torch.bfloat16)) == _hip.TL_ERR_UNSUPPORTED               # rows not 16-B aligned
    assert call(torch.randn(n, 36, device="cuda").bfloat16()[:, :32], gb, _hip.dtype_code(torch.bfloat16)) == _hip.TL_ERR_UNSUPPORTED   # row pitch not a multiple of 8
    torch.cuda.synchronize()
    # fp32 rows through ops: the plain table serves them
    a = ops.conv_wgrad(xb[:, :32].float().contiguous(), gb.float(), r, n, 27)
    b = ops.conv_wgrad(xb[:, :32].contiguous(), gb, r, n, 27)
    assert float((a - b).abs().max()) < 1e-3 * float(a.abs().max())
    _, bare = _geoms(_batch(10.0, [2]), blk_min_rows=1)
    with pytest.raises(ValueError, match="nn_table"):
        ops.conv_wgrad(xb[:, :32].float().contiguous(), gb.float(), bare.levels[0].nbr, n, 27)
