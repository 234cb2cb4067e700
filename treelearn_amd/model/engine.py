"""Fused inference engine: walks the TreeLearn module tree and issues HIP calls with eval-mode
BatchNorm+ReLU, residual adds and the skip concat folded into the sparse-conv kernels.

Dataflow ("pre-activated" form, default).  BatchNorm1d+ReLU always sits in FRONT of a conv in the
reference (blocks.py:55-70,102-123), i.e. it would have to be applied to every gathered row -- up to 27x
per voxel.  Instead the PRODUCER of a tensor writes, in its epilogue, every view its consumers need:
    raw y                      (residual branch, blocks.py:76; 1x1 i_branch input; skip concat, blocks.py:146)
    relu(bn_next(y))           (gather source of the next conv -- gathers become pure copies)
through tl_conv_fwd's out / out2 / out3.  The skip concat is two column views of one [N, 2C] buffer
(raw and activated variants), written by the encoder's last block and by the inverse conv.

Per ResidualBlock (blocks.py:55-79):   t = conv1(x_act) [epilogue bn3+relu];  y = conv2(t) + i_branch(x_raw)
Per UBlock (blocks.py:137-149):        blocks -> down conv -> child UBlock -> inverse conv -> tail blocks

`TL_ENGINE=prologue` selects the older form (BN+ReLU applied as a gather-side prologue) for A/B runs.
"""
import os

import torch
from torch import nn

from .. import ops
from ..geometry import BlockedRulebook, TileGeometry


def _bn_affine(bn: nn.BatchNorm1d):
    """eval-mode BatchNorm1d as y = x * scale + shift (fp32)."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale.contiguous(), shift.contiguous()


class View:
    """One requested view of a conv result: optional destination (column view), affine, relu."""
    __slots__ = ("buf", "scale", "shift", "relu")

    def __init__(self, scale=None, shift=None, relu=False, buf=None):
        self.buf, self.scale, self.shift, self.relu = buf, scale, shift, relu


RAW = lambda buf=None: View(buf=buf)                                   # noqa: E731
ACT = lambda aff, buf=None: View(aff[0], aff[1], True, buf)            # noqa: E731


def _conv_views(x, w, table, n, views, residual=None, one_hot=False, all_ones=False, scatter=None):
    """Run one conv producing up to three views; returns the list of result tensors (same order)."""
    outs = [v.buf if v.buf is not None else torch.empty((n, w.shape[1]), dtype=x.dtype, device=x.device) for v in views]
    v0 = views[0]
    extra = [(o, v.scale, v.shift, v.relu) for o, v in zip(outs[1:], views[1:])]
    ops.conv_fwd(x, w, table, n, out=outs[0], residual=residual, out_scale=v0.scale, out_shift=v0.shift, out_relu=v0.relu,
                 out2=extra[0] if len(extra) > 0 else None, out3=extra[1] if len(extra) > 1 else None, one_hot=one_hot, all_ones=all_ones, scatter=scatter)
    return outs


# ------------------------------------------------------------------------------------------- pre-activated form
class _Res:
    def __init__(self, block, dtype):
        cb = block.conv_branch
        self.bn0 = _bn_affine(cb[0])                                   # applied by the PRODUCER of this block's input
        self.w1 = ops.pack_weight(cb[2].weight, dtype)
        self.bn3 = _bn_affine(cb[3])
        self.w2 = ops.pack_weight(cb[5].weight, dtype)
        ib = block.i_branch[0]
        self.w1x1 = None if isinstance(ib, nn.Identity) else ops.pack_weight(ib.weight, dtype)
        # 2C -> C conv of a decoder block on block-local rows (level 1, C = 32): the staged-unit kernel contracts 32 input channels, so the
        # conv runs as its two input-channel halves -- the second launch takes the first one's result as its residual
        w = cb[2].weight
        self.w1_halves = None
        if w.shape[-1] == 64 and w.shape[0] == 32 and (dtype != torch.float32 or ops.PACK_X3):
            self.w1_halves = (ops.pack_weight(w[..., :32], dtype), ops.pack_weight(w[..., 32:], dtype))

    def run(self, x_raw, x_act, nbr, n, views):
        if isinstance(nbr, BlockedRulebook) and x_act.shape[1] == 64 and self.w1_halves is not None:
            part = ops.conv_fwd(x_act[:, :32], self.w1_halves[0], nbr, n, split=(0, 64))
            t = ops.conv_fwd(x_act[:, 32:], self.w1_halves[1], nbr, n, residual=part, out_scale=self.bn3[0], out_shift=self.bn3[1], out_relu=True,
                             split=(1, 64))
        else:
            t = ops.conv_fwd(x_act, self.w1, nbr, n, out_scale=self.bn3[0], out_shift=self.bn3[1], out_relu=True)
        res = x_raw if self.w1x1 is None else ops.conv_fwd(x_raw, self.w1x1, None, n)
        return _conv_views(t, self.w2, nbr, n, views, residual=res)


class _U:
    def __init__(self, ub, dtype):
        assert len(ub.blocks) == 2 and (len(ub.nPlanes) == 1 or len(ub.blocks_tail) == 2), \
            "the fused engine walks the reference's block_reps = 2 layout (tree_learn.py:41)"
        self.C = ub.nPlanes[0]
        self.blocks = [_Res(b, dtype) for b in ub.blocks._modules.values()]
        self.deeper = len(ub.nPlanes) > 1
        if self.deeper:
            C = self.C
            self.bn_down = _bn_affine(ub.conv[0]); self.wd = ops.pack_weight(ub.conv[2].weight, dtype)
            self.u = _U(ub.u, dtype)
            self.bn_up = _bn_affine(ub.deconv[0]); self.wu = ops.pack_weight(ub.deconv[2].weight, dtype)
            self.tail = [_Res(b, dtype) for b in ub.blocks_tail._modules.values()]
            s, h = self.tail[0].bn0                                    # BN over the 2C concat: split per half
            self.bn_cat_l = (s[:C].contiguous(), h[:C].contiguous())
            self.bn_cat_r = (s[C:].contiguous(), h[C:].contiguous())

    def run_l1_staged(self, x_raw, geom: TileGeometry, views):
        """Level 1 on block-local rows with the BatchNorm + ReLU in front of every block's first conv applied AT STAGING (tl_conv_fwd's
        gather-side prologue, served by the staged-unit kernel: once per staged row): the producers write the raw tensor only -- no
        activated second views, no activated copy of the skip concat (5 x 118 MB of writes per 40 m tile less).  Same dataflow as `run`
        otherwise (reference blocks.py:55-79,137-149)."""
        lv = geom.levels[0]
        n, C, nbr = lv.n, self.C, lv.nbr

        def block(b, x, in_aff, out_views):
            t = ops.conv_fwd(x, b.w1, nbr, n, in_scale=in_aff[0], in_shift=in_aff[1], in_relu=True, out_scale=b.bn3[0], out_shift=b.bn3[1], out_relu=True)
            res = x if b.w1x1 is None else ops.conv_fwd(x, b.w1x1, None, n)
            return _conv_views(t, b.w2, nbr, n, out_views, residual=res)

        (x1,) = block(self.blocks[0], x_raw, self.blocks[0].bn0, [RAW()])
        cat_raw = torch.empty((n, 2 * C), dtype=x_raw.dtype, device=x_raw.device)
        _, xd = block(self.blocks[1], x1, self.blocks[1].bn0, [RAW(cat_raw[:, :C]), ACT(self.bn_down)])
        nxt = geom.levels[1]
        d_raw, d_act = _conv_views(xd, self.wd, lv.child, nxt.n, [RAW(), ACT(self.u.blocks[0].bn0)])
        (e_act,) = self.u.run(d_raw, d_act, geom, 1, [ACT(self.bn_up)])
        _conv_views(e_act, self.wu, lv.inv, n, [RAW(cat_raw[:, C:])], one_hot=True, scatter=lv.child)
        b = self.tail[0]                                               # 2C -> C: the two input-channel halves, each with its slice of the BatchNorm
        part = ops.conv_fwd(cat_raw[:, :C], b.w1_halves[0], nbr, n, in_scale=self.bn_cat_l[0], in_shift=self.bn_cat_l[1], in_relu=True, split=(0, 2 * C))
        t = ops.conv_fwd(cat_raw[:, C:], b.w1_halves[1], nbr, n, in_scale=self.bn_cat_r[0], in_shift=self.bn_cat_r[1], in_relu=True, residual=part,
                         out_scale=b.bn3[0], out_shift=b.bn3[1], out_relu=True, split=(1, 2 * C))
        res = ops.conv_fwd(cat_raw, b.w1x1, None, n)
        (y,) = _conv_views(t, b.w2, nbr, n, [RAW()], residual=res)
        return block(self.tail[1], y, self.tail[1].bn0, views)

    def run(self, x_raw, x_act, geom: TileGeometry, li, views):
        """`views`: what the caller needs of this UBlock's output."""
        lv = geom.levels[li]
        n, C = lv.n, self.C
        nb = len(self.blocks)
        for i, b in enumerate(self.blocks[:-1]):
            x_raw, x_act = b.run(x_raw, x_act, lv.nbr, n, [RAW(), ACT(self.blocks[i + 1].bn0)])
        if not self.deeper:
            return self.blocks[-1].run(x_raw, x_act, lv.nbr, n, views)
        cat_raw = torch.empty((n, 2 * C), dtype=x_raw.dtype, device=x_raw.device)
        cat_act = torch.empty((n, 2 * C), dtype=x_raw.dtype, device=x_raw.device)
        _, _, xd = self.blocks[-1].run(x_raw, x_act, lv.nbr, n,
                                       [RAW(cat_raw[:, :C]), ACT(self.bn_cat_l, cat_act[:, :C]), ACT(self.bn_down)])
        nxt = geom.levels[li + 1]
        d_raw, d_act = _conv_views(xd, self.wd, lv.child, nxt.n, [RAW(), ACT(self.u.blocks[0].bn0)])
        (e_act,) = self.u.run(d_raw, d_act, geom, li + 1, [ACT(self.bn_up)])
        _conv_views(e_act, self.wu, lv.inv, n, [RAW(cat_raw[:, C:]), ACT(self.bn_cat_r, cat_act[:, C:])], one_hot=True, scatter=lv.child)
        y_raw, y_act = self.tail[0].run(cat_raw, cat_act, lv.nbr, n, [RAW(), ACT(self.tail[1].bn0)])
        return self.tail[1].run(y_raw, y_act, lv.nbr, n, views)


# ------------------------------------------------------------------------------------------- prologue form (A/B)
class _ResPro:
    def __init__(self, block, dtype):
        cb = block.conv_branch
        self.s0, self.h0 = _bn_affine(cb[0])
        self.w1 = ops.pack_weight(cb[2].weight, dtype)
        self.s3, self.h3 = _bn_affine(cb[3])
        self.w2 = ops.pack_weight(cb[5].weight, dtype)
        ib = block.i_branch[0]
        self.w1x1 = None if isinstance(ib, nn.Identity) else ops.pack_weight(ib.weight, dtype)

    def run(self, x, nbr, n, out=None):
        t = ops.conv_fwd(x, self.w1, nbr, n, in_scale=self.s0, in_shift=self.h0, in_relu=True,
                         out_scale=self.s3, out_shift=self.h3, out_relu=True)
        res = x if self.w1x1 is None else ops.conv_fwd(x, self.w1x1, None, n)
        return ops.conv_fwd(t, self.w2, nbr, n, out=out, residual=res)


class _UPro:
    def __init__(self, ub, dtype):
        self.C = ub.nPlanes[0]
        self.blocks = [_ResPro(b, dtype) for b in ub.blocks._modules.values()]
        self.deeper = len(ub.nPlanes) > 1
        if self.deeper:
            self.sd, self.hd = _bn_affine(ub.conv[0]); self.wd = ops.pack_weight(ub.conv[2].weight, dtype)
            self.u = _UPro(ub.u, dtype)
            self.su, self.hu = _bn_affine(ub.deconv[0]); self.wu = ops.pack_weight(ub.deconv[2].weight, dtype)
            self.tail = [_ResPro(b, dtype) for b in ub.blocks_tail._modules.values()]

    def run(self, x, geom, li):
        lv = geom.levels[li]
        n, C = lv.n, self.C
        if not self.deeper:
            for b in self.blocks:
                x = b.run(x, lv.nbr, n)
            return x
        cat = torch.empty((n, 2 * C), dtype=x.dtype, device=x.device)
        for i, b in enumerate(self.blocks):
            x = b.run(x, lv.nbr, n, out=cat[:, :C] if i == len(self.blocks) - 1 else None)
        nxt = geom.levels[li + 1]
        d = ops.conv_fwd(x, self.wd, lv.child, nxt.n, in_scale=self.sd, in_shift=self.hd, in_relu=True)
        d = self.u.run(d, geom, li + 1)
        ops.conv_fwd(d, self.wu, lv.inv, n, out=cat[:, C:], in_scale=self.su, in_shift=self.hu, in_relu=True)
        x = cat
        for b in self.tail:
            x = b.run(x, lv.nbr, n)
        return x


class InferencePlan:
    """Folded BatchNorms + packed weights of one TreeLearn module, for one compute dtype."""
    def __init__(self, model, dtype, x3=False):
        """`x3` (fp32 only): the "bf16x3" parity-fast mode -- fp32 storage and epilogues, every conv weight also packed in its split-bf16 form
        so that the kernels of the large levels contract hi / lo bf16 parts on the bf16 matrix cores (tl_conv_args.weight_x3)."""
        self.dtype = dtype
        self.device = model.input_conv[0].weight.device            # a plan belongs to one device (net._plan_ok: a replica on another device builds its own)
        self.x3 = bool(x3) and dtype == torch.float32
        self.preact = os.environ.get("TL_ENGINE", "preact") != "prologue"
        prev, ops.PACK_X3 = ops.PACK_X3, self.x3
        try:
            self.w_in = ops.pack_weight(model.input_conv[0].weight, dtype)
            self.unet = (_U if self.preact else _UPro)(model.unet, dtype)
        finally:
            ops.PACK_X3 = prev
        self.so, self.ho = _bn_affine(model.output_layer[0])
        w1, b1 = [], []
        for mlp in (model.semantic_linear, model.offset_linear):
            lin, bn = mlp[0], mlp[1]
            s, h = _bn_affine(bn)
            w1.append(lin.weight.detach().float() * s[:, None])                 # BN folded into Linear
            b1.append(lin.bias.detach().float() * s + h)
        self.w1 = torch.stack(w1).contiguous(); self.b1 = torch.stack(b1).contiguous()
        self.w2 = torch.cat([model.semantic_linear[3].weight.detach().float(), model.offset_linear[3].weight.detach().float()]).contiguous()
        self.b2 = torch.cat([model.semantic_linear[3].bias.detach().float(), model.offset_linear[3].bias.detach().float()]).contiguous()

    def supports_blocked(self):
        """Level 1 may live in the block-local row order (geometry.BlockedRulebook): the pre-activated engine of a 32-channel net in a 16-bit
        dtype or in the parity-fast mode (fp32 rows, split-bf16 contraction: tl_conv_blk_x3.hip; TL_BLK_X3=0 keeps that mode on the gather kernels)."""
        ok_dtype = self.dtype != torch.float32 or (self.x3 and os.environ.get("TL_BLK_X3", "1") != "0")
        return self.preact and ok_dtype and self.unet.C == 32 and os.environ.get("TL_BLK", "1") != "0"

    def run(self, voxel_feats, geom: TileGeometry, want_backbone=True, all_ones=False):
        """`all_ones`: the caller built voxel_feats as ones (use_feats = False, use_coords = False): the input conv then needs no gather."""
        lv = geom.levels[0]
        vf = voxel_feats.to(self.dtype).contiguous()
        if self.preact:
            # TL_NO_ONES_TABLE=1 (A/B knob: run the input conv on the general kernels): honoured where the canonical table exists; a blocked
            # geometry built without it (the default all-ones inference) keeps the presence-mask form -- same result either way
            ones = all_ones and (os.environ.get("TL_NO_ONES_TABLE") != "1" or (geom.blocked and lv.nbr_ref is None))
            # block-local level 1: BatchNorm + ReLU of the blocks' first convs at staging, raw tensors only (TL_BLK_PRO=0: the two-view form)
            staged = (geom.blocked and self.unet.deeper and self.unet.C == 32 and self.unet.tail[0].w1_halves is not None
                      and os.environ.get("TL_BLK_PRO", "1") != "0")
            in_views = [RAW()] if staged else [RAW(), ACT(self.unet.blocks[0].bn0)]
            if geom.blocked and not ones:
                # block-local level 1 with real input features: the input conv runs on the canonical table, its two views are
                # carried into the block-local order (not the default configuration: the reference feeds ones)
                if lv.nbr_ref is None:
                    raise RuntimeError("a blocked geometry needs ref_table=True when the input features are not all ones")
                # (the voxel features were averaged through the blocked v2p map, so they arrive in the new order)
                xs = _conv_views(vf.index_select(0, lv.nbr.o2n.long()), self.w_in, lv.nbr_ref, lv.n, in_views)
                perm = lv.nbr.perm.long()
                xs = [t.index_select(0, perm) for t in xs]
            else:
                xs = _conv_views(vf, self.w_in, lv.nbr, lv.n, in_views, all_ones=ones)
            if staged:
                (x,) = self.unet.run_l1_staged(xs[0], geom, [RAW()])
            else:
                (x,) = self.unet.run(xs[0], xs[1], geom, 0, [RAW()])
        else:
            x = ops.conv_fwd(vf, self.w_in, lv.nbr, lv.n)
            x = self.unet.run(x, geom, 0)
        if x.shape[1] in ops.HEAD_WIDTHS:
            return ops.head_mlp(x, geom.v2p, self.so, self.ho, self.w1, self.b1, self.w2, self.b2, want_backbone)
        # head widths the fused kernel has no instantiation for (the reference accepts any `channels`): output_layer through
        # tl_affine_relu, the v2p gather and both folded MLPs as plain device matmuls
        bb = ops.affine_relu(x, self.so, self.ho, True).float().index_select(0, geom.v2p)
        h = torch.relu(torch.einsum("nc,moc->mno", bb, self.w1) + self.b1[:, None, :])              # [2, N, C]
        logits = h[0] @ self.w2[:2].T + self.b2[:2]
        offsets = h[1] @ self.w2[2:].T + self.b2[2:]
        return (bb if want_backbone else None), logits, offsets
