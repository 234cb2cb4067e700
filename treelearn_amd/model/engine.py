"""Fused inference engine: walks the TreeLearn module tree and issues HIP calls with
BatchNorm(eval)+ReLU folded into conv prologues/epilogues, the residual add folded into the
second conv's epilogue and the skip concat realised as column views of one [N, 2C] buffer.

Per ResidualBlock (reference blocks.py:55-79):
    t   = conv1( relu(bn0(x)) )            prologue bn0+relu ; epilogue bn3+relu  (t is only ever read through bn3)
    out = conv2( t ) + i_branch(x)          epilogue residual add (x itself, or the 1x1 GEMM of x)
Per UBlock (blocks.py:137-149): the last block of `blocks` writes straight into cat[:, :C]; the inverse
conv writes cat[:, C:]; `blocks_tail.block0` reads cat as a [N, 2C] input.
"""
import torch
from torch import nn

from .. import ops
from ..geometry import TileGeometry


def _bn_affine(bn: nn.BatchNorm1d):
    """eval-mode BatchNorm1d as y = x * scale + shift (fp32)."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale.contiguous(), shift.contiguous()


class _ResPlan:
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


class _UPlan:
    def __init__(self, ub, dtype):
        self.C = ub.nPlanes[0]
        self.blocks = [_ResPlan(b, dtype) for b in ub.blocks._modules.values()]
        self.deeper = len(ub.nPlanes) > 1
        if self.deeper:
            self.sd, self.hd = _bn_affine(ub.conv[0]); self.wd = ops.pack_weight(ub.conv[2].weight, dtype)
            self.u = _UPlan(ub.u, dtype)
            self.su, self.hu = _bn_affine(ub.deconv[0]); self.wu = ops.pack_weight(ub.deconv[2].weight, dtype)
            self.tail = [_ResPlan(b, dtype) for b in ub.blocks_tail._modules.values()]

    def run(self, x, geom: TileGeometry, li):
        lv = geom.levels[li]
        n, C = lv.n, self.C
        if not self.deeper:
            for b in self.blocks:
                x = b.run(x, lv.nbr, n)
            return x
        cat = torch.empty((n, 2 * C), dtype=x.dtype, device=x.device)
        for i, b in enumerate(self.blocks):
            last = i == len(self.blocks) - 1
            x = b.run(x, lv.nbr, n, out=cat[:, :C] if last else None)          # identity -> left half
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
    def __init__(self, model, dtype):
        self.dtype = dtype
        self.w_in = ops.pack_weight(model.input_conv[0].weight, dtype)
        self.unet = _UPlan(model.unet, dtype)
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

    def run(self, voxel_feats, geom: TileGeometry, want_backbone=True):
        lv = geom.levels[0]
        x = ops.conv_fwd(voxel_feats.to(self.dtype).contiguous(), self.w_in, lv.nbr, lv.n)
        x = self.unet.run(x, geom, 0)
        return ops.head_mlp(x, geom.v2p, self.so, self.ho, self.w1, self.b1, self.w2, self.b2, want_backbone)
