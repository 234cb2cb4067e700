"""Building blocks of the sparse U-Net -- same module tree (names, parameter shapes) as reference
tree_learn/model/blocks.py, expressed over treelearn_amd.spconv_compat instead of spconv.

  MLP              blocks.py:8-26      Custom1x1Subm3d  blocks.py:29-39
  ResidualBlock    blocks.py:42-79     UBlock           blocks.py:81-149
The module-by-module `forward`s below are the unfused path (used for training and as a cross-check);
inference goes through treelearn_amd.model.engine, which walks the same modules and issues fused
HIP calls.
"""
from collections import OrderedDict

import torch
from torch import nn

from .. import spconv_compat as spconv
from ..spconv_compat import SparseModule


class MLP(nn.Sequential):
    def __init__(self, in_channels, out_channels, norm_fn=None, num_layers=2):
        modules = []
        for _ in range(num_layers - 1):
            modules.append(nn.Linear(in_channels, in_channels))
            if norm_fn:
                modules.append(norm_fn(in_channels))
            modules.append(nn.ReLU())
        modules.append(nn.Linear(in_channels, out_channels))
        super().__init__(*modules)

    def forward(self, x):
        from ..autograd import bias_add, bn_relu_train, fusable_bn, linear_small_f32, sparse_conv
        from ..backward import TableRef
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if fusable_bn(m, x):                               # training-mode BatchNorm1d + ReLU on the HIP kernels
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = bn_relu_train(x, m, relu)
                i += int(relu)
            elif (isinstance(m, nn.Linear) and x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and m.in_features % 32 == 0
                  and (m.out_features % 32 == 0 or m.out_features <= 8) and x.dtype in (torch.float32, torch.bfloat16, torch.float16)
                  and x.shape[0] * max(m.in_features, m.out_features) * x.element_size() < 2 ** 31):       # 32-bit buffer offsets in tl_conv_wgrad; beyond: nn.Linear
                # the Linears of the heads over millions of points = 1x1 "convs": forward, dgrad and wgrad on the HIP conv kernels
                # (the library GEMMs picked for [3.7 M, 32] x [32, 32] and [3.7 M, 32] x [32, 2] ran 25x below their memory bound:
                # 5.6 ms per call)
                n = x.shape[0]
                w5 = m.weight.view(m.out_features, 1, 1, 1, m.in_features)
                y = linear_small_f32(x, w5) if (i == len(mods) - 1 and m.out_features <= 8 and x.dtype != torch.float32) else None
                if y is not None:                              # a head's OUTPUT layer: fp32 result (and fp32 bias add) from 16-bit inputs
                    x = y
                else:
                    x = sparse_conv(x, w5, TableRef(None, n, None, n, False))
                if m.bias is not None:
                    x = bias_add(x, m.bias)
            else:
                x = m(x)
            i += 1
        return x

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)
        nn.init.normal_(self[-1].weight, 0, 0.01)
        nn.init.constant_(self[-1].bias, 0)


class Custom1x1Subm3d(spconv.SparseConv3d):
    """1x1 'conv' = dense GEMM on the feature matrix (blocks.py:29-39)."""
    def forward(self, input):
        from ..autograd import sparse_conv
        from ..backward import TableRef
        n = input.features.shape[0]
        feats = sparse_conv(input.features, self.weight, TableRef(None, n, None, n, False))
        if self.bias is not None:
            feats = feats + self.bias
        return input.replace_feature(feats)


class ResidualBlock(SparseModule):
    def __init__(self, in_channels, out_channels, norm_fn, kernel_size, indice_key=None):
        super().__init__()
        if in_channels == out_channels:
            self.i_branch = spconv.SparseSequential(nn.Identity())
        else:
            self.i_branch = spconv.SparseSequential(Custom1x1Subm3d(in_channels, out_channels, kernel_size=1, bias=False))
        pad = int((kernel_size - 1) / 2)
        self.conv_branch = spconv.SparseSequential(
            norm_fn(in_channels), nn.ReLU(),
            spconv.SubMConv3d(in_channels, out_channels, kernel_size=int(kernel_size), padding=pad, bias=False, indice_key=indice_key),
            norm_fn(out_channels), nn.ReLU(),
            spconv.SubMConv3d(out_channels, out_channels, kernel_size=int(kernel_size), padding=pad, bias=False, indice_key=indice_key))

    accepts_out = True

    def forward(self, input, out=None):
        """`out` (training, optional): the [n, Cout] column view the block's result should be written into (UBlock's skip-concat buffer)."""
        last = len(self.conv_branch) - 1
        cb = self.conv_branch
        from ..autograd import fusable_bn
        if (torch.is_grad_enabled() and input.features.shape[0] > 1 and input.features.requires_grad and fusable_bn(cb[0], input.features)
                and cb[3].training and cb[2].bias is None and cb[5].bias is None):
            # training: both BatchNorm -> ReLU -> conv triples as fused autograd nodes (autograd._BNReLUConvFn); the first hands the
            # input back (`skip`) for the identity branch, the second adds the identity result in its epilogue
            t, skip = cb[2].forward_fused(input, cb[0], True, want_skip=True)
            res = self.i_branch(input.replace_feature(skip)).features
            y, _ = cb[5].forward_fused(t, cb[3], True, residual=res, out=out)
            return y
        # the input feeds the conv branch AND the identity branch: in training the first BatchNorm hands it back (`skip`) and the
        # identity branch takes that, so the two gradients of the fan-out are added inside the BatchNorm backward kernel
        branch, skip = self.conv_branch(input, stop=last, want_skip=True)
        identity = input.replace_feature(skip if skip is not None else input.features)
        res = self.i_branch(identity).features
        return self.conv_branch[last](branch, residual=res)      # the add rides in the last conv's epilogue


class UBlock(nn.Module):
    def __init__(self, nPlanes, norm_fn, block_reps, block, kernel_size, indice_key_id=1):
        super().__init__()
        self.nPlanes = nPlanes
        key = 'subm{}'.format(indice_key_id)
        self.blocks = spconv.SparseSequential(OrderedDict(
            ('block{}'.format(i), block(nPlanes[0], nPlanes[0], norm_fn, kernel_size, indice_key=key)) for i in range(block_reps)))
        if len(nPlanes) > 1:
            skey = 'spconv{}'.format(indice_key_id)
            self.conv = spconv.SparseSequential(
                norm_fn(nPlanes[0]), nn.ReLU(),
                spconv.SparseConv3d(nPlanes[0], nPlanes[1], kernel_size=2, stride=2, bias=False, indice_key=skey))
            self.u = UBlock(nPlanes[1:], norm_fn, block_reps, block, kernel_size, indice_key_id=indice_key_id + 1)
            self.deconv = spconv.SparseSequential(
                norm_fn(nPlanes[1]), nn.ReLU(),
                spconv.SparseInverseConv3d(nPlanes[1], nPlanes[0], kernel_size=2, bias=False, indice_key=skey))
            self.blocks_tail = spconv.SparseSequential(OrderedDict(
                ('block{}'.format(i), block(nPlanes[0] * (2 - i), nPlanes[0], norm_fn, kernel_size, indice_key=key))
                for i in range(block_reps)))

    def forward(self, input):
        from ..autograd import CAT_IN_PLACE, cat_views, get_stats, set_stats
        buf = None
        f = input.features
        if len(self.nPlanes) > 1 and CAT_IN_PLACE and f.is_cuda and torch.is_grad_enabled() and f.requires_grad and self.training:
            # training: the skip concat (reference blocks.py:146) without its copy -- the encoder's last block and the inverse conv write
            # straight into the two column halves of one [n, 2C] buffer (what the inference engine does with its views)
            C = self.nPlanes[0]
            dt = spconv.SparseConvolution.amp_dtype or f.dtype
            buf = torch.empty((f.shape[0], 2 * C), dtype=dt, device=f.device)
        output = self.blocks(input, out=buf[:, :self.nPlanes[0]]) if buf is not None else self.blocks(input)
        if len(self.nPlanes) > 1:
            down, skip = self.conv(output, want_skip=True)     # same fan-out as in ResidualBlock: skip = the features the concat takes
            identity = skip if skip is not None else output.features
            dec = self.deconv(self.u(down), out=buf[:, self.nPlanes[0]:]) if buf is not None else self.deconv(self.u(down))
            cat = cat_views(identity, dec.features, buf)
            sa, sb = get_stats(identity), get_stats(dec.features)
            if sa is not None and sb is not None:              # per-channel statistics of a concat = those of its halves
                set_stats(cat, list(sa) + list(sb))
            output = output.replace_feature(cat)
            output = self.blocks_tail(output)
        return output
