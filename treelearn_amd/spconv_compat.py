"""The slice of the `spconv.pytorch` API that the reference model consumes, on the HIP library.

Lower surface of the drop-in boundary: `SparseConvTensor`, `SparseModule`,
`SparseSequential`, `SubMConv3d`, `SparseConv3d`, `SparseInverseConv3d` -- same constructor
arguments, attribute names and `.weight` layout [Cout, k, k, k, Cin] as spconv v2, so that
reference tree_learn/model/blocks.py and tree_learn.py:37-42 build an identical module tree
(identical state-dict keys) on top of it.

A SparseConvTensor here always carries a `TileGeometry` (treelearn_amd.geometry): the rulebooks of
every `indice_key` are built once per batch by the HIP voxel/rulebook kernels, which is what spconv's
`indice_dict` cache amounts to.
"""
import math
from collections import OrderedDict

import torch
from torch import nn

from . import ops
from .backward import TableRef


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, geometry=None, level=0):
        self.features = features
        self.indices = indices
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = None
        self.geometry = geometry
        self.level = level

    def replace_feature(self, feature):
        t = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.geometry, self.level)
        t.indice_dict = self.indice_dict
        t.grid = self.grid
        return t


class SparseModule(nn.Module):
    """marker base class (spconv.pytorch.modules.SparseModule)"""


class SparseSequential(SparseModule):
    def __init__(self, *args):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def forward(self, x, stop=None, want_skip=False, out=None):
        """`stop`: run only the first `stop` modules (the caller runs the rest itself, e.g. the last conv with its residual fused).
        `want_skip`: returns (result, skip) where skip is the input's feature matrix handed back by the FIRST module when that is a
        BatchNorm served by the HIP training kernels (autograd.bn_relu_train(skip=True): the caller's identity / skip path must use
        it instead of the input so that the two gradients of the fan-out are added inside the BatchNorm backward kernel), else None."""
        from .autograd import bn_relu_train, fusable_bn      # late import (autograd depends on ops)
        # `out` (training, optional): a [n, C] column view the LAST module's conv should write its result into (a half of the skip-concat
        # buffer of UBlock.forward); modules that cannot do so ignore it and the caller's concat copies
        mods = list(self._modules.values())[:stop]
        i = 0
        skip = None
        while i < len(mods):
            module = mods[i]
            if isinstance(module, SparseModule):
                x = module(x, out=out) if (out is not None and i == len(mods) - 1 and getattr(module, "accepts_out", False)) else module(x)
            elif isinstance(x, SparseConvTensor):
                if x.features.shape[0] != 0:
                    if fusable_bn(module, x.features):
                        # training-mode BatchNorm1d (+ the ReLU behind it) on the HIP kernels instead of ATen's
                        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                        nxt = mods[i + 1 + int(relu)] if i + 1 + int(relu) < len(mods) else None
                        if type(nxt) in (SubMConv3d, SparseConv3d, SparseInverseConv3d) and nxt.bias is None and torch.is_grad_enabled():
                            # BatchNorm -> ReLU -> conv as one autograd node (autograd._BNReLUConvFn): statistics and backward
                            # reductions ride on the conv kernels' epilogues
                            ws = want_skip and i == 0 and x.features.requires_grad
                            last = i + 2 + int(relu) >= len(mods)
                            x, sk = nxt.forward_fused(x, module, relu, want_skip=ws, out=out if last else None)
                            if ws:
                                skip = sk
                            i += 2 + int(relu)
                            continue
                        if want_skip and i == 0 and x.features.requires_grad:
                            y, skip = bn_relu_train(x.features, module, relu, skip=True)
                            x = x.replace_feature(y)
                        else:
                            x = x.replace_feature(bn_relu_train(x.features, module, relu))
                        i += int(relu)
                    else:
                        x = x.replace_feature(module(x.features))
            else:
                x = module(x)
            i += 1
        return (x, skip) if want_skip else x


class SparseConvolution(SparseModule):
    # Mixed-precision training (the reference trains under torch.cuda.amp.autocast, tools/training/train.py:35-40: spconv
    # convs in half precision with fp32 accumulation; autocast keeps the activations between layers in half precision and
    # BatchNorm's statistics in fp32): when set, features are cast to this dtype for the conv (forward and dgrad run on the bf16
    # MFMA kernels) and the result STAYS in it -- the HIP BatchNorm kernels read and write bf16 with fp64/fp32 statistics, so
    # no cast pass sits between layers.  Set by TreeLearn.forward.
    amp_dtype = None

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, subm=False, inverse=False):
        super().__init__()
        assert dilation == 1 and groups == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.indice_key, self.subm, self.inverse = indice_key, subm, inverse
        k = int(kernel_size)
        self.weight = nn.Parameter(torch.empty(out_channels, k, k, k, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # spconv default: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)), fan_in = Cin * k^3
        fan_in = self.in_channels * int(self.kernel_size) ** 3
        bound = 1.0 / math.sqrt(fan_in)
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    def _table(self, x):
        raise NotImplementedError

    def forward_fused(self, x, bn, relu, residual=None, want_skip=False, out=None):
        """conv(relu?(bn(x))) [+ residual] of a training-mode BatchNorm1d `bn` in front of this conv, as one autograd node
        (autograd.bn_relu_conv).  Returns (SparseConvTensor, skip features or None)."""
        from .autograd import bn_relu_conv, fusable_bn
        ref, out_level = self._table(x)
        amp = SparseConvolution.amp_dtype
        fin = x.features if (amp is None or x.features.dtype == amp) else x.features.to(amp)
        if not fusable_bn(bn, fin) or self.bias is not None:           # not a BatchNorm the HIP training kernels serve: module by module
            f = bn(x.features)
            out = self.forward(x.replace_feature(torch.relu(f) if relu else f), residual)
            return out, (x.features if want_skip else None)
        fuse = residual is not None and residual.dtype == fin.dtype and residual.is_cuda
        res_ = bn_relu_conv(fin, bn, relu, self.weight, ref, residual if fuse else None, want_skip, out=out if (residual is None or fuse) else None)
        feats, skip = res_ if want_skip else (res_, None)
        if residual is not None and not fuse:
            feats = feats + residual.to(feats.dtype)
        lv = x.geometry.levels[out_level]
        res = SparseConvTensor(feats, lv.row_coords(), list(lv.shape), x.batch_size, x.geometry, out_level)
        res.indice_dict = x.indice_dict
        res.grid = x.grid
        return res, skip

    def forward(self, x, residual=None):
        """`residual` [n_out, Cout]: added to the conv result -- inside the kernel's epilogue when the dtypes allow it (the
        `output.features + i_branch(identity).features` of reference blocks.py:76-78 without a separate add pass)."""
        from .autograd import sparse_conv                  # late import (autograd depends on ops)
        ref, out_level = self._table(x)
        amp = SparseConvolution.amp_dtype
        fin = x.features if (amp is None or x.features.dtype == amp) else x.features.to(amp)
        fuse = residual is not None and residual.dtype == fin.dtype and residual.is_cuda
        feats = sparse_conv(fin, self.weight, ref, residual if fuse else None, want_stats=self.bias is None and (residual is None or fuse))
        if residual is not None and not fuse:
            feats = feats + residual.to(feats.dtype)
        if self.bias is not None:
            feats = feats + self.bias
        lv = x.geometry.levels[out_level]
        out = SparseConvTensor(feats, lv.row_coords(), list(lv.shape), x.batch_size, x.geometry, out_level)
        out.indice_dict = x.indice_dict
        out.grid = x.grid
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, indice_key, subm=True)

    def _table(self, x):
        lv = x.geometry.levels[x.level]
        if int(self.kernel_size) == 1:
            return TableRef(None, lv.n, None, lv.n, False), x.level
        assert int(self.kernel_size) == 3, "rulebooks are built for kernel_size 3 (reference configs/_modular/model.yaml:2)"
        return TableRef(lv.nbr, lv.n, lv.nbr, lv.n, True), x.level


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, indice_key)

    def _table(self, x):
        lv = x.geometry.levels[x.level]
        if int(self.kernel_size) == 1 and int(self.stride) == 1:
            return TableRef(None, lv.n, None, lv.n, False), x.level
        assert int(self.kernel_size) == 2 and int(self.stride) == 2
        if lv.child is None:
            raise ValueError("geometry was built with too few levels for this SparseConv3d")
        return TableRef(lv.child, x.geometry.levels[x.level + 1].n, lv.inv, lv.n, False, t_one_hot=True), x.level + 1


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, **kw):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key, inverse=True)

    def _table(self, x):
        assert int(self.kernel_size) == 2 and x.level > 0
        lv = x.geometry.levels[x.level - 1]
        return TableRef(lv.inv, lv.n, lv.child, x.geometry.levels[x.level].n, False, one_hot=True), x.level - 1


class PointToVoxel:
    """`spconv.pytorch.utils.PointToVoxel` as the reference's `voxelize` uses it (tree_learn/model/tree_learn.py:136-143), on the HIP
    voxel kernels: occupancy bitmap + popcount ranks (tl_bitmap_from_points / tl_bitmap_scan / tl_expand_coords / tl_point_rank) and the
    first-<=P-points selection of tl_voxel_mean_feats.  Lets the reference's own `voxelize` run unchanged on this package
    (INTEGRATION.md seam 2); the fused engine does not go through it (geometry.build_geometry builds the same voxels in one pass).

    Semantics: c = floor((p - lo) / vsize) in float32, valid iff 0 <= c < round((hi - lo) / vsize) on every axis; one voxel
    per distinct c; the first <= max_num_points_per_voxel points of a voxel in input order are stored, later ones keep the voxel's id.
    Voxels come in ascending (x, y, z) order (spconv: hash order -- unspecified); with more than max_num_voxels voxels the first
    max_num_voxels of that order are kept and the points of the others get id -1, as do out-of-range points."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel, device=None):
        self.vsize = [float(v) for v in vsize_xyz]
        self.lo = [float(v) for v in coors_range_xyz[:3]]
        self.hi = [float(v) for v in coors_range_xyz[3:]]
        self.C, self.max_voxels, self.P = int(num_point_features), int(max_num_voxels), int(max_num_points_per_voxel)
        self.device = torch.device(device) if device is not None else None
        import numpy as np
        lo32, hi32, vs32 = (np.asarray(a, np.float32) for a in (self.lo, self.hi, self.vsize))
        self.grid = [int(g) for g in np.round((hi32.astype(np.float64) - lo32) / vs32)]

    def generate_voxel_with_id(self, pc):
        """pc f32[n, C] (x, y, z, features...) -> (voxels f32[M, P, C], indices i32[M, 3] in ZYX order, num_per_voxel i32[M],
        pc_voxel_id i64[n])."""
        from . import _hip
        L = _hip.lib()
        _hip.require_cuda(pc.contiguous(), "pc")
        pc = pc.contiguous().float()
        n, C = pc.shape
        dev = pc.device
        if C != self.C:
            raise ValueError(f"PointToVoxel was built for {self.C} point features, got {C}")
        lo = torch.tensor(self.lo, dtype=torch.float32, device=dev); vs = torch.tensor(self.vsize, dtype=torch.float32, device=dev)
        c = torch.floor((pc[:, :3] - lo) / vs)
        grid = torch.tensor(self.grid, dtype=torch.float32, device=dev)
        valid = ((c >= 0) & (c < grid)).all(1)
        ids = torch.full((n,), -1, dtype=torch.int64, device=dev)
        vidx = torch.nonzero(valid).squeeze(1)
        nv = int(vidx.numel())
        if nv == 0:
            z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)      # noqa: E731
            return z(0, self.P, C), z(0, 3, dt=torch.int32), z(0, dt=torch.int32), ids
        pcoords = torch.zeros((nv, 4), dtype=torch.int32, device=dev)
        pcoords[:, 1:] = c[vidx].int()
        ext = [int(v) + 1 for v in pcoords[:, 1:].max(0).values.tolist()]
        dims = (1, ext[0], ext[1], ext[2])
        nw = dims[1] * dims[2] * ((dims[3] + 63) // 64)
        st = _hip.stream()
        bitmap = torch.empty(nw, dtype=torch.int64, device=dev); prefix = torch.empty(nw, dtype=torch.int32, device=dev)
        total = torch.empty(1, dtype=torch.int32, device=dev)
        ws = torch.empty(int(L.tl_scan_ws_words(nw)), dtype=torch.int32, device=dev)
        _hip.check(L.tl_bitmap_from_points(_hip.ptr(pcoords), nv, _hip.dims4(dims), _hip.ptr(bitmap), st), "tl_bitmap_from_points")
        _hip.check(L.tl_bitmap_scan(_hip.ptr(bitmap), nw, _hip.ptr(prefix), _hip.ptr(total), _hip.ptr(ws), st), "tl_bitmap_scan")
        M = int(total.item())
        coords = torch.empty((M, 4), dtype=torch.int32, device=dev)
        _hip.check(L.tl_expand_coords(_hip.ptr(bitmap), _hip.ptr(prefix), _hip.dims4(dims), _hip.ptr(coords), st), "tl_expand_coords")
        v2p = torch.empty(nv, dtype=torch.int64, device=dev)
        _hip.check(L.tl_point_rank(_hip.ptr(pcoords), nv, _hip.ptr(bitmap), _hip.ptr(prefix), _hip.dims4(dims), _hip.ptr(v2p), st), "tl_point_rank")
        if M > self.max_voxels:                                        # capacity: the first max_num_voxels voxels of the canonical order survive
            v2p = torch.where(v2p < self.max_voxels, v2p, torch.full_like(v2p, -1))
            M = self.max_voxels
            coords = coords[:M]
        ids[vidx] = v2p
        # the first <= P points of every voxel in input order: the selection pass of tl_voxel_mean_feats leaves them in its workspace
        sel = torch.empty(M * self.P, dtype=torch.int32, device=dev)
        mean = torch.empty((M, C), dtype=torch.float32, device=dev)
        pv = pc.index_select(0, vidx)
        _hip.check(L.tl_voxel_mean_feats(_hip.ptr(pv), C, _hip.ptr(v2p), nv, M, self.P, _hip.ptr(sel), _hip.ptr(mean), st), "tl_voxel_mean_feats")
        sel = sel.view(M, self.P).long()
        taken = sel != 0x7FFFFFFF
        voxels = pv[sel.clamp(max=nv - 1)] * taken[:, :, None].to(pv.dtype)
        return voxels, coords[:, [3, 2, 1]].contiguous(), taken.sum(1).int(), ids
