"""torch.autograd bridge for the HIP sparse convolution (unfused, module-by-module path).

forward  : tl_conv_fwd
backward : dgrad = tl_conv_fwd over the transposed rulebook with W^T; wgrad = tl_conv_wgrad
           (reference: spconv's autograd functions behind SubMConv3d/SparseConv3d.forward,
           exercised by tools/training/train.py:40 `scaler.scale(loss).backward()`).
"""
import torch

from . import ops

_packed_cache = {}


def _packed(weight, dtype):
    """[Cout,k,k,k,Cin] parameter -> [K,Cout,Cin] kernel layout, cached per (storage, version, dtype)."""
    key = (weight.data_ptr(), weight._version, dtype, tuple(weight.shape))
    hit = _packed_cache.get(id(weight))
    if hit is not None and hit[0] == key:
        return hit[1]
    w = ops.pack_weight(weight, dtype)
    _packed_cache[id(weight)] = (key, w)
    return w


class _SparseConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, weight, ref):
        x = feats.contiguous()
        out = ops.conv_fwd(x, _packed(weight, x.dtype), ref.table, ref.n_out)
        ctx.save_for_backward(x, weight)
        ctx.ref = ref
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import backward as bw
        x, weight = ctx.saved_tensors
        gx, gw = bw.conv_backward(x, weight, ctx.ref, grad_out.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, None


def sparse_conv(feats, weight, ref):
    """`ref`: treelearn_amd.backward.TableRef (rulebook + its transpose)."""
    if torch.is_grad_enabled() and (feats.requires_grad or weight.requires_grad):
        return _SparseConvFn.apply(feats, weight, ref)
    x = feats.contiguous()
    return ops.conv_fwd(x, _packed(weight, x.dtype), ref.table, ref.n_out)
