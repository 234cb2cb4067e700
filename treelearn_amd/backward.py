"""Backward of the sparse convolution (training step, reference tools/training/train.py:40).

dgrad needs no kernel of its own: with the rulebooks held in both directions it is the forward
gather-GEMM over the TRANSPOSED table with transposed weights --
    SubM      : gin[i] = sum_k W[K-1-k]^T gout[nbr[k][i]]        (nbr[k][o] = i  <=>  nbr[K-1-k][i] = o)
    down k2s2 : gin[i] = sum_k W[k]^T gout[inv[k][i]]
    inverse   : gin[q] = sum_k W[k]^T gout[child[k][q]]
    1x1       : gin    = gout . W
all through `tl_conv_fwd`.  wgrad (gW[k] = sum_o gout[o] (x) x[table[k][o]]) is `tl_conv_wgrad` (csrc/tl_wgrad.hip): fp32
32x32x2 MFMAs over the present (output row, input row) pairs only, two pairs per instruction, no materialised gather; with bf16
inputs (mixed-precision training) the bf16 32x32x16 MFMA takes 16 pairs per instruction from LDS-staged rows.
"""
import os

import torch

from . import ops
from .geometry import BlockedRulebook


class TableRef:
    """A rulebook plus what dgrad needs: the table of the transposed conv and whether taps flip."""
    __slots__ = ("table", "n_out", "t_table", "n_in", "flip", "one_hot", "t_one_hot")

    def __init__(self, table, n_out, t_table, n_in, flip, one_hot=False, t_one_hot=False):
        self.table, self.n_out, self.t_table, self.n_in, self.flip = table, n_out, t_table, n_in, flip
        # exactly one valid entry per row (the parent table of an inverse conv = the transposed table of a strided conv): the kernels
        # gather that row once instead of issuing K gathers of which K - 1 are absent
        self.one_hot, self.t_one_hot = one_hot, t_one_hot


_side = {}


def _side_stream(device):
    st = _side.get(device)
    if st is None:
        st = _side[device] = torch.cuda.Stream(device=device)
    return st


_join_queued = {}
_keepalive = {}


def _may_defer_join(weight):
    """The deferred join is only safe when NOTHING on the main stream can read the weight gradient before backward() returns.  That rules
    out: a non-leaf weight or an existing `.grad` (AccumulateGrad / further nodes consume it at once), tensor hooks and
    post-accumulate-grad hooks on the parameter, double backward (create_graph=True: AccumulateGrad clones instead of stealing), anomaly
    mode (inspects every gradient), and gradient reducers that hook the AccumulateGrad node itself (DDP / FSDP) -- those hooks cannot be
    listed from Python, so an initialised multi-rank process group counts as "a reducer may be listening"."""
    if os.environ.get("TL_WGRAD_JOIN") == "layer" or not weight.is_leaf or weight.grad is not None:
        return False
    if getattr(weight, "_backward_hooks", None) or getattr(weight, "_post_accumulate_grad_hooks", None):
        return False
    if torch.is_grad_enabled() or torch.is_anomaly_enabled():
        return False
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1 \
            and os.environ.get("TL_WGRAD_JOIN") != "deferred":
        return False
    return True


def _join_side(cur, side, weight, held=()):
    """Make the main stream see the side stream's weight gradients.  Inside `loss.backward()` with the parameter's `.grad` still unset
    (the usual `zero_grad()` -> backward -> step loop, reference tools/training/train.py:30-44) nothing on the main stream reads a weight
    gradient before backward returns, so the join is deferred to ONE wait at the end of the backward pass (an autograd final callback):
    the weight gradients of all layers queue up on the side stream and fill whatever the main chain leaves idle, instead of the main
    stream idling 0.3-0.55 ms per layer until that layer's weight gradient is done.  Everything `_may_defer_join` lists joins at once.
    `held`: the tensors the side stream reads; references are kept until the join so that neither the caching allocator hands their
    memory to a main-stream allocation nor the autograd engine, finding itself the sole owner of an incoming gradient, accumulates
    another contribution INTO it while the weight-gradient kernel is still reading.  State is kept per (streams, graph task): a nested
    (re-entrant) backward has its own join and its own held tensors."""
    if _may_defer_join(weight):
        gid = torch._C._current_graph_task_id() if hasattr(torch._C, "_current_graph_task_id") else -1
        key = (cur.device.index, cur.cuda_stream, side.cuda_stream, gid)
        if gid >= 0:
            if key in _join_queued:
                _keepalive[key].extend(held)
                return True
            def _cb(cur=cur, side=side, key=key):
                _join_queued.pop(key, None)
                cur.wait_stream(side)
                torch.cuda.current_stream(cur.device).wait_stream(side)
                _keepalive.pop(key, None)
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_cb)
                _join_queued[key] = True
                _keepalive[key] = list(held)
                return True
            except RuntimeError:
                pass
    cur.wait_stream(side)
    return False


def _wgrad_param(x, grad_out, ref, K, weight):
    """Weight gradient in the parameter's own layout and dtype ([Cout, k, k, k, Cin]; tl_conv_wgrad_ref: no transposing copy)."""
    gw = ops.conv_wgrad(x, grad_out, ref.table, ref.n_out, K, ref_layout=True)
    if not gw._tl_ref_layout:                                               # Cin % 4 != 0: [K, Cout, Cin] came back
        gw = gw.permute(1, 0, 2)
    gw = gw.reshape(weight.shape)
    return gw if gw.dtype == weight.dtype else gw.to(weight.dtype)


def _dgrad_weight(weight, ref, dtype):
    """[K]["Cout" = Cin]["Cin" = Cout] weights of the input-gradient conv: W[k]^T, taps flipped for SubM."""
    from .autograd import _packed_dgrad, note_backward
    note_backward()                       # when this backward pass ends, the packed copies of the weights expire (autograd._pack_epoch)
    return _packed_dgrad(weight, dtype, ref.flip)


def bn_conv_backward(x, a, st, relu, weight, ref: TableRef, grad_out, need_gw, gskip):
    """Backward of y = conv(a), a = relu?(bn_train(x)) (autograd._BNReLUConvFn): returns (dx, dgamma, dbeta, gw).  The weight gradient
    runs on the side stream next to the input-gradient conv, whose epilogue applies the ReLU mask and sums g and g * xhat
    (ops.conv_fwd(epi=("bn_bwd", ...))) so that ONE more pass over (x, g) yields dx; shapes without that epilogue (small levels, > 224
    transposed output channels, the heads' 2 / 3-wide Linears) take the plain conv + tl_bn_train_bwd."""
    from .autograd import FUSE_BN
    co, ci = weight.shape[0], weight.shape[-1]
    K = weight.numel() // (co * ci)
    gw = None
    overlap = need_gw and grad_out.is_cuda and os.environ.get("TL_WGRAD_STREAM", "1") != "0"
    if overlap:
        cur = torch.cuda.current_stream(grad_out.device)
        side = _side_stream(grad_out.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if a is None:                                  # the forward applied the BatchNorm at staging and kept no activated tensor: one apply pass here
                a = ops.affine_relu(x, st[2], st[3], relu)
            gw = _wgrad_param(a, grad_out, ref, K, weight)
        gw.record_stream(cur)
    wt = _dgrad_weight(weight, ref, grad_out.dtype)
    # the transposed conv is computed in column slices when it has > 224 output channels (the MFMA kernels' limit) and for the
    # 2C -> C convs of the decoder from 128 channels on: two C -> C launches run faster than one C -> 2C (level 2: 2 x 0.6 vs 2.4 ms)
    step = ci
    if ci > 224:
        step = 128 if ci % 128 == 0 else (96 if ci % 96 == 0 else 32)
    elif K == 27 and ci == 2 * co and (ci >= 128 or isinstance(ref.t_table, BlockedRulebook)):
        step = co                              # (block-local level 1: the staged-unit kernel computes 32 output channels per launch)
    slices = [(s0, step) for s0 in range(0, ci, step)]
    wts = [wt] if len(slices) == 1 else [wt[:, s0:s0 + w_].contiguous() for s0, w_ in slices]
    res = None
    if FUSE_BN and x.dtype == grad_out.dtype:
        g = torch.empty((ref.n_in, ci), dtype=grad_out.dtype, device=grad_out.device)
        dx = torch.empty_like(x)
        dgb = torch.empty((2, ci), dtype=torch.float32, device=x.device)
        ok = True
        for (s0, w_), wsl in zip(slices, wts):
            sl = slice(s0, s0 + w_)
            r = ops.conv_fwd(grad_out, wsl, ref.t_table, ref.n_in, out=g[:, sl], one_hot=ref.t_one_hot, epi=("bn_bwd", x[:, sl], st[:, sl], relu))
            if r is None or ops.bn_train_bwd_from_parts(x[:, sl], g[:, sl], st[:, sl], r[1], r[2], dx_add=gskip[:, sl] if gskip is not None else None,
                                                        dx=dx[:, sl], dgb=dgb[:, sl]) is None:
                ok = False                                                     # no such epilogue for this shape (nothing launched): plain path
                break
        if ok:
            res = (dx, dgb[0], dgb[1])
    if res is None:
        ga = torch.empty((ref.n_in, ci), dtype=grad_out.dtype, device=grad_out.device)
        for (s0, w_), wsl in zip(slices, wts):
            ops.conv_fwd(grad_out, wsl, ref.t_table, ref.n_in, out=ga[:, s0:s0 + w_], one_hot=ref.t_one_hot)
        res = ops.bn_train_bwd(x, ga, st, relu, dx_add=gskip)
    if need_gw and not overlap:
        if a is None:
            a = ops.affine_relu(x, st[2], st[3], relu)
        gw = _wgrad_param(a, grad_out, ref, K, weight)
    if overlap:
        _join_side(cur, side, weight, (a, grad_out, x))
    return res[0], res[1], res[2], gw


def conv_backward(x, weight, ref: TableRef, grad_out, need_gx, need_gw):
    co, ci = weight.shape[0], weight.shape[-1]
    K = weight.numel() // (co * ci)
    gx = gw = None
    # dgrad and wgrad of a layer depend only on grad_out: with both wanted, the weight gradient goes to a side stream and the two run
    # side by side (the wgrad kernels are latency-chain-bound and leave the matrix pipes mostly idle, the dgrad kernels are
    # MFMA / power-bound -- the same complementarity the tile loop uses across tiles).  The main stream waits for the side stream
    # before this function returns, so everything downstream stays stream-ordered.
    overlap = need_gx and need_gw and grad_out.is_cuda and os.environ.get("TL_WGRAD_STREAM", "1") != "0"
    if overlap:
        cur = torch.cuda.current_stream(grad_out.device)
        side = _side_stream(grad_out.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gw = _wgrad_param(x, grad_out, ref, K, weight)
        gw.record_stream(cur)
    if need_gx:
        wt = _dgrad_weight(weight, ref, grad_out.dtype)                    # kernel layout [K]["Cout"=Cin]["Cin"=Cout] = W[k]^T (taps flipped for SubM)
        if ci <= 224:
            gx = ops.conv_fwd(grad_out, wt, ref.t_table, ref.n_in, one_hot=ref.t_one_hot)
        else:
            # the 2C -> C convs of the decoder have up to 448 "output" channels when transposed; the MFMA kernels cover
            # <= 224, so run column slices of the transposed weights into column views of the result
            gx = torch.empty((ref.n_in, ci), dtype=grad_out.dtype, device=grad_out.device)
            step = 128 if ci % 128 == 0 else (96 if ci % 96 == 0 else 32)
            for s in range(0, ci, step):
                ops.conv_fwd(grad_out, wt[:, s:s + step].contiguous(), ref.t_table, ref.n_in, out=gx[:, s:s + step], one_hot=ref.t_one_hot)
    if need_gw and not overlap:
        gw = _wgrad_param(x, grad_out, ref, K, weight)
    if overlap:
        _join_side(cur, side, weight, (x, grad_out))
    return gx, gw
