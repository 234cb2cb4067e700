"""torch.autograd bridge for the HIP sparse convolution (unfused, module-by-module path).

forward  : tl_conv_fwd
backward : dgrad = tl_conv_fwd over the transposed rulebook with W^T; wgrad = tl_conv_wgrad
           (reference: spconv's autograd functions behind SubMConv3d/SparseConv3d.forward,
           exercised by tools/training/train.py:40 `scaler.scale(loss).backward()`).
"""
import math
import os
import weakref

import torch

from . import ops
from .geometry import BlockedRulebook

_packed_cache = {}          # id(parameter) -> (weakref to it, (epoch, version, dtype, device, data_ptr), packed); evicted when the parameter dies

# The kernel-layout copies of a parameter are valid for ONE pack epoch.  `_version` alone is not enough: torch's single-kernel optimizers
# (`fused=True`: torch._fused_adamw_ / _fused_sgd_ / _fused_adam_) update the parameters without bumping it, and a cache keyed on the version
# would keep training on the weights of step 0.  A new epoch begins with every training forward of the model (`TreeLearn.forward_backbone`)
# and at the end of every backward pass that went through one of the conv Functions below (`note_backward`), i.e. before any optimizer can
# have run; within an epoch forward, input-gradient and weight-gradient launches of a layer share one packing.
_pack_epoch = 0
_epoch_queued = set()


def new_pack_epoch():
    global _pack_epoch
    _pack_epoch += 1


def note_backward():
    """Called from the conv Functions' backward: once per backward pass, when it ends, the pack epoch advances."""
    gid = torch._C._current_graph_task_id() if hasattr(torch._C, "_current_graph_task_id") else -1
    if gid in _epoch_queued:
        return
    def _cb(gid=gid):
        _epoch_queued.discard(gid)
        new_pack_epoch()
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_cb)
        _epoch_queued.add(gid)
    except RuntimeError:                       # not inside a backward pass (a Function's backward called by hand)
        pass


def _packed(weight, dtype):
    """[Cout,k,k,k,Cin] parameter -> [K,Cout,Cin] kernel layout, cached per parameter OBJECT (the weak reference guards against a
    recycled id) for the current pack epoch; also invalidated by in-place updates that bump `_version` or a dtype / device change."""
    key = (_pack_epoch, weight._version, dtype, weight.device, weight.data_ptr())
    hit = _packed_cache.get(id(weight))
    if hit is not None and hit[0]() is weight and hit[1] == key:
        return hit[2]
    w = ops.pack_weight(weight, dtype)
    if hit is None or hit[0]() is not weight:
        weakref.finalize(weight, _packed_cache.pop, id(weight), None)
    _packed_cache[id(weight)] = (weakref.ref(weight), key, w)
    return w


FUSE_BN = os.environ.get("TL_TRAIN_FUSE", "1") != "0"      # conv-epilogue BatchNorm reductions in training (0: the separate passes, for A/B)
# test hook: when set to a dict, every training-mode BatchNorm + ReLU served by the HIP kernels stores its ReLU decisions there
# ({BatchNorm module: bool [rows, C] = output > 0}), so that a float64 reference can differentiate the same piecewise-linear function
RELU_MASK_SINK = None
# test hook, the other direction: {BatchNorm module: bool [rows, C]} -- the ReLU behind that BatchNorm takes the GIVEN branches instead of
# deciding them from its own pre-activations (forward: affine without ReLU, times the mask; backward: the plain input-gradient conv, the
# mask, tl_bn_train_bwd without ReLU).  Lets two runs of different precision differentiate the same piecewise-linear function.
RELU_MASK_SOURCE = None


def _forced_mask(bn, relu):
    return RELU_MASK_SOURCE.get(bn) if (RELU_MASK_SOURCE is not None and relu) else None


# block-local level 1 in training: BatchNorm + ReLU applied when the conv stages its rows instead of an apply pass (the activated tensor is then
# recomputed for the weight gradient on its stream).  OPT-IN (TL_BLK_TRAIN_PRO=1): measured at no gain on the step (61.2 vs 61.3 ms, three
# alternations) -- the pass it removes from the main stream re-appears on the weight-gradient stream and the two streams share the device --
# and the 2C -> C conv as two staged halves rounds its half sums once more (gradients at cosine >= 0.9989 of the default's)
STAGE_TRAIN = os.environ.get("TL_BLK_TRAIN_PRO", "0") == "1"
CAT_IN_PLACE = os.environ.get("TL_TRAIN_CAT", "1") != "0"   # skip concat without a copy: both producers write into the halves of one buffer (0: torch.cat, for A/B)


def _rows_ok(x):
    """A feature matrix the kernels take as it is: unit column stride, 16-B aligned rows (a column half of a skip-concat buffer qualifies)."""
    return x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0


def set_stats(t, stats):
    """Attach a conv epilogue's per-channel partial sums to its result `t` for the BatchNorm that consumes it, with the tensor's version
    counter and address: an in-place change of the features in between (an inplace ReLU, `mul_`, a user module in a SparseSequential) makes
    the sums stale, and get_stats() then returns None (the statistics kernel reads the tensor instead)."""
    t._tl_stats = (stats, t._version, t.data_ptr())


def get_stats(t):
    rec = getattr(t, "_tl_stats", None)
    if rec is None:
        return None
    stats, version, addr = rec
    return stats if (t._version == version and t.data_ptr() == addr) else None


def _conv_with_stats(x, w_packed, ref, residual, holder, out=None):
    """Forward conv whose epilogue also sums y and y^2 per channel when the kernel family can (ops.conv_fwd(epi="stats")); the partial
    sums go to holder["stats"] = [(parts, nparts, Cout)] for the BatchNorm that consumes the result (attached to the output tensor as
    `_tl_stats` by the caller)."""
    if holder is not None and FUSE_BN and ref.n_out > 1:
        r = ops.conv_fwd(x, w_packed, ref.table, ref.n_out, out=out, residual=residual, one_hot=ref.one_hot, epi="stats")
        if r is not None:
            holder["stats"] = [(r[1], r[2], int(w_packed.shape[1]))]
            return r[0]
    return ops.conv_fwd(x, w_packed, ref.table, ref.n_out, out=out, residual=residual, one_hot=ref.one_hot)


_dgrad_cache = {}           # id(parameter) -> (weakref, (version, dtype, device, data_ptr, flip), [K, Cin, Cout] weights of the input-gradient conv)


def _packed_dgrad(weight, dtype, flip):
    """W[k]^T (taps flipped for SubM) for the layer's input-gradient conv, cached like `_packed`."""
    key = (_pack_epoch, weight._version, dtype, weight.device, weight.data_ptr(), bool(flip))
    hit = _dgrad_cache.get(id(weight))
    if hit is not None and hit[0]() is weight and hit[1] == key:
        return hit[2]
    w = ops.pack_weight_dgrad(weight, dtype, flip)
    if hit is None or hit[0]() is not weight:
        weakref.finalize(weight, _dgrad_cache.pop, id(weight), None)
    _dgrad_cache[id(weight)] = (weakref.ref(weight), key, w)
    return w


class PackPlan:
    """Every conv weight of a model packed in ONE launch per optimizer step (tl_pack_weights_batch): the kernel layout, its
    fragment-order copy and the input-gradient layout, into persistent buffers; the per-parameter caches above are then hits for the
    whole step.  `convs`: [(weight parameter [Cout,k,k,k,Cin], flip)] with flip = taps flip in the input-gradient conv (SubM)."""

    def __init__(self, convs, dtype):
        import ctypes
        from . import _hip
        self.dtype = dtype
        self.params = [w for w, _ in convs]
        self.sig = None
        dev = self.params[0].device
        self.out = []                                                   # per conv: (packed, frag or None, dgrad)
        descs, blocks = [], []

        def add(w, dst, form, co, K, ci):
            n = co * K * ci
            descs.append((w.data_ptr(), dst.data_ptr(), co, K, ci, form))
            blocks.extend((len(descs) - 1, b) for b in range((n + 4095) // 4096))
        for w, flip in convs:
            co, ci = w.shape[0], w.shape[-1]
            K = w.numel() // (co * ci)
            pk = torch.empty((K, co, ci), dtype=dtype, device=dev)
            add(w, pk, 0, co, K, ci)
            fr = None
            if co % 32 == 0 and ci % 32 == 0 and K > 1:
                fr = torch.empty(K * co * ci, dtype=dtype, device=dev)
                add(w, fr, 1, co, K, ci)
                pk._tl_frag = fr
            dg = torch.empty((K, ci, co), dtype=dtype, device=dev)
            add(w, dg, 3 if flip else 2, co, K, ci)
            self.out.append((pk, fr, dg, bool(flip)))

        class _D(ctypes.Structure):
            _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("Cout", ctypes.c_int32), ("K", ctypes.c_int32), ("Cin", ctypes.c_int32), ("form", ctypes.c_int32)]
        arr = (_D * len(descs))(*[_D(*d) for d in descs])
        self.descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.blocks = torch.tensor(blocks, dtype=torch.int32).to(dev)
        self.ptrs = tuple(w.data_ptr() for w in self.params)

    def valid_for(self, convs, dtype):
        return dtype == self.dtype and len(convs) == len(self.params) and all(a is b for (a, _), b in zip(convs, self.params)) \
            and tuple(w.data_ptr() for w in self.params) == self.ptrs

    def refresh(self):
        from . import _hip
        sig = (_pack_epoch,) + tuple(w._version for w in self.params)
        if sig == self.sig:
            return
        L = _hip.lib()
        _hip.check(L.tl_pack_weights_batch(_hip.ptr(self.descs), _hip.ptr(self.blocks), self.blocks.shape[0], _hip.dtype_code(self.dtype), _hip.stream()),
                   "tl_pack_weights_batch")
        self.sig = sig
        for w, (pk, fr, dg, flip) in zip(self.params, self.out):
            for cache, key, val in ((_packed_cache, (_pack_epoch, w._version, self.dtype, w.device, w.data_ptr()), pk),
                                    (_dgrad_cache, (_pack_epoch, w._version, self.dtype, w.device, w.data_ptr(), flip), dg)):
                hit = cache.get(id(w))
                if hit is None or hit[0]() is not w:
                    weakref.finalize(w, cache.pop, id(w), None)
                cache[id(w)] = (weakref.ref(w), key, val)


class _SparseConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, weight, ref, residual, holder=None):
        x = feats.contiguous()
        out = _conv_with_stats(x, _packed(weight, x.dtype), ref, residual, holder)
        ctx.save_for_backward(x, weight)
        ctx.ref = ref
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import backward as bw
        x, weight = ctx.saved_tensors
        if grad_out.stride(1) != 1 or grad_out.stride(0) % 8 or grad_out.data_ptr() % 16:
            grad_out = grad_out.contiguous()
        gx, gw = bw.conv_backward(x, weight, ctx.ref, grad_out, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, None, (grad_out if ctx.needs_input_grad[3] else None), None      # d(out)/d(residual) = identity: no kernel


def sparse_conv(feats, weight, ref, residual=None, want_stats=False):
    """`ref`: treelearn_amd.backward.TableRef (rulebook + its transpose); `residual` is added in the kernel's epilogue.
    want_stats (training): the result carries `_tl_stats`, its per-channel partial sums from the conv epilogue, when available."""
    if torch.is_grad_enabled() and (feats.requires_grad or weight.requires_grad or (residual is not None and residual.requires_grad)):
        holder = {} if want_stats else None
        out = _SparseConvFn.apply(feats, weight, ref, residual, holder)
        if holder:
            set_stats(out, holder["stats"])
        return out
    x = feats.contiguous()
    return ops.conv_fwd(x, _packed(weight, x.dtype), ref.table, ref.n_out, residual=residual, one_hot=ref.one_hot)


def _bn_forward_stats(x, bn, gamma, beta, stats_in):
    """st [4, C] (mean, rstd, scale, shift) of the training-mode BatchNorm `bn` over x; running statistics updated.  `stats_in`: the
    producer's conv-epilogue partial sums (no pass over x), else the statistics kernel reads x."""
    track = bn.track_running_stats and bn.running_mean is not None
    g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
    rm, rv, nbt = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if track else (None, None, None)
    if stats_in is not None and sum(sg[2] for sg in stats_in) == x.shape[1]:
        return ops.bn_train_finish(stats_in, x.shape[0], g32, b32, bn.eps, bn.momentum, rm, rv, nbt)
    return ops.bn_train_stats(x, g32, b32, bn.eps, bn.momentum, rm, rv, nbt)


_halves_cache = {}


def _packed_halves(weight, dtype):
    """The two input-channel halves [K, Cout, Cin / 2] of a packed conv weight, cached like `_packed`."""
    key = (_pack_epoch, weight._version, dtype, weight.device, weight.data_ptr())
    hit = _halves_cache.get(id(weight))
    if hit is not None and hit[0]() is weight and hit[1] == key:
        return hit[2]
    w = _packed(weight, dtype)
    h = w.shape[2] // 2
    val = (w[:, :, :h].contiguous(), w[:, :, h:].contiguous())
    if hit is None or hit[0]() is not weight:
        weakref.finalize(weight, _halves_cache.pop, id(weight), None)
    _halves_cache[id(weight)] = (weakref.ref(weight), key, val)
    return val


def _staged_bn_conv(x, st, relu, weight, ref, residual, holder, out):
    """conv(relu?(x * scale + shift)) on the staged-unit kernel with the affine applied at staging and the statistics epilogue; None when the
    kernel does not serve the launch (nothing enqueued)."""
    n = ref.n_out
    if weight.shape[-1] == 32:
        r = ops.conv_fwd(x, _packed(weight, x.dtype), ref.table, n, out=out, residual=residual, in_scale=st[2], in_shift=st[3], in_relu=relu, epi="stats")
    else:
        if residual is not None:
            return None
        w0, w1 = _packed_halves(weight, x.dtype)
        part = ops.conv_fwd(x[:, :32], w0, ref.table, n, in_scale=st[2, :32], in_shift=st[3, :32], in_relu=relu, split=(0, 64))
        r = ops.conv_fwd(x[:, 32:], w1, ref.table, n, out=out, residual=part, in_scale=st[2, 32:], in_shift=st[3, 32:], in_relu=relu, epi="stats", split=(1, 64))
    if r is None:
        return None
    holder["stats"] = [(r[1], r[2], 32)]
    return r[0]


class _BNReLUConvFn(torch.autograd.Function):
    """y = conv(relu?(BatchNorm1d_train(x))) [+ residual] as ONE autograd node -- the `norm_fn(C), nn.ReLU(), conv` triple of reference
    blocks.py:55-70,102-123 in training mode -- so that the BatchNorm's reductions ride on the neighbouring conv kernels:
      forward : statistics of x from the producer's epilogue (`stats_in`), one apply pass, the conv -- whose epilogue in turn sums its
                result for the next BatchNorm (holder["stats"]);
      backward: the weight gradient on the side stream; the input-gradient conv masks its result with the ReLU and sums g and g * xhat
                in its epilogue (ops.conv_fwd(epi=("bn_bwd", ...))); ONE pass forms dx (+ the gradient of the identity / skip use).
    Where a kernel family has no such epilogue the separate passes run instead (same results up to summation order)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, weight, residual, bn, relu, ref, want_skip, stats_in, holder, out=None):
        if not _rows_ok(x) or (stats_in is None and not x.is_contiguous()):     # (the statistics kernel reads dense matrices; a column view with
            x = x.contiguous()                                                  #  its producer's partial sums goes through as it is)
        st = _bn_forward_stats(x, bn, gamma, beta, stats_in)
        fm = _forced_mask(bn, relu)
        y = a = None
        if (STAGE_TRAIN and fm is None and RELU_MASK_SINK is None and FUSE_BN and holder is not None and x.dtype in (torch.bfloat16, torch.float16)
                and isinstance(ref.table, BlockedRulebook) and weight.shape[0] == 32 and weight.numel() // (32 * weight.shape[-1]) == 27
                and weight.shape[-1] in (32, 64)):
            # block-local level 1: the staged-unit kernel applies relu(x * scale + shift) to every row it stages (1.8 rows per output row), so the
            # activated tensor is neither written nor read here; the weight gradient recomputes it on its own stream (backward.bn_conv_backward).
            # The 2C -> C conv of the decoder block runs as its two input-channel halves, the second taking the first one's sums as residual.
            y = _staged_bn_conv(x, st, relu, weight, ref, residual, holder, out)
        if y is None:
            if fm is not None:
                a = ops.affine_relu(x, st[2], st[3], False) * fm.to(x.dtype)
            else:
                a = ops.affine_relu(x, st[2], st[3], relu)
            if RELU_MASK_SINK is not None and relu:
                RELU_MASK_SINK[bn] = fm if fm is not None else a > 0
            y = _conv_with_stats(a, _packed(weight, a.dtype), ref, residual, holder, out=out)
        ctx.save_for_backward(x, a, st, weight)
        ctx.ref, ctx.relu, ctx.fm = ref, relu, fm
        if want_skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, gy, gskip=None):
        from . import backward as bw
        x, a, st, weight = ctx.saved_tensors
        # the gradient of a skip concat arrives as column views of one [n, 2C] tensor: the kernels take a row stride, no copies
        if gy.stride(1) != 1 or gy.stride(0) % 8 or gy.data_ptr() % 16:
            gy = gy.contiguous()
        if gskip is not None and (gskip.dtype != x.dtype or gskip.stride(1) != 1 or gskip.stride(0) % 8 or gskip.data_ptr() % 16):
            gskip = gskip.to(x.dtype).contiguous()
        need_gw = ctx.needs_input_grad[3]
        if ctx.fm is not None:                                   # test hook (RELU_MASK_SOURCE): the given ReLU branches
            ga, gw = bw.conv_backward(a, weight, ctx.ref, gy, True, need_gw)
            dx, dgamma, dbeta = ops.bn_train_bwd(x, ga * ctx.fm.to(ga.dtype), st, False, dx_add=gskip)
        else:
            dx, dgamma, dbeta, gw = bw.bn_conv_backward(x, a, st, ctx.relu, weight, ctx.ref, gy, need_gw, gskip)
        return dx, dgamma, dbeta, gw, (gy if ctx.needs_input_grad[4] else None), None, None, None, None, None, None, None


class _CatViewsFn(torch.autograd.Function):
    """cat((a, b), dim=1) when a and b ARE the two column halves of `buf` (their producers wrote them there: reference blocks.py:146 without
    the copy).  Backward hands the halves of the incoming gradient on as column views (the kernels take a row stride)."""

    @staticmethod
    def forward(ctx, a, b, buf):
        ctx.C = a.shape[1]
        return buf.detach()

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.C], g[:, ctx.C:], None


def cat_views(a, b, buf):
    """The skip concat of UBlock.forward: `buf` if a / b alias its halves, else torch.cat."""
    C = a.shape[1]
    if (buf is not None and a.dtype == buf.dtype and b.dtype == buf.dtype and b.shape[1] == buf.shape[1] - C and a.data_ptr() == buf.data_ptr()
            and b.data_ptr() == buf.data_ptr() + C * buf.element_size() and a.stride(0) == buf.stride(0) and b.stride(0) == buf.stride(0)):
        return _CatViewsFn.apply(a, b, buf)
    return torch.cat((a, b), dim=1)


def bn_relu_conv(x, bn, relu, weight, ref, residual=None, want_skip=False, out=None):
    """Fused training-mode BatchNorm1d(+ReLU) -> sparse conv of the feature matrix x; returns y, or (y, skip) with want_skip (skip = x
    passed through: the caller's identity path must use it, see _BNReLUTrainFn).  y carries `_tl_stats` when the conv kernel summed it."""
    holder = {}
    stats_in = get_stats(x) if FUSE_BN else None
    if out is not None and (out.dtype != x.dtype or not _rows_ok(out) or out.shape != (ref.n_out, weight.shape[0])):
        out = None                                              # not a view the conv can write: plain result (the caller's concat then copies)
    out = _BNReLUConvFn.apply(x, bn.weight, bn.bias, weight, residual, bn, relu, ref, want_skip, stats_in, holder, out)
    y = out[0] if want_skip else out
    if "stats" in holder:
        set_stats(y, holder["stats"])
    if want_skip:
        if stats_in is not None:
            set_stats(out[1], stats_in)
        return y, out[1]
    return y


class _BNReLUTrainFn(torch.autograd.Function):
    """y = relu?(BatchNorm1d_train(x)) on the HIP kernels (tl_bn_train_stats + tl_affine_relu forward, tl_bn_train_bwd backward):
    the `norm_fn(C), nn.ReLU()` pairs of reference blocks.py:55-70,102-123 / tree_learn.py:42-46 in training mode.
    With `skip`, x itself comes back as a second output: the block's identity / skip path takes THAT tensor instead of x, so x has one
    consumer in the graph and the gradient of its other use arrives here as `dskip` and is added inside the backward kernel
    (tl_bn_train_bwd's dx_add) -- otherwise autograd accumulates the two gradients of every residual / skip fan-out with a kernel of
    its own (26 per step of the default architecture)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, relu, skip, stats_in=None):
        x = x.contiguous()
        st = _bn_forward_stats(x, bn, gamma, beta, stats_in)
        fm = _forced_mask(bn, relu)
        if fm is not None:
            y = ops.affine_relu(x, st[2], st[3], False) * fm.to(x.dtype)
        else:
            y = ops.affine_relu(x, st[2], st[3], relu)
        if RELU_MASK_SINK is not None and relu:
            RELU_MASK_SINK[bn] = fm if fm is not None else y > 0
        ctx.save_for_backward(x, st)
        ctx.relu, ctx.fm = relu, fm
        if skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, st = ctx.saved_tensors
        if dy.stride(1) != 1 or dy.stride(0) % 8 or dy.data_ptr() % 16:
            dy = dy.contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            dy = dy.float()
        if dskip is not None and (dskip.dtype != x.dtype or dskip.stride(1) != 1 or dskip.stride(0) % 8 or dskip.data_ptr() % 16):
            dskip = dskip.to(x.dtype).contiguous()
        if ctx.fm is not None:                                   # test hook (RELU_MASK_SOURCE)
            dx, dgamma, dbeta = ops.bn_train_bwd(x, dy * ctx.fm.to(dy.dtype), st, False, dx_add=dskip)
        else:
            dx, dgamma, dbeta = ops.bn_train_bwd(x, dy, st, ctx.relu, dx_add=dskip)    # dx in x's dtype (bf16 stays bf16 under mixed precision)
        return dx, dgamma, dbeta, None, None, None, None


def bn_relu_train(x, bn, relu=True, skip=False):
    """Training-mode BatchNorm1d `bn` (+ ReLU) of the feature matrix x on the HIP library; skip=True returns (y, x passed through)."""
    stats_in = get_stats(x) if FUSE_BN else None
    out = _BNReLUTrainFn.apply(x, bn.weight, bn.bias, bn, relu, skip, stats_in)
    if skip and stats_in is not None:
        set_stats(out[1], stats_in)
    return out


class _BiasAddFn(torch.autograd.Function):
    """x + bias over millions of rows (the heads' Linear biases): the bias gradient is a column sum over all rows, done on the
    statistics kernel (ops.column_sum) instead of ATen's dim-0 reduction."""

    @staticmethod
    def forward(ctx, x, bias):
        return x + bias.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        n, C = g.shape
        if g.is_cuda and C % 4 == 0 and C <= 1024 and n > 1 and g.dtype in (torch.float32, torch.bfloat16, torch.float16):
            db = ops.column_sum(g)
        elif g.is_cuda and C < 4 and n >= 64 and g.dtype in (torch.float32, torch.bfloat16, torch.float16):
            # 2 or 3 columns (the heads' output layers): fold rows so that the matrix is a multiple of four wide -- [n, C] read as
            # [n / r, r * C] with r * C = lcm(C, 4) -- sum its columns on the same kernel, then add the r partial rows (and the < r
            # left-over rows); ATen's dim-0 reduction of a [3.7 M, 3] matrix takes 0.6 ms
            r = 4 // math.gcd(C, 4)
            m = n - n % r
            db = ops.column_sum(g[:m].view(m // r, r * C)).view(r, C).sum(0)
            if m < n:
                db = db + g[m:].float().sum(0)
        else:
            db = g.float().sum(0)
        return g, db


def bias_add(x, bias):
    return _BiasAddFn.apply(x, bias)


def fusable_bn(module, x):
    """True when `module` is a BatchNorm1d that the HIP training kernels serve for x (batch statistics, affine, CUDA, C % 4 == 0)."""
    return (isinstance(module, torch.nn.BatchNorm1d) and module.training and module.affine and module.momentum is not None and x.is_cuda
            and x.dim() == 2 and x.shape[1] % 4 == 0 and x.shape[1] <= 1024 and x.shape[0] > 1 and x.dtype in (torch.float32, torch.bfloat16, torch.float16))


class _GatherRowsFn(torch.autograd.Function):
    """point_feats = voxel_feats[v2p] (reference tree_learn.py:98) on tl_gather_rows; backward = tl_scatter_add_rows over the stable argsort
    of v2p (cached on the geometry object: one sort per batch), deterministic."""

    @staticmethod
    def forward(ctx, feats, idx, cache):
        feats = feats.contiguous()
        out = ops.gather_rows(feats, idx)
        ctx.save_for_backward(idx)
        ctx.n_rows, ctx.cache = feats.shape[0], cache
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        if g.stride(1) != 1:
            g = g.contiguous()
        srt = ctx.cache.get("v2p_sort") if ctx.cache is not None else None
        if srt is None:
            sidx, order = torch.sort(idx, stable=True)
            srt = (order, sidx)
            if ctx.cache is not None:
                ctx.cache["v2p_sort"] = srt
        return ops.scatter_add_rows(g, srt[0], srt[1], ctx.n_rows), None, None


def gather_rows(feats, idx, cache=None):
    """feats[idx] for a 2-D feature matrix and an int64 index vector; differentiable w.r.t. feats.  `cache`: a dict that lives as long
    as idx is valid (TileGeometry.cache) and keeps the argsort the backward pass needs."""
    if feats.is_cuda and feats.dim() == 2 and idx.dtype == torch.int64 and feats.dtype in (torch.float32, torch.bfloat16, torch.float16) \
            and (feats.shape[1] * feats.element_size()) % 16 == 0 and feats.shape[0] > 0 and idx.numel() > 0:
        return _GatherRowsFn.apply(feats, idx, cache)
    return feats[idx]


class _LinearSmallFn(torch.autograd.Function):
    """The output Linear of a head (32 -> 2 / 3) with an fp32 RESULT from 16-bit inputs (ops.linear_small_f32): under mixed precision the
    offsets keep fp32 resolution (bf16 would quantise 8-16 m to 3-6 cm).  Backward: the incoming fp32 gradient is rounded to x's dtype
    once and takes the same kernels as any 1x1 conv."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        out = ops.linear_small_f32(x, _packed(weight, x.dtype))
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import backward as bw
        x, weight = ctx.saved_tensors
        n = x.shape[0]
        gx, gw = bw.conv_backward(x, weight, bw.TableRef(None, n, None, n, False), g.to(x.dtype).contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw


def linear_small_f32(x, weight5):
    """x . W^T -> fp32 for a [Cout <= 8, 1, 1, 1, Cin] weight view, or None when the shape is not served (caller falls back)."""
    if x.is_cuda and x.dim() == 2 and x.shape[1] % 8 == 0 and weight5.shape[0] <= 8 and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and x.shape[0] > 0:
        return _LinearSmallFn.apply(x, weight5)
    return None
