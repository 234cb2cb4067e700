"""Thin host wrappers over the HIP operators (tl_conv_fwd, tl_head_mlp, ...)."""
import ctypes

import os

import torch

from . import _hip
from .geometry import BlockedRulebook

# When set to a list, every conv_fwd launch is bracketed by HIP events on the launch stream and
# (start, end, meta) is appended -- bench.py's live per-kernel timing.  None = no overhead.
PROFILE = None
BLK_LAUNCHES = 0        # conv_fwd calls that carried the block-local rulebook form (a test / tool counter)

HEAD_WIDTHS = (8, 16, 32, 64)      # backbone widths tl_head_mlp is instantiated for
PACK_X3 = False         # while True (engine.InferencePlan(..., x3=True)), fp32 weights also get their split-bf16 copy (tl_pack_weight_x3)


def pack_weight(w_ref: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Reference conv weight [Cout,k,k,k,Cin] (spconv layout) -> kernel layout [K,Cout,Cin] in `dtype`."""
    L = _hip.lib()
    w = w_ref.detach().contiguous().float()
    _hip.require_cuda(w, "weight")
    co, ci = w.shape[0], w.shape[-1]
    K = w.numel() // (co * ci)
    out = torch.empty((K, co, ci), dtype=dtype, device=w.device)
    _hip.check(L.tl_pack_weight(_hip.ptr(w), co, K, ci, _hip.ptr(out), _hip.dtype_code(dtype), _hip.stream()), "tl_pack_weight")
    if co % 32 == 0 and ci % 32 == 0 and K > 1:
        # second copy in MFMA-fragment order for the kernels that read B operands straight from global memory (small levels);
        # it rides on the packed tensor as an attribute so that callers keep passing one object
        frag = torch.empty(K * co * ci, dtype=dtype, device=w.device)
        _hip.check(L.tl_pack_weight_frag(_hip.ptr(w), co, K, ci, _hip.ptr(frag), _hip.dtype_code(dtype), _hip.stream()), "tl_pack_weight_frag")
        out._tl_frag = frag
    if PACK_X3 and dtype == torch.float32 and ci % 32 == 0:
        # split-bf16 copy (hi / lo bf16 parts in MFMA-piece order; as many bytes as the fp32 tensor): tl_conv_args.weight_x3
        x3 = torch.empty(int(L.tl_pack_weight_x3_bytes(co, K, ci)) // 4, dtype=torch.float32, device=w.device)
        _hip.check(L.tl_pack_weight_x3(_hip.ptr(w), co, K, ci, _hip.ptr(x3), _hip.stream()), "tl_pack_weight_x3")
        out._tl_x3 = x3
    return out


def pack_weight_dgrad(w_ref: torch.Tensor, dtype: torch.dtype, flip: bool) -> torch.Tensor:
    """Reference conv weight [Cout,k,k,k,Cin] -> the [K, Cin, Cout] weights of the layer's input-gradient conv (W[k]^T, taps flipped
    for SubM), in `dtype`: one kernel instead of reshape / permute / flip / contiguous / cast."""
    L = _hip.lib()
    w = w_ref.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    _hip.require_cuda(w, "weight")
    co, ci = w.shape[0], w.shape[-1]
    K = w.numel() // (co * ci)
    out = torch.empty((K, ci, co), dtype=dtype, device=w.device)
    _hip.check(L.tl_pack_weight_dgrad(_hip.ptr(w), co, K, ci, int(bool(flip)), _hip.ptr(out), _hip.dtype_code(dtype), _hip.stream()), "tl_pack_weight_dgrad")
    return out


def _check_table(table, K, n_out, device):
    """The C ABI takes the rulebook as a bare pointer: a table of the wrong level (too few columns) would be read out of bounds on the
    device, so its shape is checked here."""
    if table is None:
        if K != 1:
            raise ValueError(f"a {K}-tap conv needs a rulebook")
        return
    if isinstance(table, BlockedRulebook):
        if K != 27 or table.n != n_out or table.unit.device != device:
            raise ValueError(f"block-local rulebook of {table.n} rows on {table.unit.device} for a {K}-tap conv over {n_out} rows on {device}")
        return
    if table.dtype != torch.int32 or table.dim() != 2 or tuple(table.shape) != (K, n_out) or not table.is_contiguous() or table.device != device:
        raise ValueError(f"rulebook must be a contiguous int32 [{K}, {n_out}] tensor on {device}, got {table.dtype} {tuple(table.shape)} on {table.device}")


def conv_fwd(x: torch.Tensor, w_packed: torch.Tensor, table, n_out: int, out: torch.Tensor = None,
             in_scale=None, in_shift=None, in_relu=False, residual=None, out_scale=None, out_shift=None, out_relu=False,
             out2=None, out3=None, one_hot=False, epi=None, all_ones=False, split=None, scatter=None):
    """out[o] = epi(sum_k W[k] . pro(x[table[k][o]])); x / out / residual may be column views of wider
    row-major buffers (their stride(0) is the leading dimension) -- that is how the skip concat is fused.

    `epi` (training): "stats" -> the kernel also sums y and y^2 per channel (the statistics of the BatchNorm that consumes the result);
    ("bn_bwd", x_bn, st, relu) -> this is the input-gradient conv of a layer fed by relu?(bn(x_bn)): the result is masked by the ReLU
    and the sums of g and g * xhat are formed (tl_conv_args.epi_mode).  Returns (out, parts f64[.,2,Cout], nparts), or None when the
    kernel family that serves the shape has no such epilogue (nothing was launched: the caller runs the separate passes).

    `table` may be a geometry.BlockedRulebook (K = 27): x / out / residual are then in the block-local row order.  `split` = (part, Cin of
    the logical conv) marks a launch as one input-channel half of a wider conv (bench.py's per-launch accounting).  `scatter` (with one_hot):
    the table's scatter form i32[K][n_in] (tl_conv_args.table_scatter) -- the inverse conv then walks its input rows."""
    L = _hip.lib()
    K, Cout, Cin = w_packed.shape
    if x.stride(1) != 1 or x.shape[1] != Cin:
        raise ValueError("bad input view")
    if out is None:
        out = torch.empty((n_out, Cout), dtype=x.dtype, device=x.device)
    if out.stride(1) != 1 or out.shape[1] != Cout or out.shape[0] != n_out:
        raise ValueError("bad output view")
    if w_packed.dtype != x.dtype or out.dtype != x.dtype:
        raise ValueError("dtype mismatch")
    _check_table(table, K, n_out, x.device)
    a = _hip.ConvArgs()
    a.in_ = x.data_ptr(); a.in_ld = x.stride(0)
    a.weight = w_packed.data_ptr()
    blk = table if isinstance(table, BlockedRulebook) else None
    if blk is not None:                          # rows in block-local order: units / halo lists / local rulebooks instead of the table
        table = blk.nn_table                     # (the plain table in the new order, if the geometry carries one: shapes the staged-unit kernel does not serve)
        global BLK_LAUNCHES
        BLK_LAUNCHES += 1
        if x.shape[0] != n_out:
            raise ValueError("a block-local SubM conv maps the level onto itself")
        a.blk_unit = blk.unit.data_ptr(); a.blk_counter = blk.counter.data_ptr(); a.blk_halo = blk.halo.data_ptr()
        a.blk_lrb = blk.lrb.data_ptr(); a.blk_pmask = blk.pmask.data_ptr()
    a.table = table.data_ptr() if table is not None else None
    a.weight_frag = _hip.ptr(getattr(w_packed, "_tl_frag", None))
    a.weight_x3 = _hip.ptr(getattr(w_packed, "_tl_x3", None))      # fp32 weights of a bf16x3 plan: the split-bf16 contraction where a kernel offers it
    a.table_one_hot = int(bool(one_hot))          # inverse conv: one valid entry per output row
    if scatter is not None:                       # ... and its scatter form (the stride-2 conv's rulebook, i32[K][n_in])
        if not one_hot or scatter.dtype != torch.int32 or tuple(scatter.shape) != (K, x.shape[0]) or not scatter.is_contiguous() or scatter.device != x.device:
            raise ValueError("scatter: the one-hot table's transpose, i32[K][n_in] on the input's device")
        a.table_scatter = scatter.data_ptr()
    a.in_all_ones = int(bool(all_ones))           # the caller guarantees x == 1 everywhere (default reference flags): presence-mask table, no gather
    a.table_compact = _hip.ptr(getattr(table, "_tl_compact", None)) if (table is not None and os.environ.get("TL_NO_COMPACT") != "1") else None
    a.n_out = n_out; a.n_in = x.shape[0]
    a.K = K; a.Cin = Cin; a.Cout = Cout; a.dtype = _hip.dtype_code(x.dtype)
    a.in_scale = in_scale.data_ptr() if in_scale is not None else None
    a.in_shift = in_shift.data_ptr() if in_shift is not None else None
    a.in_relu = int(bool(in_relu)); a.out_relu = int(bool(out_relu))
    if residual is not None:
        if residual.dtype != x.dtype or residual.stride(1) != 1:
            raise ValueError("bad residual view")
        a.residual = residual.data_ptr(); a.res_ld = residual.stride(0)
    else:
        a.residual = None; a.res_ld = 0
    a.out_scale = out_scale.data_ptr() if out_scale is not None else None
    a.out_shift = out_shift.data_ptr() if out_shift is not None else None
    a.out = out.data_ptr(); a.out_ld = out.stride(0)
    # extra views: (tensor [n_out, Cout] (may be a column view), scale or None, shift or None, relu)
    for name, spec in (("out2", out2), ("out3", out3)):
        if spec is None:
            setattr(a, name, None); setattr(a, name + "_ld", 0); setattr(a, name + "_scale", None); setattr(a, name + "_shift", None); setattr(a, name + "_relu", 0)
            continue
        t, sc, sh, relu = spec
        if t.dtype != x.dtype or t.stride(1) != 1 or t.shape[0] != n_out or t.shape[1] != Cout:
            raise ValueError(f"bad {name} view")
        setattr(a, name, t.data_ptr()); setattr(a, name + "_ld", t.stride(0))
        setattr(a, name + "_scale", sc.data_ptr() if sc is not None else None)
        setattr(a, name + "_shift", sh.data_ptr() if sh is not None else None)
        setattr(a, name + "_relu", int(bool(relu)))
    if epi is not None:
        parts = torch.empty((int(L.tl_conv_red_parts(n_out)), 2, Cout), dtype=torch.float64, device=x.device)
        nparts = ctypes.c_int32(0)
        a.red_part = parts.data_ptr(); a.red_nparts = ctypes.pointer(nparts)
        if epi == "stats":
            a.epi_mode = _hip.TL_EPI_STATS
        else:
            _, xb, st, relu = epi
            if xb.dtype != x.dtype or xb.stride(1) != 1 or xb.shape[0] != n_out or xb.shape[1] != Cout:
                raise ValueError("bad BatchNorm input view for the bn_bwd epilogue")
            a.epi_mode = _hip.TL_EPI_BN_BWD
            a.bn_x = xb.data_ptr(); a.bn_x_ld = xb.stride(0)
            a.bn_mean = st[0].data_ptr(); a.bn_rstd = st[1].data_ptr(); a.bn_scale = st[2].data_ptr(); a.bn_shift = st[3].data_ptr()
            a.bn_relu = int(bool(relu))
        rc = L.tl_conv_fwd(ctypes.byref(a), _hip.stream())
        if rc == _hip.TL_ERR_UNSUPPORTED:
            return None
        _hip.check(rc, "tl_conv_fwd")
        return out, parts, int(nparts.value)
    if PROFILE is not None:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _hip.check(L.tl_conv_fwd(ctypes.byref(a), _hip.stream()), "tl_conv_fwd")
        e1.record()
        PROFILE.append((e0, e1, dict(K=K, Cin=Cin, Cout=Cout, n_out=n_out, n_in=x.shape[0], table=blk if blk is not None else table,
                                     residual=residual is not None, esize=x.element_size(), split=split,
                                     in_scale=in_scale is not None or bool(in_relu))))
        return out
    _hip.check(L.tl_conv_fwd(ctypes.byref(a), _hip.stream()), "tl_conv_fwd")
    return out


WGRAD_BLK = os.environ.get("TL_WGRAD_BLK", "1") != "0"        # level 1 (block-local rows): the staged-unit weight gradient (tl_conv_wgrad_blk)


def _conv_wgrad_blk(x, g, rb, ref_layout):
    """tl_conv_wgrad_blk over a block-local level: 32 input channels per call (a 64 -> 32 conv = its two input halves, joined along Cin).
    None when the library does not cover the call (the caller falls back to the plain table)."""
    L = _hip.lib()
    ws = torch.empty(int(L.tl_conv_wgrad_blk_ws_floats()), dtype=torch.float32, device=x.device)
    parts = []
    for c0 in range(0, x.shape[1], 32):
        xs = x[:, c0:c0 + 32]
        gw = torch.empty((32, 27, 32) if ref_layout else (27, 32, 32), dtype=torch.float32, device=x.device)
        rc = L.tl_conv_wgrad_blk(_hip.ptr(xs), xs.stride(0), _hip.ptr(g), g.stride(0), _hip.dtype_code(x.dtype), _hip.ptr(rb.unit), _hip.ptr(rb.counter), _hip.ptr(rb.halo),
                                 _hip.ptr(rb.lrb), rb.n, 32, 32, _hip.ptr(gw), int(bool(ref_layout)), _hip.ptr(ws), _hip.stream())
        if rc == _hip.TL_ERR_UNSUPPORTED:
            return None
        _hip.check(rc, "tl_conv_wgrad_blk")
        parts.append(gw)
    out = parts[0] if len(parts) == 1 else torch.cat(parts, dim=2)
    out._tl_ref_layout = bool(ref_layout)
    return out


def conv_wgrad(x: torch.Tensor, grad_out: torch.Tensor, table, n_out: int, K: int, ref_layout: bool = False) -> torch.Tensor:
    """gW[k][co][ci] = sum_o grad_out[o][co] * x[table[k][o]][ci] (fp32, present entries only) -> [K, Cout, Cin]; with ref_layout the
    result comes back as [Cout, K, Cin], the layout of the module parameter (spconv `.weight` [Cout,k,k,k,Cin])."""
    L = _hip.lib()
    if x.dtype in (torch.bfloat16, torch.float16) and grad_out.dtype == x.dtype:
        g = grad_out                                     # mixed precision: the 16-bit operands go to the matrix cores as they are
    else:
        x = x.float(); g = grad_out.float()
    if x.stride(1) != 1: x = x.contiguous()
    if g.stride(1) != 1: g = g.contiguous()
    if not (x.is_cuda and g.is_cuda):
        raise RuntimeError("conv_wgrad: tensors must live on the GPU; the HIP path has no CPU fallback")
    ci, co = x.shape[1], g.shape[1]
    if g.shape[0] != n_out:
        raise ValueError(f"grad_out has {g.shape[0]} rows, the rulebook {n_out}")
    # (32 -> 32 only: the 64 -> 32 decoder conv as two staged halves measured no faster than one dense-over-taps launch, 0.755 against 0.732 ms)
    if isinstance(table, BlockedRulebook) and K == 27 and co == 32 and ci == 32 and x.dtype != torch.float32 and WGRAD_BLK:
        gw = _conv_wgrad_blk(x, g, table, ref_layout)
        if gw is not None:
            return gw
    if isinstance(table, BlockedRulebook):
        if table.nn_table is None:
            raise ValueError("the weight gradient over a block-local level needs the geometry's nn_table (build_geometry(nn_table=True))")
        table = table.nn_table
    _check_table(table, K, n_out, x.device)
    ref_layout = bool(ref_layout) and ci % 4 == 0
    gw = torch.empty((co, K, ci) if ref_layout else (K, co, ci), dtype=torch.float32, device=x.device)
    ws = torch.empty(int(L.tl_conv_wgrad_ws_floats(n_out, K, ci, co)), dtype=torch.float32, device=x.device)
    _hip.check((L.tl_conv_wgrad_ref if ref_layout else L.tl_conv_wgrad)(_hip.ptr(x), x.stride(0), _hip.ptr(g), g.stride(0), _hip.dtype_code(x.dtype), _hip.ptr(table) if table is not None else None, n_out, x.shape[0],
                               K, ci, co, _hip.ptr(gw), _hip.ptr(ws), _hip.stream()), "tl_conv_wgrad")
    gw._tl_ref_layout = ref_layout                # the layout actually produced ([Cout, K, Cin] needs Cin % 4 == 0)
    return gw


def head_mlp(feats, v2p, pro_scale, pro_shift, w1, b1, w2, b2, want_backbone=True):
    L = _hip.lib()
    N = v2p.shape[0]
    C = feats.shape[1]
    dev = feats.device
    backbone = torch.empty((N, C), dtype=torch.float32, device=dev) if want_backbone else None
    logits = torch.empty((N, 2), dtype=torch.float32, device=dev)
    offsets = torch.empty((N, 3), dtype=torch.float32, device=dev)
    _hip.check(L.tl_head_mlp(_hip.ptr(feats), feats.stride(0), _hip.dtype_code(feats.dtype), C, _hip.ptr(v2p), N,
                             _hip.ptr(pro_scale), _hip.ptr(pro_shift), _hip.ptr(w1), _hip.ptr(b1), _hip.ptr(w2), _hip.ptr(b2),
                             _hip.ptr(backbone), _hip.ptr(logits), _hip.ptr(offsets), _hip.stream()), "tl_head_mlp")
    return backbone, logits, offsets


def compact_rows(x: torch.Tensor, mask: torch.Tensor):
    """x[mask] on the device without a boolean-index sync storm: returns (buffer [n,C], count tensor)."""
    L = _hip.lib()
    n, C = x.shape
    m8 = mask.to(torch.uint8).contiguous()
    out = torch.empty_like(x)
    count = torch.empty(1, dtype=torch.int32, device=x.device)
    ws = torch.empty(int(L.tl_compact_ws_words(n)), dtype=torch.int32, device=x.device)
    _hip.check(L.tl_compact_rows(_hip.ptr(x), C, _hip.ptr(m8), n, _hip.ptr(out), _hip.ptr(count), _hip.ptr(ws), _hip.stream()), "tl_compact_rows")
    return out, count


def affine_relu(x, scale, shift, relu, out_dtype=None):
    """y = relu?(x * scale + shift) over the rows of x (tl_affine_relu); x and y may differ in dtype only through `out_dtype`
    == x.dtype (the kernel is single-dtype), so a bf16 result of an fp32 input is produced by the caller's cast."""
    L = _hip.lib()
    if not x.is_cuda or x.stride(1) != 1:
        raise RuntimeError("affine_relu: x must live on the GPU with unit column stride (a column view of a wider matrix is fine); no CPU fallback")
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _hip.check(L.tl_affine_relu(_hip.ptr(x), x.stride(0), _hip.ptr(y), y.stride(0), x.shape[0], x.shape[1], _hip.dtype_code(x.dtype),
                                _hip.ptr(scale), _hip.ptr(shift), int(bool(relu)), _hip.stream()), "tl_affine_relu")
    return y


def bn_train_stats(x, gamma, beta, eps, momentum, running_mean=None, running_var=None, num_batches_tracked=None):
    """Batch statistics of x [n, C] -> (mean, rstd, scale, shift), running statistics updated in place (tl_bn_train_stats)."""
    L = _hip.lib()
    _hip.require_cuda(x, "x")
    n, C = x.shape
    dev = x.device
    ws = torch.empty(int(L.tl_bn_ws_doubles(n, C)), dtype=torch.float64, device=dev)
    st = torch.empty((4, C), dtype=torch.float32, device=dev)
    _hip.check(L.tl_bn_train_stats(_hip.ptr(x), x.stride(0), n, C, _hip.dtype_code(x.dtype), _hip.ptr(gamma), _hip.ptr(beta), float(eps), float(momentum),
                                   _hip.ptr(ws), _hip.ptr(st[0]), _hip.ptr(st[1]), _hip.ptr(st[2]), _hip.ptr(st[3]), _hip.ptr(running_mean),
                                   _hip.ptr(running_var), _hip.ptr(num_batches_tracked), _hip.stream()), "tl_bn_train_stats")
    return st


def bn_train_finish(segments, n, gamma, beta, eps, momentum, running_mean=None, running_var=None, num_batches_tracked=None):
    """Batch statistics from conv-epilogue partial sums (tl_bn_train_finish): `segments` = [(parts f64[.,2,Ci], nparts, Ci), ...] covers
    the channels in order (one segment per producer: the two halves of a skip concat come from different convs) -> st [4, C] as
    bn_train_stats returns it; running statistics updated in place."""
    L = _hip.lib()
    C = sum(sg[2] for sg in segments)
    st = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    c0 = 0
    for i, (parts, nparts, Ci) in enumerate(segments):
        sl = slice(c0, c0 + Ci)
        _hip.check(L.tl_bn_train_finish(_hip.ptr(parts), nparts, n, Ci, _hip.ptr(gamma[sl]), _hip.ptr(beta[sl]), float(eps), float(momentum),
                                        _hip.ptr(st[0, sl]), _hip.ptr(st[1, sl]), _hip.ptr(st[2, sl]), _hip.ptr(st[3, sl]),
                                        _hip.ptr(running_mean[sl]) if running_mean is not None else None,
                                        _hip.ptr(running_var[sl]) if running_var is not None else None,
                                        _hip.ptr(num_batches_tracked) if (num_batches_tracked is not None and i == 0) else None, _hip.stream()),
                   "tl_bn_train_finish")
        c0 += Ci
    return st


def bn_train_bwd_from_parts(x, g, st, parts, nparts, dx_add=None, dx=None, dgb=None):
    """(dx, dgamma, dbeta) from an already masked g and the conv epilogue's partial sums of g and g * xhat (tl_bn_train_bwd_from_parts);
    None if the views do not allow the vector kernel.  x, g, st, dx_add, dx (out) and dgb (out, [2, C]) may be column slices of wider
    tensors: a layer whose input gradient is computed in channel slices runs this once per slice."""
    L = _hip.lib()
    n, C = x.shape
    if dx is None:
        dx = torch.empty((n, C), dtype=x.dtype, device=x.device)
    if dgb is None:
        dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
    if dx_add is not None and (dx_add.shape != x.shape or dx_add.dtype != x.dtype or dx_add.stride(1) != 1):
        raise ValueError("dx_add must match x in shape and dtype")
    rc = L.tl_bn_train_bwd_from_parts(_hip.ptr(x), x.stride(0), _hip.dtype_code(x.dtype), _hip.ptr(g), g.stride(0), _hip.dtype_code(g.dtype), n, C,
                                      _hip.ptr(st[0]), _hip.ptr(st[1]), _hip.ptr(st[2]), _hip.ptr(st[3]), _hip.ptr(parts), nparts,
                                      _hip.ptr(dgb[0]), _hip.ptr(dgb[1]), _hip.ptr(dx), dx.stride(0), _hip.ptr(dx_add),
                                      dx_add.stride(0) if dx_add is not None else 0, _hip.stream())
    if rc == _hip.TL_ERR_UNSUPPORTED:
        return None
    _hip.check(rc, "tl_bn_train_bwd_from_parts")
    return dx, dgb[0], dgb[1]


def column_sum(x):
    """sum over the rows of x [n, C] -> f32[C] on the BatchNorm statistics kernel (fp64 partial sums in a fixed order: deterministic;
    one read of x at memory speed -- ATen's reduction over dim 0 of a [3.7 M, 32] matrix runs five times slower)."""
    n, C = x.shape
    ones = torch.ones(C, dtype=torch.float32, device=x.device)
    st = bn_train_stats(x, ones, torch.zeros_like(ones), 1e-5, 0.0)
    return st[0] * float(n)                                  # mean * n


def bn_train_bwd(x, dy, st, relu, dx_add=None):
    """(dx, dgamma, dbeta) of y = relu?(batchnorm_train(x)) given dy (tl_bn_train_bwd); st = bn_train_stats' result; dx_add (x's dtype
    and shape) is added to dx in the same pass."""
    L = _hip.lib()
    n, C = x.shape
    dev = x.device
    ws = torch.empty(int(L.tl_bn_ws_doubles(n, C)), dtype=torch.float64, device=dev)
    dx = torch.empty((n, C), dtype=x.dtype, device=dev)
    dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
    if dx_add is not None:
        if dx_add.shape != x.shape or dx_add.dtype != x.dtype or dx_add.stride(1) != 1 or dx_add.device != dev:
            raise ValueError("dx_add must match x in shape, dtype and device")
    _hip.check(L.tl_bn_train_bwd(_hip.ptr(x), x.stride(0), _hip.dtype_code(x.dtype), _hip.ptr(dy), dy.stride(0), _hip.dtype_code(dy.dtype), n, C,
                                 _hip.ptr(st[0]), _hip.ptr(st[1]), _hip.ptr(st[2]), _hip.ptr(st[3]), int(bool(relu)), _hip.ptr(ws),
                                 _hip.ptr(dgb[0]), _hip.ptr(dgb[1]), _hip.ptr(dx), dx.stride(0), _hip.ptr(dx_add),
                                 dx_add.stride(0) if dx_add is not None else 0, _hip.stream()), "tl_bn_train_bwd")
    return dx, dgb[0], dgb[1]


def gather_rows(x, idx):
    """x[idx] (tl_gather_rows): x [n, C] f32 / bf16 with 16-B rows, idx i64[N]."""
    L = _hip.lib()
    _hip.require_cuda(x, "x")
    out = torch.empty((idx.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
    _hip.check(L.tl_gather_rows(_hip.ptr(x), x.stride(0), _hip.dtype_code(x.dtype), x.shape[1], x.shape[0], _hip.ptr(idx), idx.shape[0], _hip.ptr(out),
                                out.stride(0), _hip.stream()), "tl_gather_rows")
    return out


def scatter_add_rows(g, order, sorted_idx, n_rows):
    """out[v] = sum of g[p] over idx[p] == v in ascending p (tl_scatter_add_rows); order / sorted_idx = stable argsort of idx / idx[order]."""
    L = _hip.lib()
    out = torch.empty((n_rows, g.shape[1]), dtype=g.dtype, device=g.device)
    _hip.check(L.tl_scatter_add_rows(_hip.ptr(g), g.stride(0), _hip.dtype_code(g.dtype), g.shape[1], _hip.ptr(order), _hip.ptr(sorted_idx), g.shape[0], n_rows,
                                     _hip.ptr(out), out.stride(0), _hip.stream()), "tl_scatter_add_rows")
    return out


def linear_small_f32(x, w_packed):
    """x [n, Cin] (f32 / bf16) . W^T with W = w_packed[0] ([1, Cout <= 8, Cin], x's dtype) -> f32 [n, Cout] (tl_linear_small_f32); None if
    the shape is not served."""
    L = _hip.lib()
    n, ci = x.shape
    co = w_packed.shape[1]
    out = torch.empty((n, co), dtype=torch.float32, device=x.device)
    rc = L.tl_linear_small_f32(_hip.ptr(x), x.stride(0), _hip.dtype_code(x.dtype), _hip.ptr(w_packed), ci, co, n, _hip.ptr(out), out.stride(0), _hip.stream())
    if rc == _hip.TL_ERR_UNSUPPORTED:
        return None
    _hip.check(rc, "tl_linear_small_f32")
    return out
