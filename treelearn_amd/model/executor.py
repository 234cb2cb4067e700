"""The C-side forward executor: ONE ctypes call per tile (`tl_forward`, include/treelearn_hip.h, csrc/tl_exec.hip).

The reference's tile loop calls `model(batch, return_loss=False)` once per tile (tree_learn/util/pipeline.py:86).  The Python-driven engine
(model/engine.py) turns that into ~80 ctypes calls; this module describes the same network once as a `tl_net_desc` (device pointers of the
InferencePlan's packed weights and folded BatchNorms) and hands every forward to the library in one call: voxel hashing, rulebooks, the 72
conv launches and the heads are enqueued from C, the two geometry read-backs happen there, and all intermediate device memory comes out of
one arena per (device, stream) that this module keeps and grows on demand.  Served: the reference's default configuration (all-ones voxel
features) on the pre-activated engine with every developer switch at its default; anything else stays on the Python-driven engine
(`supported()`), which issues the same launches -- results are bit-identical (tests/test_gpu_exec.py).
"""
import ctypes
import os
import threading

import torch

from .. import _hip, ops

_ENV_DEFAULTS = (("TL_BLK_PRO", "1"), ("TL_NO_ONES_TABLE", "0"), ("TL_NO_COMPACT", "0"), ("TL_ENGINE", "preact"), ("TL_GEOM", ""), ("TL_EXEC", "1"))
KINDS = ("subm", "down", "inverse", "1x1", "input")


def _aff(dst, pair):
    dst.scale = pair[0].data_ptr(); dst.shift = pair[1].data_ptr()


def _wt(dst, w):
    if w is None:
        return
    dst.w = w.data_ptr(); dst.frag = _hip.ptr(getattr(w, "_tl_frag", None)); dst.x3 = _hip.ptr(getattr(w, "_tl_x3", None))
    dst.K, dst.Cout, dst.Cin = w.shape


def _res(dst, b):
    _aff(dst.bn0, b.bn0); _wt(dst.w1, b.w1); _aff(dst.bn3, b.bn3); _wt(dst.w2, b.w2); _wt(dst.w1x1, b.w1x1)
    if b.w1_halves is not None:
        _wt(dst.w1_half[0], b.w1_halves[0]); _wt(dst.w1_half[1], b.w1_halves[1])


class Executor:
    """`tl_net_desc` of one InferencePlan + the per-stream contexts and arenas of its forwards."""

    def __init__(self, plan, model):
        import weakref
        self._plan_ref = weakref.ref(plan)        # (the plan owns this object: no reference cycle, so `del model` frees the arenas at once)
        self._keep = (plan.w_in, plan.so, plan.ho, plan.w1, plan.b1, plan.w2, plan.b2)
        d = self.desc = _hip.NetDesc()
        d.dtype = _hip.dtype_code(plan.dtype)
        d.num_levels = model.num_blocks
        d.voxel_size = float(model.voxel_size)
        d.has_shape = int(model.spatial_shape is not None)
        if model.spatial_shape is not None:
            d.spatial_shape[:] = [int(v) for v in model.spatial_shape]
        d.in_channels = plan.w_in.shape[2]
        _wt(d.w_in, plan.w_in)
        u, li = plan.unet, 0
        while True:
            ud = d.u[li]
            ud.C = u.C; ud.deeper = int(u.deeper)
            for i, b in enumerate(u.blocks):
                _res(ud.blocks[i], b)
            if not u.deeper:
                break
            _aff(ud.bn_down, u.bn_down); _wt(ud.wd, u.wd); _aff(ud.bn_up, u.bn_up); _wt(ud.wu, u.wu)
            for i, b in enumerate(u.tail):
                _res(ud.tail[i], b)
            _aff(ud.bn_cat_l, u.bn_cat_l); _aff(ud.bn_cat_r, u.bn_cat_r)
            u = u.u; li += 1
        assert li + 1 == d.num_levels
        _aff(d.out_bn, (plan.so, plan.ho))
        d.head_w1 = plan.w1.data_ptr(); d.head_b1 = plan.b1.data_ptr(); d.head_w2 = plan.w2.data_ptr(); d.head_b2 = plan.b2.data_ptr()
        self.head_C = u_top_C = plan.unet.C
        assert u_top_C in ops.HEAD_WIDTHS
        self._ctx = {}          # (device index, stream handle) -> [tl_exec*, arena tensor]
        self.profiling = False

    @staticmethod
    def supported(plan, model):
        """The configurations tl_forward serves (everything else: the Python-driven engine)."""
        if not plan.preact or model.use_coords or model.use_feats or not (2 <= model.num_blocks <= _hip.TL_MAX_LEVELS):
            return False
        if plan.unet.C not in ops.HEAD_WIDTHS or plan.w_in.shape[0] != 27:
            return False
        return True

    @staticmethod
    def env_default():
        """Developer switches that change what the Python-driven engine issues are honoured by staying on that engine."""
        return ops.PROFILE is None and all(os.environ.get(k, dv) == dv for k, dv in _ENV_DEFAULTS)

    def __del__(self):
        try:
            L = _hip.lib()
            for c in self._ctx.values():
                L.tl_exec_destroy(c[0])
        except Exception:                                        # noqa: BLE001  (interpreter shutdown)
            pass

    def release_memory(self):
        """Drop the arenas (they are re-allocated by the next forward)."""
        for c in self._ctx.values():
            c[1] = None

    def _context(self, dev, stream):
        key = (dev.index, stream.cuda_stream)
        c = self._ctx.get(key)
        if c is None:
            ex = _hip.lib().tl_exec_create()
            if not ex:
                raise RuntimeError("tl_exec_create failed")
            c = self._ctx[key] = [ex, None, threading.Lock()]          # handle, arena, and a lock: one forward at a time per (device, stream) context
        return c

    def profile(self, enable):
        """Live per-launch HIP-event timing of the following forwards (bench.py's roofline pass)."""
        self.profiling = bool(enable)
        for c in self._ctx.values():
            _hip.lib().tl_exec_profile(c[0], int(self.profiling))

    def profile_read(self, dev=None, stream=None):
        """Launch records of the last profiled forward on the current stream: list of dicts (level, kind, K, Cin, Cout, n_out, n_in, residual, esize,
        split, in_scale, ms)."""
        stream = stream or torch.cuda.current_stream()
        dev = dev or torch.device("cuda", torch.cuda.current_device())
        ex = self._context(dev, stream)[0]
        buf = (_hip.LaunchRec * 512)()
        n = _hip.lib().tl_exec_profile_read(ex, buf, 512)
        if n < 0:
            _hip.check(n, "tl_exec_profile_read")
        out = []
        for r in buf[:n]:
            out.append(dict(level=r.level, kind=KINDS[r.kind], K=r.K, Cin=r.Cin, Cout=r.Cout, n_out=r.n_out, n_in=r.n_in, residual=bool(r.residual),
                            esize=r.esize, split=(r.split_part, r.split_cin) if r.split_part >= 0 else None, in_scale=bool(r.in_prologue), ms=r.ms))
        return out

    def forward(self, coords, batch_ids, batch_size, want_backbone=True):
        """coords f32[N, 3], batch_ids i64[N] on the device -> (backbone f32[N, C] or None, logits f32[N, 2], offsets f32[N, 3])."""
        L = _hip.lib()
        _hip.require_cuda(coords, "coords"); _hip.require_cuda(batch_ids, "batch_ids")
        if coords.dtype != torch.float32 or batch_ids.dtype != torch.int64:
            raise TypeError("coords must be float32 [N,3] and batch_ids int64 [N]")
        N = coords.shape[0]
        if N == 0:
            raise ValueError("empty tile")
        dev = coords.device
        stream = torch.cuda.current_stream(dev)
        ctx = self._context(dev, stream)
        if self.profiling:
            L.tl_exec_profile(ctx[0], 1)
        plan = self._plan_ref()
        if plan is None:
            raise RuntimeError("the InferencePlan of this executor is gone")
        self.desc.blocked = int(plan.supports_blocked())
        bb = torch.empty((N, self.head_C), dtype=torch.float32, device=dev) if want_backbone else None
        logits = torch.empty((N, 2), dtype=torch.float32, device=dev)
        offsets = torch.empty((N, 3), dtype=torch.float32, device=dev)
        a = _hip.ForwardArgs()
        a.xyz = coords.data_ptr(); a.batch_ids = batch_ids.data_ptr(); a.N = N; a.B = int(batch_size)
        a.backbone = _hip.ptr(bb); a.logits = logits.data_ptr(); a.offsets = offsets.data_ptr()
        # the unit builder beside the other levels' rulebook kernels for a lone forward on the default stream (geometry.build_geometry's rule)
        want_side = os.environ.get("TL_BLK_SIDE")
        use_side = (stream == torch.cuda.default_stream(dev)) if want_side is None else want_side != "0"
        if use_side:
            from ..geometry import _side_stream
            a.side_stream = _side_stream(dev).cuda_stream
        with ctx[2]:                                                   # (ctypes releases the GIL: two Python threads on one stream would share read-back buffer and arena)
            for attempt in range(4):
                if ctx[1] is None:
                    ctx[1] = torch.empty(max(1 << 20, int(N * 1536)), dtype=torch.uint8, device=dev)      # first guess: ~1.5 KB per point
                a.arena = ctx[1].data_ptr(); a.arena_bytes = ctx[1].numel()
                rc = L.tl_forward(ctx[0], ctypes.byref(self.desc), ctypes.byref(a), stream.cuda_stream)
                if rc != _hip.TL_ERR_ARENA:
                    break
                ctx[1] = None                                          # grow: the exact figure when the level counts were known, a guess before that
                ctx[1] = torch.empty(int(a.needed_bytes * 1.15) + (1 << 20), dtype=torch.uint8, device=dev)
        if rc == _hip.TL_ERR_REACH_ZERO:
            raise ValueError("sparse conv output spatial shape reach zero!!! (a level of the tile is empty or its spatial shape collapsed)")
        if rc == _hip.TL_ERR_EXTENT:
            raise ValueError("voxelize: tile extent exceeds spatial_shape, batch id out of range or voxel coordinate outside [0, 65536)")
        _hip.check(rc, "tl_forward")
        self.last = dict(level_n=list(a.level_n[:self.desc.num_levels]), blocked=bool(a.blocked_used), launches=a.launches, arena_bytes=a.needed_bytes)
        return bb, logits, offsets
