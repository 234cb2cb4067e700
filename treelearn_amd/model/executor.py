"""The C-side forward executor: ONE ctypes call per tile (`tl_forward`, include/treelearn_hip.h, csrc/tl_exec.hip).

The reference's tile loop calls `model(batch, return_loss=False)` once per tile (tree_learn/util/pipeline.py:86).  The Python-driven engine
(model/engine.py) turns that into ~80 ctypes calls; this module describes the same network once as a `tl_net_desc` (device pointers of the
InferencePlan's packed weights and folded BatchNorms) and hands every forward to the library in one call: voxel hashing, rulebooks, the 72
conv launches and the heads are enqueued from C, the two geometry read-backs happen there, and all intermediate device memory comes out of
one arena per (device, stream) that this module keeps and grows on demand -- at most `MAX_CONTEXTS` of them, least recently used first out.
Served: the yaml's configuration (all-ones voxel features) and the reference constructor's own defaults (`use_feats=True`,
tree_learn.py:18: voxel-mean features) on the pre-activated engine with every developer switch at its default; anything else stays on the
Python-driven engine (`supported()`), which issues the same launches -- results are bit-identical (tests/test_gpu_exec.py).
"""
import collections
import ctypes
import os
import threading

import torch

from .. import _hip, ops

MAX_CONTEXTS = int(os.environ.get("TL_EXEC_MAX_CONTEXTS", "8"))      # (device, stream) contexts -- each with its arena -- an executor keeps alive
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
        d.use_coords = int(bool(model.use_coords)); d.use_feats = int(bool(model.use_feats))
        d.max_points_per_voxel = int(model.max_num_points_per_voxel)
        self.needs_feats = bool(model.use_coords or model.use_feats)
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
        self._ctx = collections.OrderedDict()      # (device index, stream handle) -> [tl_exec*, arena tensor, lock], least recently used first
        self._ctx_lock = threading.Lock()
        self.profiling = False

    @staticmethod
    def supported(plan, model):
        """The configurations tl_forward serves (everything else: the Python-driven engine)."""
        if not plan.preact or not (2 <= model.num_blocks <= _hip.TL_MAX_LEVELS):
            return False
        if plan.unet.C not in ops.HEAD_WIDTHS or plan.w_in.shape[0] != 27:
            return False
        if (model.use_coords or model.use_feats) and not (3 < plan.w_in.shape[2] <= 8):
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
        """Drop the arenas (they are re-allocated by the next forward).  Waits for a forward that is inside the library on another thread."""
        with self._ctx_lock:
            ctxs = list(self._ctx.items())
        for k, c in ctxs:
            with c[2]:
                self._drop_arena(c, torch.cuda.ExternalStream(k[1], device=torch.device("cuda", k[0])) if k[1] else None)

    def _context(self, dev, stream, pin=False):
        """The (device, stream) context, most recently used last.  Beyond MAX_CONTEXTS the least recently used one that no thread is inside
        goes: its arena returns to torch's allocator (stream-ordered: kernels already enqueued on its stream finish first), its handle is
        destroyed.  A caller that runs every request on a fresh stream therefore holds MAX_CONTEXTS arenas, not one per stream it ever used."""
        key = (dev.index, stream.cuda_stream)
        evicted = []
        with self._ctx_lock:
            c = self._ctx.get(key)
            if c is None:
                ex = _hip.lib().tl_exec_create()
                if not ex:
                    raise RuntimeError("tl_exec_create failed")
                c = self._ctx[key] = [ex, None, threading.Lock(), 0]   # handle, arena, a lock (one forward at a time per context), callers holding it
            else:
                self._ctx.move_to_end(key)
            if pin:
                c[3] += 1                                              # (released by the caller through _unpin: an eviction must not take a context
                                                                       #  between this look-up and the caller's acquisition of its lock)
            if len(self._ctx) > max(MAX_CONTEXTS, 1):
                for k in list(self._ctx.keys()):
                    if len(self._ctx) <= max(MAX_CONTEXTS, 1):
                        break
                    old = self._ctx[k]
                    if k != key and old[3] == 0 and old[2].acquire(blocking=False):   # (a context another thread holds or runs a forward on stays)
                        del self._ctx[k]
                        evicted.append((k, old))
        for k, old in evicted:
            try:
                self._raise_if_flagged(_hip.lib().tl_exec_check(old[0]))
            finally:
                self._drop_arena(old, torch.cuda.ExternalStream(k[1], device=torch.device("cuda", k[0])) if k[1] else None)
                _hip.lib().tl_exec_destroy(old[0])
                old[2].release()
        return c

    def _unpin(self, c):
        with self._ctx_lock:
            c[3] -= 1

    @staticmethod
    def _new_arena(nbytes, dev, stream):
        """Arenas come out of the DEFAULT stream's pool of torch's caching allocator whatever stream the forward runs on: freed blocks of
        evicted / shrunk arenas are then reusable by every other context instead of staying cached per stream handle."""
        d0 = torch.cuda.default_stream(dev)
        if stream == d0:
            return torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with torch.cuda.stream(d0):
            t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        stream.wait_stream(d0)                                         # the block's previous owner may still be running on the default stream
        return t

    @staticmethod
    def _drop_arena(ctx, stream):
        t, ctx[1] = ctx[1], None
        if t is not None and stream is not None and stream != torch.cuda.default_stream(t.device):
            t.record_stream(stream)                                    # kernels of this context's stream may still be reading / writing it

    @staticmethod
    def _raise_if_flagged(rc):
        if rc == _hip.TL_ERR_BLK:
            raise RuntimeError("tl_forward: the block-local unit builder skipped units in an EARLIER forward on this stream (internal assertion); "
                               "that tile's outputs are invalid -- rerun it with TL_BLK=0")
        _hip.check(rc, "tl_exec_check")

    def check(self, dev=None, stream=None):
        """Verdict of the unit builder's assertion for every forward enqueued so far on (dev, stream) -- default: every context of this executor.
        Waits for a 4-byte read-back behind the LAST forward's geometry kernels (not its convs).  The tile loop calls this once after its last
        tile (util/pipeline.get_pointwise_preds); `TL_EXEC_CHECK=1` makes every forward call it on the way out (a per-tile verdict, at the price
        of waiting for the tile's geometry)."""
        if stream is not None:
            dev = dev or torch.device("cuda", torch.cuda.current_device())
            with self._ctx_lock:
                c = self._ctx.get((dev.index, stream.cuda_stream))
            ctxs = [c] if c is not None else []
        else:
            with self._ctx_lock:
                ctxs = list(self._ctx.values())
        for c in ctxs:
            with c[2]:
                self._raise_if_flagged(_hip.lib().tl_exec_check(c[0]))

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

    def forward(self, coords, batch_ids, batch_size, want_backbone=True, input_feats=None):
        """coords f32[N, 3], batch_ids i64[N] (and, for a net built with use_coords / use_feats, input_feats f32[N, dim_feat]) on the device ->
        (backbone f32[N, C] or None, logits f32[N, 2], offsets f32[N, 3])."""
        L = _hip.lib()
        _hip.require_cuda(coords, "coords"); _hip.require_cuda(batch_ids, "batch_ids")
        if coords.dtype != torch.float32 or batch_ids.dtype != torch.int64:
            raise TypeError("coords must be float32 [N,3] and batch_ids int64 [N]")
        N = coords.shape[0]
        if self.needs_feats:
            if input_feats is None:
                raise ValueError("this network reads point features (use_coords / use_feats): pass input_feats")
            _hip.require_cuda(input_feats, "input_feats")
            if input_feats.dtype != torch.float32 or not input_feats.is_contiguous() or tuple(input_feats.shape) != (N, self.desc.in_channels - 3):
                raise TypeError(f"input_feats must be contiguous float32 [N, {self.desc.in_channels - 3}]")
        if N == 0:
            raise ValueError("empty tile")
        dev = coords.device
        stream = torch.cuda.current_stream(dev)
        ctx = self._context(dev, stream, pin=True)
        try:
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
            a.point_feats = input_feats.data_ptr() if self.needs_feats else None
            # the unit builder beside the other levels' rulebook kernels for a lone forward on the default stream (geometry.build_geometry's rule)
            want_side = os.environ.get("TL_BLK_SIDE")
            use_side = (stream == torch.cuda.default_stream(dev)) if want_side is None else want_side != "0"
            if use_side:
                from ..geometry import _side_stream
                a.side_stream = _side_stream(dev).cuda_stream
            with ctx[2]:                                                   # (ctypes releases the GIL: two Python threads on one stream would share read-back buffer and arena)
                for attempt in range(4):
                    if ctx[1] is None:
                        ctx[1] = self._new_arena(max(1 << 20, int(N * 1536)), dev, stream)                    # first guess: ~1.5 KB per point
                    a.arena = ctx[1].data_ptr(); a.arena_bytes = ctx[1].numel()
                    rc = L.tl_forward(ctx[0], ctypes.byref(self.desc), ctypes.byref(a), stream.cuda_stream)
                    if rc != _hip.TL_ERR_ARENA:
                        break
                    self._drop_arena(ctx, stream)                          # grow: the exact figure when the level counts were known, a guess before that
                    ctx[1] = self._new_arena(int(a.needed_bytes * 1.15) + (1 << 20), dev, stream)
                # an arena far larger than this context's tiles need (a one-off big tile) goes back: the next forward allocates what it needs
                if rc == _hip.TL_OK and ctx[1].numel() > 4 * int(a.needed_bytes) + (64 << 20):
                    self._drop_arena(ctx, stream)
                if rc == _hip.TL_OK and os.environ.get("TL_EXEC_CHECK") == "1":
                    rc = L.tl_exec_check(ctx[0])                             # per-tile verdict of the unit builder's assertion (waits for this tile's geometry)
        finally:
            self._unpin(ctx)
        if rc == _hip.TL_ERR_BLK:
            self._raise_if_flagged(rc)
        if rc == _hip.TL_ERR_REACH_ZERO:
            raise ValueError("sparse conv output spatial shape reach zero!!! (a level of the tile is empty or its spatial shape collapsed)")
        if rc == _hip.TL_ERR_EXTENT:
            raise ValueError("voxelize: tile extent exceeds spatial_shape, batch id out of range or voxel coordinate outside [0, 65536)")
        _hip.check(rc, "tl_forward")
        self.last = dict(level_n=list(a.level_n[:self.desc.num_levels]), blocked=bool(a.blocked_used), launches=a.launches, arena_bytes=a.needed_bytes)
        return bb, logits, offsets
