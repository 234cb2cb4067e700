"""TreeLearn -- the drop-in for reference `tree_learn.model.TreeLearn` (tree_learn/model/tree_learn.py).

Same constructor keywords (tree_learn.py:12-24), same `forward(batch, return_loss)` contract
(:75-81), same 414 state-dict keys, same `train()` handling of
`fixed_modules` (:66-72), same "reach zero!!!" failure mode for collapsing tiles.  The device work is
done by libtreelearn_hip.so: voxel hashing + rulebooks (treelearn_amd.geometry), fused sparse convs
and heads (treelearn_amd.model.engine) -- no spconv, no CPU fallback.
"""
import functools

import os

import torch
import torch.nn as nn

from .. import geometry as G
from .. import spconv_compat as spconv
from ..util.train import cuda_cast, loss_mask_rows, point_wise_loss
from .engine import InferencePlan
from .unet import MLP, ResidualBlock, UBlock

LOSS_MULTIPLIER_SEMANTIC = 50          # tree_learn.py:9


def _drop_plan_after_load(module, incompatible):
    module.invalidate_plan()                       # (a module-level function, not a lambda: the model stays picklable)


class TreeLearn(nn.Module):
    def __init__(self, channels=32, num_blocks=7, kernel_size=3, dim_coord=3, dim_feat=1, fixed_modules=[],
                 use_feats=True, use_coords=False, spatial_shape=None, max_num_points_per_voxel=3, voxel_size=0.1,
                 compute_dtype=torch.float32, **kwargs):
        super().__init__()
        self.voxel_size = voxel_size
        self.fixed_modules = fixed_modules
        self.use_feats = use_feats
        self.use_coords = use_coords
        self.spatial_shape = spatial_shape
        self.max_num_points_per_voxel = max_num_points_per_voxel
        self.num_blocks = num_blocks
        # torch.float32 (exact parity mode) | torch.bfloat16 / torch.float16 (throughput) | "bf16x3" (parity-fast: fp32 storage, the convs of the
        # large levels contract split-bf16 parts on the bf16 matrix cores -- inside the 1e-3 gate at a third of the exact mode's time; inference)
        self.split_bf16 = isinstance(compute_dtype, str) and compute_dtype.lower() == "bf16x3"
        self.compute_dtype = torch.float32 if self.split_bf16 else compute_dtype
        self.return_backbone_feats = True           # reference always returns them (tree_learn.py:100)
        self._plan = None
        self._geom_stream = None

        norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        self.input_conv = spconv.SparseSequential(
            spconv.SubMConv3d(dim_coord + dim_feat, channels, kernel_size=kernel_size, padding=1, bias=False, indice_key='subm1'))
        self.unet = UBlock([channels * (i + 1) for i in range(num_blocks)], norm_fn, 2, ResidualBlock, kernel_size, indice_key_id=1)
        self.output_layer = spconv.SparseSequential(norm_fn(channels), nn.ReLU())
        self.semantic_linear = MLP(channels, 2, norm_fn=norm_fn, num_layers=2)
        self.offset_linear = MLP(channels, 3, norm_fn=norm_fn, num_layers=2)
        self.init_weights()
        for name in fixed_modules:
            for param in getattr(self, name).parameters():
                param.requires_grad = False
        self.register_load_state_dict_post_hook(_drop_plan_after_load)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, MLP):
                m.init_weights()

    def _plan_ok(self, dtype):
        return (self._plan is not None and self._plan.dtype == dtype and self._plan.x3 == (self.split_bf16 and dtype == torch.float32)
                and getattr(self._plan, "device", None) == self.input_conv[0].weight.device)

    def ensure_plan(self):
        """Build the fused inference plan (folded BatchNorms, packed weights) now, on the current stream.  Callers that spread
        forwards over several streams do this first: the plan is otherwise built by the first forward, on that forward's stream,
        and a forward on another stream could read weights that are still being packed."""
        if not self.training and (self._plan is None or not self._plan_ok(self.active_dtype(False))):
            self._plan = InferencePlan(self, self.active_dtype(False), x3=self.split_bf16)
        return self

    def _refresh_packed(self, dtype):
        """Training: all conv weights -> kernel layouts in ONE launch when any parameter changed (autograd.PackPlan), instead of three small
        launches per layer and step."""
        from ..autograd import PackPlan
        mods = getattr(self, "_conv_modules", None)                     # (the module tree is fixed after __init__: walk it once)
        if mods is None:
            mods = self._conv_modules = [m for m in self.modules() if isinstance(m, spconv.SparseConvolution)]
        convs = [(m.weight, bool(m.subm) and int(m.kernel_size) == 3) for m in mods
                 if m.weight.is_cuda and m.weight.dtype == torch.float32 and m.weight.is_contiguous()]
        plan = getattr(self, "_pack_plan", None)
        if plan is None or not plan.valid_for(convs, dtype):
            plan = self._pack_plan = PackPlan(convs, dtype) if convs else None
        if plan is not None:
            plan.refresh()

    # Derived state (the eval plan with its packed weights, C-side executor handles and arenas; the training pack plan; side streams) belongs
    # to THIS object and to the device memory it was built on: a copy or a pickle of the model carries the parameters only and rebuilds the rest
    # on its first forward.  (Copying it would hand one `tl_exec*` to two owners and leave the copy's descriptors pointing at the original's
    # tensors; pickling it fails on the ctypes pointers.)
    _DERIVED = ("_plan", "_pack_plan", "_geom_stream", "_conv_modules", "_last_geom")

    def __getstate__(self):
        st = dict(self.__dict__)
        for k in self._DERIVED:
            if k in st:
                st[k] = None
        return st

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in self._DERIVED else copy.deepcopy(v, memo)
        return new

    def invalidate_plan(self):
        """Drop the folded-BN / packed-weight cache (call after mutating parameters in place)."""
        self._plan = None

    def train(self, mode=True):
        if mode != self.training:                      # the packed weights of the fused plan stay valid across repeated .eval() calls
            self._plan = None
        super().train(mode)
        for name in self.fixed_modules:
            for m in getattr(self, name).modules():
                if isinstance(m, nn.BatchNorm1d):
                    m.eval()
        return self

    def _apply(self, fn, *a, **k):
        self._plan = None
        return super()._apply(fn, *a, **k)

    # ------------------------------------------------------------------ forward
    def _executor(self, dtype):
        """The C-side forward executor of the current plan (model/executor.py), or None when this configuration stays on the
        Python-driven engine."""
        from .executor import Executor
        if self._plan is None or not self._plan_ok(dtype):
            self._plan = InferencePlan(self, dtype, x3=self.split_bf16)
        ex = getattr(self._plan, "_exec", False)
        if ex is False:
            ex = self._plan._exec = Executor(self._plan, self) if Executor.supported(self._plan, self) else None
        return ex if (ex is not None and Executor.env_default()) else None

    def forward(self, batch, return_loss):
        if not return_loss and not (self.training or torch.is_grad_enabled()) and torch.is_tensor(batch['coords']):
            # eval forward in the reference's default configuration: ONE call into the library per tile (tl_forward) -- geometry, the 72
            # conv launches and the heads are enqueued from C; the module-by-module path below issues the same launches from Python
            ex = self._executor(self.active_dtype(False))
            if ex is not None:
                mv = lambda t: t if t.is_cuda else t.cuda(non_blocking=True)                         # noqa: E731
                coords = mv(batch['coords']).float().contiguous()
                bids = mv(batch['batch_ids']).long().contiguous()
                feats = mv(batch['input_feats']).float().contiguous() if ex.needs_feats else None
                bb, logits, offsets = ex.forward(coords, bids, int(batch['batch_size']), want_backbone=self.return_backbone_feats, input_feats=feats)
                return dict(backbone_feats=bb, semantic_prediction_logits=logits, offset_predictions=offsets)
        # only the tensors the backbone reads cross PCIe here (the reference's cuda_cast also ships every label /
        # mask / centre tensor of the batch, ~80 MB per tile that inference never touches; util/train.py:28-43)
        mask_rows = None
        if return_loss and os.environ.get("TL_LOSS_ROWS", "1") != "0":
            mask_rows = loss_mask_rows(batch['masks_sem'], batch['masks_off'])        # before anything is enqueued: no wait behind the forward
        backbone_output, v2p_map = self.forward_backbone(coords=batch['coords'], input_feats=batch['input_feats'],
                                                         batch_ids=batch['batch_ids'], batch_size=batch['batch_size'])
        output = self.forward_head(backbone_output, v2p_map)
        if return_loss:
            output = self.get_loss(model_output=output, mask_rows=mask_rows, **batch)
        return output

    def _voxelize(self, coords, input_feats, batch_ids, batch_size, blocked=False, nn_table=False):
        """`voxelize` of tree_learn.py:129-167 on the HIP library: geometry + voxel features
        ([M, dim_feat+dim_coord] in (feat, x, y, z) order; ones unless use_feats/use_coords).  `blocked`: level 1 in the block-local row
        order of the fused inference engine (geometry.BlockedRulebook); the voxel features are then averaged through the blocked v2p map,
        i.e. they come back in the NEW row order (the engine treats them so)."""
        geom = G.build_geometry(coords.contiguous(), batch_ids.contiguous(), int(batch_size), self.voxel_size,
                                self.num_blocks, self.spatial_shape, blocked=blocked, ref_table=(self.use_coords or self.use_feats) and not nn_table,
                                nn_table=nn_table)
        M = geom.levels[0].n
        C = coords.shape[1] + input_feats.shape[1]
        if self.use_coords or self.use_feats:
            pf = torch.hstack([coords, input_feats]).contiguous()
            mean = G.voxel_mean_feats(pf, geom, self.max_num_points_per_voxel)
            if not self.use_coords:
                mean[:, :3] = 1.0
            if not self.use_feats:
                mean[:, 3:] = 1.0
            vfeats = torch.hstack([mean[:, 3:], mean[:, :3]]).contiguous()
        else:
            vfeats = torch.ones((M, C), dtype=torch.float32, device=coords.device)
        return vfeats, geom

    def active_dtype(self, training=None):
        """The dtype the sparse convs run in for this call.  Inside `torch.cuda.amp.autocast` / `torch.autocast("cuda")` -- how the
        reference selects mixed precision (tools/training/train.py:32 `autocast(enabled=config.fp16)`) -- it is the autocast dtype,
        whatever `compute_dtype` says; outside autocast: `compute_dtype`.  float16 is served end to end: the conv / head units AND the
        training units (weight gradient, BatchNorm train forward / backward, epilogue reductions, row gather / scatter-add) are compiled a
        second time for IEEE half (csrc/tl_half.h, tl_f16_train.h), so the reference's `autocast(enabled=config.fp16)` + GradScaler step
        (tools/training/train.py:32,40-44) runs in float16 here as it does there; under a bf16 autocast it runs bf16 (no scaler needed)."""
        dt = self.compute_dtype
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
            if dt not in (torch.float16, torch.bfloat16):
                dt = torch.float32
        training = (self.training or torch.is_grad_enabled()) if training is None else training
        if dt == torch.float16 and training and os.environ.get("TL_F16_TRAIN", "1") == "0":
            dt = torch.bfloat16                     # (A/B switch: a float16 training region on the bf16 kernels, as before round 5)
        return dt

    @cuda_cast
    def forward_backbone(self, coords, input_feats, batch_ids, batch_size, **kwargs):
        dtype = self.active_dtype()
        fused = not (self.training or torch.is_grad_enabled())
        if fused and (self._plan is None or not self._plan_ok(dtype)):
            self._plan = InferencePlan(self, dtype, x3=self.split_bf16)
        # training under mixed precision: level 1 in the block-local order too (the staged-unit kernel serves its 32 -> 32 forward and
        # input-gradient convs; weight gradients and the other widths read the plain table in the new order, BlockedRulebook.nn_table)
        blk_train = (not fused and dtype in (torch.bfloat16, torch.float16) and self.unet.nPlanes[0] == 32 and os.environ.get("TL_BLK", "1") != "0"
                     and os.environ.get("TL_BLK_TRAIN", "1") != "0")
        vfeats, geom = self._voxelize(coords.float(), input_feats.float(), batch_ids.long(), batch_size,
                                      blocked=(fused and self._plan.supports_blocked()) or blk_train, nn_table=blk_train)
        if not fused:
            # module-by-module path (batch-statistics BatchNorm, autograd through the HIP convs)
            from .. import autograd as _ag
            if _ag.RELU_MASK_SINK is not None:
                self._last_geom = geom                                # test hook: the exported masks of level 1 are in this geometry's row order
            lv = geom.levels[0]
            x = spconv.SparseConvTensor(vfeats, lv.row_coords(), list(lv.shape), batch_size, geometry=geom, level=0)
            prev = spconv.SparseConvolution.amp_dtype
            spconv.SparseConvolution.amp_dtype = None if dtype == torch.float32 else dtype
            if torch.is_grad_enabled():
                _ag.new_pack_epoch()                                  # the packed copies of the last step are stale whether or not `_version` says so (fused optimizers)
                self._plan = None                                     # ... and so is an eval plan once this step's optimizer has run (a model kept in .eval()
                                                                      # while it is fine-tuned never passes through train(), which drops the plan otherwise)
            if torch.is_grad_enabled() and os.environ.get("TL_PACK_BATCH", "1") != "0":
                self._refresh_packed(dtype)                           # every conv weight packed in one launch (once per training forward)
            try:                                                       # 16-bit: mixed precision as under the reference's autocast
                x = self.output_layer(self.unet(self.input_conv(x)))
            finally:
                spconv.SparseConvolution.amp_dtype = prev
            return x, geom.v2p
        return (vfeats, geom), geom.v2p

    # ------------------------------------------------------------------ two-phase inference (software-pipelined tile loop)
    def prepare(self, batch):
        """Phase 1 of an inference forward: H2D (if needed) + voxel hashing + all rulebooks, issued on a side stream.
        Returns a handle for `infer`.  Calling `prepare(next_tile)` right after `infer(this_tile)` lets the next tile's
        geometry (small latency-bound kernels + two host syncs) run while this tile's convs occupy the main stream."""
        assert not self.training, "prepare/infer is the eval-mode fused path"
        if self._plan is None or not self._plan_ok(self.active_dtype(False)):
            self._plan = InferencePlan(self, self.active_dtype(False), x3=self.split_bf16)
        if self._geom_stream is None:
            self._geom_stream = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        self._geom_stream.wait_stream(main)                            # inputs produced on the main stream are visible
        with torch.cuda.stream(self._geom_stream), torch.no_grad():
            mv = lambda t: t.cuda(non_blocking=True) if not t.is_cuda else t                 # noqa: E731
            vfeats, geom = self._voxelize(mv(batch['coords']).float(), mv(batch['input_feats']).float(),
                                          mv(batch['batch_ids']).long(), batch['batch_size'], blocked=self._plan.supports_blocked())
            ev = torch.cuda.Event(); ev.record(self._geom_stream)
        return (vfeats, geom, ev)

    def infer(self, handle):
        """Phase 2: the fused U-Net + heads on the current stream."""
        vfeats, geom, ev = handle
        main = torch.cuda.current_stream()
        main.wait_event(ev)
        with torch.no_grad():
            bb, logits, offsets = self._plan.run(vfeats, geom, want_backbone=self.return_backbone_feats,
                                                 all_ones=not (self.use_coords or self.use_feats))
        for t in geom.tensors() + [vfeats]:                            # allocated on the side stream, consumed on this one
            t.record_stream(main)
        return dict(backbone_feats=bb, semantic_prediction_logits=logits, offset_predictions=offsets)

    def forward_head(self, backbone_output, v2p_map):
        output = dict()
        if isinstance(backbone_output, tuple):                       # fused inference path
            vfeats, geom = backbone_output
            bb, logits, offsets = self._plan.run(vfeats, geom, want_backbone=self.return_backbone_feats,
                                                 all_ones=not (self.use_coords or self.use_feats))
            output['backbone_feats'] = bb
            output['semantic_prediction_logits'] = logits
            output['offset_predictions'] = offsets
            return output
        from ..autograd import gather_rows
        cache = getattr(v2p_map, "_tl_cache", None)                  # lives with the batch's v2p map: the argsort the gather's backward needs
        if cache is None and v2p_map.is_cuda:
            cache = {}
            try:
                v2p_map._tl_cache = cache
            except AttributeError:
                pass
        backbone_feats = gather_rows(backbone_output.features, v2p_map, cache)
        if not (self.training and torch.is_grad_enabled() and backbone_feats.dtype in (torch.bfloat16, torch.float16) and os.environ.get("TL_HEAD_FP32") != "1"):
            backbone_feats = backbone_feats.float()
        # mixed-precision TRAINING keeps the heads in bf16 like the backbone (the reference's autocast runs their nn.Linear layers in
        # half precision too; get_loss casts logits / offsets to fp32): half the traffic of the gather, the two MLPs and their backward
        # over millions of points.  Inference / fp32 training: fp32 heads.
        output['backbone_feats'] = backbone_feats
        output['semantic_prediction_logits'] = self.semantic_linear(backbone_feats)
        output['offset_predictions'] = self.offset_linear(backbone_feats)
        return output

    @cuda_cast
    def get_loss(self, model_output, semantic_labels, offset_labels, masks_off, masks_sem, mask_rows=None, **kwargs):
        semantic_loss, offset_loss = point_wise_loss(
            model_output['semantic_prediction_logits'].float(), model_output['offset_predictions'].float(),
            masks_sem, masks_off, semantic_labels, offset_labels, mask_rows=mask_rows)
        loss_dict = dict(semantic_loss=semantic_loss * LOSS_MULTIPLIER_SEMANTIC, offset_loss=offset_loss)
        loss = sum(v for v in loss_dict.values())
        return loss, loss_dict
