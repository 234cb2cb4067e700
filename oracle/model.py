"""Oracle: the whole TreeLearn forward on CPU, driven by a reference-layout state dict.

Restates (functionally, no nn.Module) reference
  tree_learn/model/tree_learn.py:83-103   forward_backbone / forward_head
  tree_learn/model/blocks.py:8-18         MLP
  tree_learn/model/blocks.py:29-39        Custom1x1Subm3d
  tree_learn/model/blocks.py:42-79        ResidualBlock
  tree_learn/model/blocks.py:81-149       UBlock
  tree_learn/util/train.py:145-166, tree_learn.py:106-126   loss
State-dict key names are the reference's, so any checkpoint the
reference can load drives this oracle unchanged.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import voxel, sparse_ops

BN_EPS = 1e-4           # tree_learn.py:34
LOSS_MULTIPLIER_SEMANTIC = 50   # tree_learn.py:9


def bn_relu(x, sd, prefix, training=False, relu=True):
    """BatchNorm1d(eps=1e-4) [+ ReLU] on features (spconv SparseSequential applies plain
    modules to .features -> normalises over all active voxels)."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if training:
        mean = x.mean(0); var = x.var(0, unbiased=False)
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    y = (x - mean) / torch.sqrt(var + BN_EPS) * w + b
    if relu and RELU_MASKS is not None and prefix in RELU_MASKS:
        # kink-pinned: the ReLU takes the branches another implementation took (train_step_grads(relu_masks=...)), so that autograd
        # differentiates the SAME piecewise-linear function even where a pre-activation lies within rounding of zero
        return y * RELU_MASKS[prefix].to(y.dtype)
    return torch.relu(y) if relu else y


RELU_MASKS = None       # {BatchNorm key prefix: bool [n, C]} while train_step_grads(relu_masks=...) runs


class Level:
    """Per-level geometry: coords + subm table (+ down tables to the next level)."""
    def __init__(self, coords, shape):
        self.coords = coords
        self.shape = np.asarray(shape, np.int64)
        self.nbr = voxel.rulebook_subm(coords, 3)
        self.parent = self.child = None


def build_levels(vcoords, spatial_shape, num_blocks):
    levels = [Level(vcoords, spatial_shape)]
    for _ in range(num_blocks - 1):
        cur = levels[-1]
        cc, parent, child, oshape = voxel.rulebook_down(cur.coords, cur.shape)
        cur.parent, cur.child = parent, child
        levels.append(Level(cc, oshape))
    return levels


def residual_block(x, sd, p, nbr, training):
    """blocks.py:72-79: out = conv_branch(x) + i_branch(x)."""
    key1x1 = p + ".i_branch.0.weight"
    if key1x1 in sd:                                              # Custom1x1Subm3d, blocks.py:29-39
        W = sd[key1x1]
        idt = x @ W.view(W.shape[0], W.shape[-1]).T
    else:
        idt = x
    y = bn_relu(x, sd, p + ".conv_branch.0", training)
    y = sparse_ops.conv_table(y, sd[p + ".conv_branch.2.weight"], nbr)
    y = bn_relu(y, sd, p + ".conv_branch.3", training)
    y = sparse_ops.conv_table(y, sd[p + ".conv_branch.5.weight"], nbr)
    return y + idt


def ublock(x, sd, p, levels, li, training, block_reps=2):
    """blocks.py:137-149."""
    L = levels[li]
    for i in range(block_reps):
        x = residual_block(x, sd, f"{p}.blocks.block{i}", L.nbr, training)
    if li + 1 < len(levels):
        identity = x
        d = bn_relu(x, sd, p + ".conv.0", training)
        d = sparse_ops.conv_table(d, sd[p + ".conv.2.weight"], L.child)
        d = ublock(d, sd, p + ".u", levels, li + 1, training, block_reps)
        d = bn_relu(d, sd, p + ".deconv.0", training)
        d = sparse_ops.inverse_conv(d, sd[p + ".deconv.2.weight"], L.parent, L.coords)
        x = torch.cat((identity, d), dim=1)
        for i in range(block_reps):
            x = residual_block(x, sd, f"{p}.blocks_tail.block{i}", L.nbr, training)
    return x


def mlp(x, sd, p, training):
    """blocks.py:8-18 with num_layers=2: Linear, BN, ReLU, Linear."""
    y = x @ sd[p + ".0.weight"].T + sd[p + ".0.bias"]
    y = bn_relu(y, sd, p + ".1", training)
    return y @ sd[p + ".3.weight"].T + sd[p + ".3.bias"]


def forward(sd, coords, input_feats, batch_ids, batch_size, voxel_size=0.1, num_blocks=7,
            use_coords=False, use_feats=False, spatial_shape=None, max_num_points_per_voxel=3,
            training=False, return_intermediates=False):
    """tree_learn.py:75-103.  `sd`: reference-layout state dict of CPU fp32 tensors."""
    vfeats, vcoords, v2p, sshape = voxel.voxelize(
        coords, input_feats, batch_ids, batch_size, voxel_size, use_coords, use_feats, max_num_points_per_voxel)
    if spatial_shape is not None:
        sshape = np.asarray(spatial_shape, np.int64)                  # tree_learn.py:86-87
    levels = build_levels(vcoords, sshape, num_blocks)
    x = torch.from_numpy(vfeats)
    x = sparse_ops.conv_table(x, sd["input_conv.0.weight"], levels[0].nbr)       # tree_learn.py:90
    x = ublock(x, sd, "unet", levels, 0, training)                               # :92
    x = bn_relu(x, sd, "output_layer.0", training)                               # :93
    bf = x[torch.from_numpy(v2p)]                                                # :99
    out = dict(backbone_feats=bf,
               semantic_prediction_logits=mlp(bf, sd, "semantic_linear", training),
               offset_predictions=mlp(bf, sd, "offset_linear", training))
    if return_intermediates:
        out["_levels"] = levels; out["_v2p"] = v2p; out["_voxel_feats"] = x
    return out


def point_wise_loss(logits, offsets, masks_sem, masks_off, semantic_labels, offset_labels):
    """tree_learn/util/train.py:145-166."""
    if masks_sem.sum() == 0:
        sem = 0 * logits.sum()
    else:
        sem = F.cross_entropy(logits[masks_sem], semantic_labels[masks_sem], reduction="sum") / int(masks_sem.sum())
    if masks_off.sum() == 0:
        off = 0 * offsets.sum()
    else:
        off = (offsets[masks_off] - offset_labels[masks_off]).pow(2).sum(1).sqrt().mean()
    return sem, off


def get_loss(model_output, semantic_labels, offset_labels, masks_off, masks_sem):
    """tree_learn.py:106-126."""
    sem, off = point_wise_loss(model_output["semantic_prediction_logits"].float(),
                               model_output["offset_predictions"].float(),
                               masks_sem, masks_off, semantic_labels, offset_labels)
    d = dict(semantic_loss=sem * LOSS_MULTIPLIER_SEMANTIC, offset_loss=off)
    return sum(d.values()), d


# ---------------------------------------------------------------- state-dict helpers
# The deterministic synthetic weights live with the other synthetic inputs (treelearn_amd/synth.py) so that
# bench.py's GPU process imports nothing from oracle/; re-exported here for the tests and the golden script.
from treelearn_amd.synth import random_state_dict, state_dict_manifest  # noqa: E402,F401


def train_step_grads(sd, batch, voxel_size, num_blocks, spatial_shape, dtype=torch.float64, relu_masks=None, use_coords=False, use_feats=False):
    """Loss and the gradient of every parameter for one training-mode forward (batch-statistics BatchNorm), by torch autograd
    through this restatement in `dtype` (float64: a round-off-free second opinion next to the reference-generated golden).
    Reference: the step body of tools/training/train.py:30-44 up to `.backward()`.  Returns (loss, {name: grad}).
    `relu_masks` ({BatchNorm key prefix: bool [rows, C]}, e.g. exported by the implementation under test): every ReLU behind such a
    BatchNorm multiplies by the given mask instead of deciding on its own sign test -- a ReLU within fp32 rounding of zero then no
    longer takes a different branch in float64 (one row of a 117-voxel level is 11 % of a channel's gradient)."""
    global RELU_MASKS
    RELU_MASKS = relu_masks
    try:
        return _train_step_grads(sd, batch, voxel_size, num_blocks, spatial_shape, dtype, use_coords, use_feats)
    finally:
        RELU_MASKS = None


def _train_step_grads(sd, batch, voxel_size, num_blocks, spatial_shape, dtype, use_coords=False, use_feats=False):
    p = {k: (v.detach().to(dtype).requires_grad_(True) if (v.is_floating_point() and not k.endswith(("running_mean", "running_var")))
             else (v.to(dtype) if v.is_floating_point() else v)) for k, v in sd.items()}
    T = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))                # noqa: E731
    vfeats, vcoords, v2p, sshape = voxel.voxelize(T(batch["coords"]).numpy(), T(batch["input_feats"]).numpy(), T(batch["batch_ids"]).numpy(),
                                                  int(batch["batch_size"]), voxel_size, use_coords, use_feats, 3)
    if spatial_shape is not None:
        sshape = np.asarray(spatial_shape, np.int64)
    levels = build_levels(vcoords, sshape, num_blocks)
    x = sparse_ops.conv_table(torch.from_numpy(vfeats).to(dtype), p["input_conv.0.weight"], levels[0].nbr)
    x = bn_relu(ublock(x, p, "unet", levels, 0, True), p, "output_layer.0", True)
    bf = x[torch.from_numpy(v2p)]
    sem, off = point_wise_loss(mlp(bf, p, "semantic_linear", True), mlp(bf, p, "offset_linear", True), T(batch["masks_sem"]), T(batch["masks_off"]),
                               T(batch["semantic_labels"]), T(batch["offset_labels"]).to(dtype))
    loss = sem * LOSS_MULTIPLIER_SEMANTIC + off
    loss.backward()
    return float(loss.detach()), {k: v.grad.detach() for k, v in p.items() if v.is_floating_point() and v.grad is not None}
