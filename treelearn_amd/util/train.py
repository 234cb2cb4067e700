"""Host utilities the model path needs, mirroring reference tree_learn/util/train.py:
`cuda_cast` (:28-43), `point_wise_loss` (:145-166), `load_checkpoint` key handling (:65-102)."""
import functools

import torch
import torch.nn.functional as F


def cuda_cast(func):
    """Move every tensor argument to the current GPU (non_blocking: pinned batches overlap the copy)."""
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        mv = lambda x: x.cuda(non_blocking=True) if isinstance(x, torch.Tensor) and not x.is_cuda else x
        return func(*[mv(a) for a in args], **{k: mv(v) for k, v in kwargs.items()})
    return wrapper


@cuda_cast
def point_wise_loss(semantic_prediction_logits, offset_predictions, masks_sem, masks_off, semantic_labels, offset_labels, weights=None, mask_rows=None):
    return point_wise_loss_impl(semantic_prediction_logits, offset_predictions, masks_sem, masks_off, semantic_labels, offset_labels, weights, mask_rows)


def loss_mask_rows(masks_sem, masks_off):
    """Row indices of the two loss masks.  They depend on the batch only, so `TreeLearn.forward` takes them BEFORE the network is enqueued: the
    boolean indexing and the `mask.sum() == 0` tests of the reference's loss (util/train.py:147,153,159,163) each read a count back from the
    device, i.e. wait for the whole forward and then drain the queue four to six times; taken up front the reads find an idle device (or,
    for host-resident masks, no device at all) and the loss itself enqueues without a wait."""
    rows = []
    for m in (masks_sem, masks_off):
        idx = m.nonzero(as_tuple=False).view(-1)
        rows.append(idx if idx.is_cuda or not torch.cuda.is_available() else idx.cuda(non_blocking=True))
    return tuple(rows)


def point_wise_loss_impl(logits, offsets, masks_sem, masks_off, semantic_labels, offset_labels, weights=None, mask_rows=None):
    """Masked CE (sum / count) and masked mean L2 offset error; an empty mask yields `0 * sum`
    so the graph stays connected (train.py:147-148,159-160).  `mask_rows` = loss_mask_rows(masks_sem, masks_off): the masks as row lists
    (`x[mask]` == `x.index_select(0, rows)`, same order, same values) -- no read-back between the forward and the loss."""
    if mask_rows is None:
        mask_rows = loss_mask_rows(masks_sem, masks_off)
    sem_rows, off_rows = (r.to(logits.device) for r in mask_rows)
    n_sem = sem_rows.numel()
    if n_sem == 0:
        semantic_loss = 0 * logits.sum()
    else:
        ce = F.cross_entropy(logits.index_select(0, sem_rows), semantic_labels.index_select(0, sem_rows), reduction='sum' if weights is None else 'none')
        semantic_loss = (ce if weights is None else (ce * weights).sum()) / n_sem
    if off_rows.numel() == 0:
        offset_loss = 0 * offsets.sum()
    else:
        offset_loss = (offsets.index_select(0, off_rows) - offset_labels.index_select(0, off_rows)).pow(2).sum(1).sqrt().mean()
    return semantic_loss, offset_loss


def load_checkpoint(checkpoint, logger, model, optimizer=None, strict=False):
    """Load a reference `.pth` ({'net','optimizer','epoch'}): size-mismatched keys are dropped,
    `strict=False`, returns epoch + 1 (train.py:65-102)."""
    if hasattr(model, 'module'):
        model = model.module
    state = torch.load(checkpoint, map_location='cpu')
    src = state['net']
    tgt = model.state_dict()
    skipped = [k for k in src if k in tgt and src[k].size() != tgt[k].size()]
    for k in skipped:
        del src[k]
    missing, unexpected = model.load_state_dict(src, strict=strict)
    if logger is not None:
        if skipped:
            logger.info(f'removed keys in source state_dict due to size mismatch: {", ".join(skipped)}')
        if missing:
            logger.info(f'missing keys in source state_dict: {", ".join(missing)}')
        if unexpected:
            logger.info(f'unexpected key in source state_dict: {", ".join(unexpected)}')
    if optimizer is not None:
        assert 'optimizer' in state
        optimizer.load_state_dict(state['optimizer'])
    return state.get('epoch', 0) + 1
