"""Oracle: sparse convolution arithmetic on CPU (torch fp32), two independent ways.

(i)  rulebook form  -- gather rows -> mm with W[tap] -> index_add  (scales to full tiles)
(ii) dense form     -- scatter into a dense grid, torch.nn.functional.conv3d /
                       conv_transpose3d, sample at the active sites (small tiles only)
(i) == (ii) is checked in tests/test_host_cpu.py (test_oracle_sparse_equals_dense); the identities are SURVEY.md Appendix B.

Weights are in the reference's state-dict layout [Cout, kx, ky, kz, Cin]
(tree_learn/util/train.py:70-72 comment; spconv SubMConv3d/SparseConv3d `.weight`).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _w_taps(weight):
    """[Cout,k,k,k,Cin] -> [k^3, Cin, Cout]  (tap-major, ready for x @ W[tap])."""
    co = weight.shape[0]; ci = weight.shape[-1]
    return weight.reshape(co, -1, ci).permute(1, 2, 0).contiguous()


def conv_table(feats, weight, table, n_out=None):
    """out[o] = sum_k W[k]^T . in[table[o,k]]   (table i32[n_out,K], -1 = absent).
    Serves SubMConv3d (nbr table), SparseConv3d k2s2 (child table)."""
    W = _w_taps(weight)
    table = torch.as_tensor(table).long()
    n_out = table.shape[0] if n_out is None else n_out
    out = torch.zeros(n_out, W.shape[2], dtype=feats.dtype)
    for k in range(table.shape[1]):
        idx = table[:, k]
        o = torch.nonzero(idx >= 0).squeeze(1)
        if len(o) == 0:
            continue
        out.index_add_(0, o, feats[idx[o]] @ W[k])
    return out


def inverse_conv(feats_coarse, weight, parent, fine_coords):
    """SparseInverseConv3d(k=2) paired with a k2s2 down conv: out[p] = W[tap(p)]^T . in[parent[p]],
    tap(p) = (x&1)*4 + (y&1)*2 + (z&1); rows whose parent is -1 get zeros (Appendix B)."""
    W = _w_taps(weight)                                   # [8, Cin_coarse, Cout_fine]
    parent = torch.as_tensor(parent).long()
    c = torch.as_tensor(fine_coords).long()
    tap = (c[:, 1] & 1) * 4 + (c[:, 2] & 1) * 2 + (c[:, 3] & 1)
    out = torch.zeros(len(parent), W.shape[2], dtype=feats_coarse.dtype)
    for k in range(8):
        o = torch.nonzero((tap == k) & (parent >= 0)).squeeze(1)
        if len(o):
            out[o] = feats_coarse[parent[o]] @ W[k]
    return out


# ---------------------------------------------------------------- dense forms (independent check)
def _dense(feats, coords, shape, batch_size):
    c = torch.as_tensor(coords).long()
    d = torch.zeros(batch_size, feats.shape[1], *[int(s) for s in shape], dtype=feats.dtype)
    d[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = feats
    return d


def _sample(d, coords):
    c = torch.as_tensor(coords).long()
    return d[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]]


def subm_conv_dense(feats, coords, weight, shape, batch_size):
    k = weight.shape[1]
    d = _dense(feats, coords, shape, batch_size)
    y = F.conv3d(d, weight.permute(0, 4, 1, 2, 3), padding=k // 2)
    return _sample(y, coords)


def down_conv_dense(feats, coords, weight, shape, batch_size, coarse_coords):
    d = _dense(feats, coords, shape, batch_size)
    y = F.conv3d(d, weight.permute(0, 4, 1, 2, 3), stride=2)
    return _sample(y, coarse_coords)


def inverse_conv_dense(feats_coarse, coarse_coords, weight, coarse_shape, batch_size, fine_coords, fine_shape):
    d = _dense(feats_coarse, coarse_coords, coarse_shape, batch_size)
    y = F.conv_transpose3d(d, weight.permute(4, 0, 1, 2, 3), stride=2)
    # conv_transpose output is 2*coarse_shape; fine coords beyond it (odd fine extents) have no parent
    c = torch.as_tensor(fine_coords).long()
    lim = torch.tensor(list(y.shape[2:]))
    ok = (c[:, 1:] < lim[None, :]).all(dim=1)
    out = torch.zeros(len(c), y.shape[1], dtype=feats_coarse.dtype)
    out[ok] = _sample(y, c[ok])
    return out
