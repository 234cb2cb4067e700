"""Backward of the sparse convolution (training step, reference tools/training/train.py:40).

dgrad needs no kernel of its own: with the rulebooks held in both directions it is the forward
gather-GEMM over the TRANSPOSED table with transposed weights --
    SubM      : gin[i] = sum_k W[K-1-k]^T gout[nbr[k][i]]        (nbr[k][o] = i  <=>  nbr[K-1-k][i] = o)
    down k2s2 : gin[i] = sum_k W[k]^T gout[inv[k][i]]
    inverse   : gin[q] = sum_k W[k]^T gout[child[k][q]]
    1x1       : gin    = gout . W
all through `tl_conv_fwd`.  wgrad (gW[k] = sum_o gout[o] (x) x[table[k][o]]) is, in this round, one gather of
all (row, tap) neighbour rows followed by a single library GEMM on the GPU (torch.mm -> rocBLAS/hipBLASLt):
[Cout x N] x [N x K*Cin].  A fused HIP wgrad (no materialised gather) is the next step.
"""
import torch

from . import ops


class TableRef:
    """A rulebook plus what dgrad needs: the table of the transposed conv and whether taps flip."""
    __slots__ = ("table", "n_out", "t_table", "n_in", "flip")

    def __init__(self, table, n_out, t_table, n_in, flip):
        self.table, self.n_out, self.t_table, self.n_in, self.flip = table, n_out, t_table, n_in, flip


def conv_backward(x, weight, ref: TableRef, grad_out, need_gx, need_gw):
    co, ci = weight.shape[0], weight.shape[-1]
    K = weight.numel() // (co * ci)
    gx = gw = None
    if need_gx:
        w = weight.detach().reshape(co, K, ci).permute(1, 2, 0)            # [K][Cin][Cout] = W[k]^T
        if ref.flip:
            w = w.flip(0)
        wt = w.contiguous().to(grad_out.dtype)                             # kernel layout [K]["Cout"=Cin]["Cin"=Cout]
        gx = ops.conv_fwd(grad_out, wt, ref.t_table, ref.n_in)
    if need_gw:
        g32 = grad_out.float()
        if ref.table is None:
            gw = (g32.t() @ x.float()).reshape(co, 1, ci)
        else:
            # one gather of every (row, tap) neighbour row ([N, K, Cin]; absent -> the appended zero row), then ONE
            # [Cout x N] x [N x K*Cin] GEMM: a tall reduction the library handles far better than K thin ones
            n = x.shape[0]
            xpad = torch.cat([x.float(), x.new_zeros((1, ci), dtype=torch.float32)], 0)
            idx = ref.table.t().long()                                         # [n_out, K]
            idx = torch.where(idx < 0, torch.full_like(idx, n), idx)
            xg = xpad[idx].reshape(idx.shape[0], K * ci)
            gw = (g32.t() @ xg).reshape(co, K, ci)
        gw = gw.reshape(weight.shape).to(weight.dtype)
    return gx, gw
