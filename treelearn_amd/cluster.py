"""Instance grouping on the GPU (reference tree_learn/util/pipeline.py:173-191)."""
import numpy as np
import torch

from . import _hip


def dbscan_min2(xy, eps, device="cuda"):
    """sklearn DBSCAN(eps, min_samples=2).fit(xy).labels_ on the HIP library: int64 labels, -1 = noise,
    clusters numbered by their smallest point index.  `xy`: float32 [n,2] numpy array or tensor."""
    L = _hip.lib()
    t = torch.as_tensor(np.ascontiguousarray(xy, dtype=np.float32)) if not torch.is_tensor(xy) else xy.float().contiguous()
    n = t.shape[0]
    if n == 0:
        return np.zeros(0, np.int64)
    t = t.to(device)
    labels = torch.empty(n, dtype=torch.int32, device=t.device)
    ncl = torch.empty(1, dtype=torch.int32, device=t.device)
    ws = torch.empty(int(L.tl_cluster_ws_bytes(n)), dtype=torch.uint8, device=t.device)
    _hip.check(L.tl_cluster_grid(_hip.ptr(t), n, float(eps), _hip.ptr(labels), _hip.ptr(ncl), _hip.ptr(ws), _hip.stream()), "tl_cluster_grid")
    return labels.cpu().numpy().astype(np.int64)


def hdbscan(xy, min_cluster_size):
    raise NotImplementedError("HDBSCAN grouping is not built yet on the HIP path (use grouping.use_hdbscan=False: "
                              "DBSCAN(eps=tau_group, min_samples=2)); there is deliberately no CPU fallback")
