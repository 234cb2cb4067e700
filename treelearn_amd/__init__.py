"""treelearn_amd: the MI355X-native per-tile sparse-conv segmentation path of TreeLearn."""
import os

# The tile loop keeps several forwards in flight on streams of their own and a lone forward runs the block-local unit builder on a side
# stream; the HIP runtime maps all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4) -- a fifth stream shares a queue
# with a compute stream and serialises behind it (measured: 7.37 -> 7.80 ms per tile with three tiles in flight).  Read when the HIP runtime
# initialises, i.e. effective if this package is imported before the first GPU call; an explicit setting of the caller wins.
import sys as _sys

_t = _sys.modules.get("torch")
# True when the HIP runtime was already up at import (the setting below then comes too late) and the caller had not chosen a queue count:
# the tile loop then keeps three tiles in flight instead of four (util/pipeline.py) -- with the default four hardware queues a fifth
# stream shares a queue with a compute stream
LATE_IMPORT = bool(_t is not None and getattr(_t, "cuda", None) is not None and _t.cuda.is_initialized() and "GPU_MAX_HW_QUEUES" not in os.environ)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def hw_queues():
    """Hardware queues the HIP runtime of this process maps its streams onto (GPU_MAX_HW_QUEUES as it stood when the runtime started)."""
    if LATE_IMPORT:
        return 4
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4
