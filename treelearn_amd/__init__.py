"""treelearn_amd: the MI355X-native per-tile sparse-conv segmentation path of TreeLearn."""
import os

# The tile loop keeps several forwards in flight on streams of their own and a lone forward runs the block-local unit builder on a side
# stream; the HIP runtime maps all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4) -- a fifth stream shares a queue
# with a compute stream and serialises behind it (measured: 7.37 -> 7.80 ms per tile with three tiles in flight).  Read when the HIP runtime
# initialises, i.e. effective if this package is imported before the first GPU call; an explicit setting of the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
