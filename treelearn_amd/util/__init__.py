from .train import cuda_cast, point_wise_loss, load_checkpoint  # noqa: F401
from .pipeline import get_pointwise_preds, get_instances, group_dbscan, make_labels_consecutive  # noqa: F401
