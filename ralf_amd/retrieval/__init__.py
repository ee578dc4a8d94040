from .knn import FlatIPIndex, knn_topk_ip  # noqa: F401
