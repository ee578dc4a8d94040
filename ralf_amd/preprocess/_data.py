"""Dataset access of the preprocess drivers.  Loading and transforming the PKU / CGL parquet splits is the reference's own code
(image2layout.train.data.get_dataset) and stays there: inside the reference's environment the drivers call it, exactly like
image2layout/preprocess/build_retrieval_indexes.py:56-64 does.  Outside of it, `--dataset_path` may point at a directory written by
`datasets.DatasetDict.save_to_disk` whose splits carry the layout columns (+ `saliency` for the saliency backbone)."""
from __future__ import annotations

import os


def load_splits(dataset_path: str, dataset_name: str, max_seq_length: int):
    """-> (dict split -> datasets.Dataset, features)"""
    data_dir = os.path.join(dataset_path, f"{dataset_name}{max_seq_length}" if dataset_name == "pku" else dataset_name)
    try:
        from image2layout.train.config import get_mock_train_cfg   # the reference's loader, when this runs inside its environment
        from image2layout.train.data import get_dataset
    except ImportError:
        import datasets as ds

        root = data_dir if os.path.isdir(data_dir) else dataset_path
        dd = ds.load_from_disk(root)
        return {k: dd[k] for k in dd.keys()}, dd[next(iter(dd.keys()))].features
    cfg = get_mock_train_cfg(max_seq_length, data_dir)
    return get_dataset(dataset_cfg=cfg.dataset, transforms=list(cfg.data.transforms), remove_column_names=["image_width", "image_height"])
