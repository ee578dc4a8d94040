"""Retrieval-table pipeline on the HBM-resident exact index.

Mirrors image2layout/train/models/retrieval/retriever.py:33-229 (`Retriever`, `preprocess_retrieval_cache`),
image2layout/train/helpers/retrieval_dataset_wrapper.py:17-148 (`load_cache_table`, `RetrievalDatasetWrapper`)
and retrieval/image.py:35-44 (`coarse_saliency`): same constructor arguments, same cache file names and the
same on-disk format (`torch.save` of a plain dict data_id -> list[db index]), so tables written here are read by
the reference's wrapper and vice versa.  Differences, all on purpose:
  * the scan is ONE batched ralf_knn_topk_ip call per split instead of one faiss call per query;
  * the image embedders (DreamSim / CLIP / VGG) are third-party models and stay outside: embeddings come from a
    `feature_fn(example) -> vector` callback or a precomputed array; only the 16x16 saliency feature is built in;
  * the dataset wrapper does not materialise the K retrieved IMAGES unless asked (they are unused when
    use_reference_image=False, yet the reference decodes and ships ~1 GB of them per step).
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .faiss_io import METRIC_INNER_PRODUCT, index_cache_path, read_flat_index, write_flat_index
from .knn import FlatIPIndex

LAYOUT_KEYS = ["center_x", "center_y", "width", "height", "label", "mask"]


def coarse_saliency(saliency: torch.Tensor, size=(16, 16)) -> np.ndarray:
    """[1,H,W] saliency in [0,1] -> 256-d vector in [-1,1] (nearest down-sampling, retrieval/image.py:35-44)."""
    h = F.interpolate(saliency.reshape(1, 1, *saliency.shape[-2:]).float(), size=size).flatten()
    return (2 * torch.clamp(h, 0.0, 1.0) - 1.0).numpy()


def _pad(values: Sequence, n: int):
    fill = False if isinstance(values[0], bool) else (0 if isinstance(values[0], int) else 0.0)
    return list(values) + [fill] * (n - len(values))


def table_path(dataset_name: str, split: str, backbone: str, top_k: int, cache_dir: str = "cache") -> str:
    return os.path.join(cache_dir, f"{dataset_name}_{split}_{backbone}_wo_head_table_between_dataset_indexes_top_k{top_k}.pt")


def _ids_of(dataset) -> list:
    """the `id` column without touching the other columns (an HF image dataset decodes every image on row access)"""
    if hasattr(dataset, "column_names") and "id" in dataset.column_names:
        return list(dataset["id"])
    return [dataset[i]["id"] for i in range(len(dataset))]


PRECOMPUTED_WEIGHT_DIR = "./cache/PRECOMPUTED_WEIGHT_DIR"   # image2layout/train/global_variables.py:21


class _TablePickle:
    """pickle-module shim for torch.load of retrieval tables: only the container classes such a table holds resolve
    (collections.defaultdict / OrderedDict and builtins); everything else is refused.  (torch's weights_only unpickler cannot
    rebuild a defaultdict at all: "Can only SETITEM for dict ...".)"""
    import pickle as _pk

    # (score tables hold one numpy array per sample: retriever.py:216-221 saves `scores[i]` as returned by faiss)
    _ALLOWED = {("collections", "defaultdict"), ("collections", "OrderedDict"), ("builtins", "list"), ("builtins", "dict"),
                ("builtins", "int"), ("builtins", "str"), ("builtins", "float"), ("builtins", "tuple"), ("builtins", "set"),
                ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("_codecs", "encode")}   # (protocol-2 pickles carry an array's bytes as a latin-1 string)

    class Unpickler(_pk.Unpickler):
        def find_class(self, module, name):
            if ("builtins" if module == "__builtin__" else module, name) in _TablePickle._ALLOWED:   # protocol-2 pickles spell it __builtin__
                return super().find_class(module, name)
            raise _TablePickle._pk.UnpicklingError(f"retrieval table: refusing to load {module}.{name}")

    load = staticmethod(lambda f, **kw: _TablePickle.Unpickler(f, **kw).load())
    __name__ = "pickle"


def load_cache_table(cache_path: str, top_k: int) -> dict:
    """helpers/retrieval_dataset_wrapper.py:17-33: the table at `cache_path`, else the precomputed one of the same file name under
    PRECOMPUTED_WEIGHT_DIR/retrieval_indexes.  The reference writes its tables as collections.defaultdict(list)
    (models/retrieval/retriever.py:188-221), which torch >= 2.6's default weights_only=True unpickler rejects; they are read
    through an allow-listing unpickler and returned as a plain dict."""
    if not os.path.exists(cache_path):
        alt = os.path.join(PRECOMPUTED_WEIGHT_DIR, "retrieval_indexes", os.path.basename(cache_path))
        if not os.path.exists(alt):
            raise ValueError(f"Cache not found in {alt}")
        cache_path = alt
    table = torch.load(cache_path, pickle_module=_TablePickle, weights_only=False)
    return {k: v[:top_k] for k, v in dict(table).items()}


class Retriever:
    def __init__(self, features=None, db_dataset=None, max_seq_length: int = 10, top_k: int = 1, dataset_name: str = "pku",
                 retrieval_backbone: str = "saliency", saliency_k=None, feature_fn: Optional[Callable] = None,
                 db_vectors: Optional[np.ndarray] = None, cache_dir: str = "cache", device: str = "cuda", **kwargs):
        self.features, self.db_dataset = features, db_dataset
        self.max_seq_length, self.top_k = max_seq_length, top_k
        self.dataset_name, self.retrieval_backbone = dataset_name, retrieval_backbone
        self.cache_dir, self.device = cache_dir, device
        self.index_name = "search_feat"
        if retrieval_backbone == "saliency" and feature_fn is None:
            feature_fn = lambda ex: coarse_saliency(torch.as_tensor(ex["saliency"]))  # noqa: E731
        self.feature_fn = feature_fn
        # the reference's embedding cache (retriever.py:65-88): read `{dataset}_{backbone}_wo_head_index.faiss` when it exists,
        # otherwise embed the split and write it, so either implementation can consume the other's cache directory
        cache_file = index_cache_path(dataset_name, retrieval_backbone, cache_dir)
        if db_vectors is None and kwargs.get("use_index_cache", True) and os.path.exists(cache_file):
            db_vectors, metric = read_flat_index(cache_file)
            if metric != METRIC_INNER_PRODUCT:
                raise ValueError(f"{cache_file}: metric_type {metric}, the retrieval index is inner product")
            if db_dataset is not None and len(db_dataset) != db_vectors.shape[0]:
                raise ValueError(f"{cache_file}: {db_vectors.shape[0]} vectors for a database of {len(db_dataset)}")
        elif db_vectors is None:
            assert feature_fn is not None, "pass feature_fn (image -> embedding) or db_vectors for non-saliency backbones"
            db_vectors = np.stack([np.asarray(feature_fn(db_dataset[i]), np.float32) for i in range(len(db_dataset))])
            if kwargs.get("use_index_cache", True) and kwargs.get("save_index_cache", True):
                os.makedirs(cache_dir, exist_ok=True)
                write_flat_index(cache_file, db_vectors)
        self.index = FlatIPIndex(np.ascontiguousarray(db_vectors, np.float32), device=device)
        self.table_paired_id_idx = {self._id(v): i for i, v in enumerate(_ids_of(db_dataset))}

    def _id(self, data_id):
        return int(data_id) if "pku" in self.dataset_name else data_id

    def search(self, queries: np.ndarray, k: int):
        scores, idx = self.index.search(np.ascontiguousarray(queries, np.float32), k)
        return scores.cpu().numpy(), idx.cpu().numpy()

    def preprocess_retrieval_cache(self, split: str, dataset, top_k: int, run_on_local: bool = True, save_scores: bool = False,
                                   queries: Optional[np.ndarray] = None) -> dict:
        """top-(k+1) search of every sample of `dataset` against the train-split index; on the train split the
        rank-0 hit (the sample itself) is dropped (retriever.py:211-213).  Writes the reference's table file."""
        if queries is None:
            queries = np.stack([np.asarray(self.feature_fn(dataset[i]), np.float32) for i in range(len(dataset))])
        scores, idx = self.search(queries, top_k + 1)
        lo = 1 if split == "train" else 0
        table, score_table = {}, {}
        for i, raw_id in enumerate(_ids_of(dataset)):
            data_id = self._id(raw_id)
            table[data_id] = [int(j) for j in idx[i, lo:]]
            score_table[data_id] = scores[i, lo:]
        os.makedirs(self.cache_dir, exist_ok=True)
        path = table_path(self.dataset_name, split, self.retrieval_backbone, top_k, self.cache_dir)
        torch.save(table, path)
        if save_scores:
            torch.save(score_table, path.replace("indexes", "scores"))
        return table


def merged_vectors(vector_sets: Sequence[np.ndarray], where_norm: str) -> np.ndarray:
    """concatenation of several backbones' embeddings (retriever.py:255-270 / :304-315): `where_norm` in
    {"before_concat", "after_concat"} divides by numpy's `LA.norm(x, ord=2)` of each part / of the concatenation --
    for a 2-D database block that is its SPECTRAL norm (a single scalar), for a 1-D query its length; kept as is."""
    assert where_norm in ("before_concat", "after_concat")
    parts = [np.asarray(v, np.float32) for v in vector_sets]
    if where_norm == "before_concat":
        parts = [v / np.linalg.norm(v, ord=2) for v in parts]
    out = np.concatenate(parts, axis=-1)
    if where_norm == "after_concat":
        out = out / np.linalg.norm(out, ord=2)
    return np.ascontiguousarray(out, np.float32)


def load_backbone_vectors(dataset_name: str, backbones: Sequence[str], cache_dir: str = "cache"):
    """the per-backbone embedding caches the reference merges (retriever.py:255-259: faiss.read_index + reconstruct_n)"""
    return [read_flat_index(index_cache_path(dataset_name, b, cache_dir))[0] for b in backbones]


def merge_retrieval_cache(dataset_name: str, split: str, backbones: Sequence[str], db_vector_sets, query_sets, data_ids, top_k: int,
                          where_norm: str, cache_dir: str = "cache", device: str = "cuda") -> dict:
    """`Retriever.preprocess_to_merge_retrieval_cache` (retriever.py:231-343): one index over the concatenated embeddings
    of several backbones; queries are normalised one by one (the reference's per-query 2-norm), searched in ONE batched
    scan, rank 0 dropped on the train split; written to the reference's file name."""
    name = "merge_" + "_".join(backbones)
    index = FlatIPIndex(merged_vectors(db_vector_sets, where_norm), device=device)
    n = len(data_ids)
    queries = np.stack([merged_vectors([qs[i] for qs in query_sets], where_norm) for i in range(n)])
    _, idx = index.search(queries, top_k + 1)
    idx = idx.cpu().numpy()
    lo = 1 if split == "train" else 0
    table = {(int(i) if "pku" in dataset_name else i): [int(j) for j in idx[r, lo:]] for r, i in enumerate(data_ids)}
    os.makedirs(cache_dir, exist_ok=True)
    torch.save(table, os.path.join(cache_dir, f"{dataset_name}_{split}_{name}_{where_norm}__topk{top_k}.pt"))
    return table


def cross_dataset_table(source: str, reference: str, split: str, backbone: str, reference_vectors: np.ndarray, queries: np.ndarray, source_ids,
                        top_k: int, save_scores: bool = False, cache_dir: str = "cache", device: str = "cuda") -> dict:
    """`CrossRetriever.preprocess_retrieval_cache` (cross_retriever.py:133-207): queries of one dataset against the other's
    index; keeps all top_k + 1 hits (nothing is dropped: a sample is never in the other dataset)."""
    index = FlatIPIndex(np.ascontiguousarray(reference_vectors, np.float32), device=device)
    scores, idx = index.search(np.ascontiguousarray(queries, np.float32), top_k + 1)
    scores, idx = scores.cpu().numpy(), idx.cpu().numpy()
    table = {(int(i) if source == "pku" else i): [int(j) for j in idx[r]] for r, i in enumerate(source_ids)}
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"source_{source}_reference_{reference}_{split}_{backbone}_cross_dataset_indexes_top_k{top_k}.pt")
    torch.save(table, path)
    if save_scores:
        torch.save({k: scores[r] for r, k in enumerate(table)}, path.replace("indexes", "scores"))
    return table


class RetrievalDatasetWrapper(torch.utils.data.Dataset):
    """table lookup -> K exemplar layouts per sample, padded to max_seq_length ([K, N] fields)."""

    def __init__(self, dataset_name: str, dataset, db_dataset, split: str, top_k: int, max_seq_length: int, retrieval_backbone: str,
                 random_retrieval: bool = False, saliency_k=None, num_cache_indexes_per_sample: int = 32, cache_dir: str = "cache",
                 with_images: bool = False, **_):
        self.dataset_name, self.dataset, self.db_dataset = dataset_name, dataset, db_dataset
        self.top_k, self.max_seq_length, self.with_images = top_k, max_seq_length, with_images
        # random_retrieval=True: the reference's train.py:136-137 picks RandomRetrievalDatasetWrapper (no table file is read, K uniform draws
        # per item).  Constructed through THIS class with the flag set, the behaviour is the same -- never a silent table lookup.
        self.random_retrieval = bool(random_retrieval) or isinstance(self, RandomRetrievalDatasetWrapper)
        self.table_idx = None if self.random_retrieval else load_cache_table(
            table_path(dataset_name, split, retrieval_backbone, num_cache_indexes_per_sample, cache_dir), top_k)
        self._layouts = self._layout_table(db_dataset, max_seq_length)

    @staticmethod
    def _layout_table(db_dataset, N: int) -> dict:
        """padded [n_db, N] arrays of the database layouts, built ONCE from the layout columns only: an item then costs one fancy
        index per field instead of K row fetches (which, on an image dataset, decode K images that are never used)"""
        cols = ["label", "center_x", "center_y", "width", "height"]
        src = db_dataset.select_columns(cols) if hasattr(db_dataset, "select_columns") else db_dataset
        n = len(src)
        out = {k: np.zeros((n, N), np.int64 if k == "label" else np.float32) for k in cols}
        out["mask"] = np.zeros((n, N), bool)
        for i in range(n):
            r = src[i]
            m = len(r["label"])
            assert m <= N, f"database layout {i} has {m} elements > max_seq_length {N}"
            for k in cols:
                out[k][i, :m] = np.asarray(r[k])
            out["mask"][i, :m] = True
        return out

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index: int) -> dict:
        data = dict(self.dataset[index])
        if self.random_retrieval:
            # helpers/random_retrieval_dataset_wrapper.py:75-77: K draws from torch's GLOBAL generator, upper bound = len(self) (the split's
            # own length, not the database's), used as database rows -- the same call, so a seeded run draws the same exemplars
            hits = torch.randint(low=0, high=len(self), size=[self.top_k]).tolist()
        else:
            data_id = int(data["id"]) if "pku" in self.dataset_name else data["id"]
            hits = self.table_idx[data_id]
        assert len(hits) == self.top_k, f"{len(hits)=} != {self.top_k=}"
        sel = np.asarray(hits, np.int64)
        retrieved = {"index": hits}
        for key in LAYOUT_KEYS:
            retrieved[key] = torch.from_numpy(self._layouts[key][sel])
        if self.with_images:
            rows = [self.db_dataset[j] for j in hits]
            for key in ("image", "saliency"):
                retrieved[key] = torch.stack([torch.as_tensor(r[key]) for r in rows])
        else:  # 1x1 placeholder keeps the 4-channel assertion of preprocess(); pixels are unused (use_reference_image=False)
            retrieved["image"] = torch.zeros(self.top_k, 4, 1, 1)
        data["retrieved"] = [retrieved]
        return data


class RandomRetrievalDatasetWrapper(RetrievalDatasetWrapper):
    """`helpers/random_retrieval_dataset_wrapper.py:12-112` (the `generator.random_retrieval=true` ablation, train/train.py:136-137): the K
    exemplars of an item are K uniform draws `torch.randint(0, len(self), [K])` from torch's global generator instead of a table lookup; no
    retrieval cache file is needed.  Same constructor keywords; `random_retrieval` is forced on as in the reference (its line 38)."""

