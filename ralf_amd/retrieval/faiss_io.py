"""Reader / writer for faiss FLAT index files: the reference caches every embedding set as
`cache/{dataset}_{backbone}_wo_head_index.faiss` (retriever.py:65-88, written through datasets' `save_faiss_index`
= `faiss.write_index`) and re-reads them with `faiss.read_index(...).reconstruct_n(0, ntotal)` when it merges
backbones (retriever.py:255-259) or builds cross-dataset tables (cross_retriever.py:73-90).  faiss is a third-party
dependency absent from the reference tree (pyproject.toml: faiss-cpu ^1.7.4) and is NOT installed here, so this module
restates the published on-disk layout of a flat index (faiss/impl/index_write.cpp `write_index`, `write_index_header`,
WRITEXBVECTOR; faiss 1.7.x) -- little endian:

    4 B   fourcc  "IxFI" (inner product) | "IxF2" (L2) | "IxFl" (other metric)
    4 B   int32   d
    8 B   int64   ntotal
    8 B   int64   dummy (1 << 20)        x 2
    1 B   bool    is_trained
    4 B   int32   metric_type            0 = inner product, 1 = L2;   > 1: followed by float32 metric_arg
    8 B   uint64  number of float32 values (= ntotal * d; the code bytes / 4)
    ...   float32 [ntotal][d] row major

PARITY UNPINNED against a real faiss build (none available in this image): the tests check the byte layout above
literally, the round trip, and the error paths.  An index type other than flat (IVF, HNSW, ...) is refused loudly.
"""
from __future__ import annotations

import struct

import numpy as np

METRIC_INNER_PRODUCT, METRIC_L2 = 0, 1
_FOURCC = {b"IxFI": METRIC_INNER_PRODUCT, b"IxF2": METRIC_L2, b"IxFl": None}
_HEAD = struct.Struct("<4siqqq?i")


def read_flat_index(path: str):
    """-> (vectors float32 [ntotal, d], metric_type).  Raises ValueError on anything that is not a complete flat index."""
    with open(path, "rb") as f:
        head = f.read(_HEAD.size)
        if len(head) < _HEAD.size:
            raise ValueError(f"{path}: truncated faiss index header ({len(head)} bytes)")
        fourcc, d, ntotal, _d0, _d1, _trained, metric = _HEAD.unpack(head)
        if fourcc not in _FOURCC:
            raise ValueError(f"{path}: index type {fourcc!r} is not a flat index (only IxFI / IxF2 / IxFl are supported)")
        if _FOURCC[fourcc] is not None and _FOURCC[fourcc] != metric:
            raise ValueError(f"{path}: fourcc {fourcc!r} does not match metric_type {metric}")
        if metric > 1:
            f.read(4)   # metric_arg
        if d <= 0 or ntotal < 0:
            raise ValueError(f"{path}: bad shape d={d} ntotal={ntotal}")
        raw = f.read(8)
        if len(raw) < 8:
            raise ValueError(f"{path}: truncated before the vector block")
        (count,) = struct.unpack("<Q", raw)
        if count != ntotal * d:
            raise ValueError(f"{path}: {count} stored values for ntotal={ntotal} x d={d}")
        data = np.fromfile(f, dtype="<f4", count=count)
        if data.size != count:
            raise ValueError(f"{path}: truncated vector block ({data.size} of {count} values)")
    return np.ascontiguousarray(data.reshape(ntotal, d), np.float32), metric


def write_flat_index(path: str, vectors: np.ndarray, metric_type: int = METRIC_INNER_PRODUCT) -> None:
    x = np.ascontiguousarray(vectors, "<f4")
    if x.ndim != 2 or x.shape[1] == 0:
        raise ValueError("write_flat_index needs a non-empty [ntotal, d] matrix")
    fourcc = {METRIC_INNER_PRODUCT: b"IxFI", METRIC_L2: b"IxF2"}.get(metric_type)
    if fourcc is None:
        raise ValueError(f"metric_type {metric_type}: only inner product (0) and L2 (1) flat indexes are written")
    with open(path, "wb") as f:
        f.write(_HEAD.pack(fourcc, x.shape[1], x.shape[0], 1 << 20, 1 << 20, True, metric_type))
        f.write(struct.pack("<Q", x.size))
        x.tofile(f)


def index_cache_path(dataset_name: str, backbone: str, cache_dir: str = "cache") -> str:
    """retriever.py:65-67"""
    import os

    return os.path.join(cache_dir, f"{dataset_name}_{backbone}_wo_head_index.faiss")
