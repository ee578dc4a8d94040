"""HBM-resident exact inner-product index (the MI355X replacement for the faiss flat index the
reference builds in image2layout/train/models/retrieval/retriever.py:79-84 and searches one
query at a time at :200-202)."""
from __future__ import annotations

import torch

from .. import _lib


def knn_topk_ip(index: torch.Tensor, queries: torch.Tensor, k: int, workspace: torch.Tensor | None = None):
    """index [N,D] fp32 cuda, queries [nq,D] fp32 cuda -> (scores [nq,k] fp32, idx [nq,k] int64),
    sorted by (score desc, index asc); exact (bit-identical to oracle/knn_oracle.c)."""
    assert index.is_cuda and queries.is_cuda, "knn_topk_ip runs on the GPU only (no CPU fallback)"
    assert index.dtype == torch.float32 and queries.dtype == torch.float32
    assert index.dim() == 2 and queries.dim() == 2 and index.shape[1] == queries.shape[1]
    index, queries = index.contiguous(), queries.contiguous()
    N, D = index.shape
    nq = queries.shape[0]
    L = _lib.lib()
    need = L.ralf_knn_topk_ip_workspace_bytes(N, D, nq, k)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=index.device)
    idx = torch.empty(nq, k, dtype=torch.int64, device=index.device)
    val = torch.empty(nq, k, dtype=torch.float32, device=index.device)
    rc = L.ralf_knn_topk_ip(_lib.ptr(index), N, D, _lib.ptr(queries), nq, k, _lib.ptr(idx), _lib.ptr(val),
                            _lib.ptr(workspace), workspace.numel(), _lib.stream_ptr())
    _lib.check(rc, "ralf_knn_topk_ip")
    return val, idx


def knn_scores(index: torch.Tensor, queries: torch.Tensor) -> torch.Tensor:
    N, D = index.shape
    nq = queries.shape[0]
    S = torch.empty(nq, N, dtype=torch.float32, device=index.device)
    rc = _lib.lib().ralf_knn_scores(_lib.ptr(index.contiguous()), N, D, _lib.ptr(queries.contiguous()), nq, _lib.ptr(S), _lib.stream_ptr())
    _lib.check(rc, "ralf_knn_scores")
    return S


def knn_select(scores: torch.Tensor, k: int):
    nq, N = scores.shape
    L = _lib.lib()
    need = L.ralf_knn_topk_ip_workspace_bytes(N, 4, nq, k)
    ws = torch.empty(need, dtype=torch.uint8, device=scores.device)
    idx = torch.empty(nq, k, dtype=torch.int64, device=scores.device)
    val = torch.empty(nq, k, dtype=torch.float32, device=scores.device)
    rc = L.ralf_knn_select(_lib.ptr(scores.contiguous()), N, nq, k, _lib.ptr(idx), _lib.ptr(val), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "ralf_knn_select")
    return val, idx


class FlatIPIndex:
    """Flat inner-product index kept in HBM; `search` mirrors faiss.IndexFlat.search(x, k) -> (D, I)."""

    def __init__(self, vectors: torch.Tensor, device: str = "cuda"):
        self.vectors = torch.as_tensor(vectors, dtype=torch.float32).to(device).contiguous()
        self._ws = None

    @property
    def ntotal(self) -> int:
        return self.vectors.shape[0]

    @property
    def d(self) -> int:
        return self.vectors.shape[1]

    def search(self, queries, k: int):
        q = torch.as_tensor(queries, dtype=torch.float32).to(self.vectors.device)
        if q.dim() == 1:
            q = q[None]
        need = _lib.lib().ralf_knn_topk_ip_workspace_bytes(self.ntotal, self.d, q.shape[0], k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.vectors.device)
        return knn_topk_ip(self.vectors, q, k, self._ws)
