"""Embedding producers for the retrieval index and the MMR re-rank driver.

Replaces the one-item-per-call loops of the reference with batched device work:
  * `coarse_saliency_batch`  -- retrieval/image.py:35-44 + `extract_dataset_features` (:108-120) for the
    "saliency" backbone: the 16x16 feature of a whole split in one call (DreamSim / CLIP / VGG stay third-party);
  * `layout_features`        -- FIDNetV3.extract_features (fid/model.py:90-103) over all layouts of a split in
    large batches through the frozen layout-encoder kernels of ralf_amd/nn.py (the reference embeds K layouts
    per dataloader item, preprocess/rerank_indexes.py:117-120);
  * `rerank_tables`          -- preprocess/rerank_indexes.py:86-148: per sample, cosine similarity between the
    pooled exemplars' layout features (batched Gram matrices on the device, ralf_gemm) and greedy MMR / random
    re-rank on the host (ralf_amd/retrieval/reranker.py), written in the reference's table format.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from .reranker import maximal_marginal_relevance, reranker_random

LAYOUT_FIELDS = ("label", "mask", "center_x", "center_y", "width", "height")


def coarse_saliency_batch(saliency: torch.Tensor, size=(16, 16)) -> torch.Tensor:
    """[B,1,H,W] (or [B,H,W]) saliency in [0,1] -> [B, 256] features in [-1,1]; row b equals
    `coarse_saliency(saliency[b])` (nearest down-sampling is per image, so batching changes nothing)."""
    s = saliency if saliency.dim() == 4 else saliency.unsqueeze(1)
    h = F.interpolate(s.float(), size=size).flatten(1)   # index selection only: plumbing
    return 2 * torch.clamp(h, 0.0, 1.0) - 1.0


@torch.no_grad()
def layout_features(layout_encoder, fields: dict, rt, batch_rows: int = 16384, device: Optional[str] = None) -> torch.Tensor:
    """fields: the six layout tensors [M, N] (padded, `mask` True on real elements) of M layouts -> fp32 [M, 256].
    `layout_encoder` is ralf_amd.nn.LayoutEncoder (the model's frozen `layout_encoer`), `rt` its Runtime."""
    dev = torch.device(device) if device is not None else next(layout_encoder.parameters()).device
    M = fields["label"].shape[0]
    out = torch.empty(M, layout_encoder.d, dtype=torch.float32, device=dev)
    was_training, rt.training = rt.training, False   # index embeddings are deterministic (no dropout)
    try:
        for lo in range(0, M, batch_rows):
            chunk = {k: torch.as_tensor(fields[k][lo:lo + batch_rows]).to(dev) for k in LAYOUT_FIELDS}
            out[lo:lo + batch_rows] = layout_encoder.extract_features(chunk, rt.to(dev)).float()
    finally:
        rt.training = was_training
    return out


@torch.no_grad()
def pool_cosine(features: torch.Tensor, pools: torch.Tensor) -> np.ndarray:
    """features fp32 [M, d] (device), pools int64 [S, K] rows of `features` -> cosine similarity [S, K, K] (host)
    (nn.CosineSimilarity(dim=1, eps=1e-8) between every pair of a pool): Gram matrices by one batched ralf_gemm,
    the K x K normalisation on the host next to the MMR loop that consumes it."""
    from .. import ops

    S, K = pools.shape
    d = features.shape[1]
    f = features[pools.reshape(-1).to(features.device)].contiguous()      # gather: plumbing
    g = ops.gemm(f, f, K, K, d, batch=(S, 1), sA=(K * d, 0), sB=(K * d, 0))   # [1, S, K, K] = F F^T per pool
    g = g.view(S, K, K).cpu().numpy()
    n = np.maximum(np.sqrt(np.einsum("skk->sk", g)), 1e-8)
    return g / (n[:, :, None] * n[:, None, :])


def rerank_tables(table_indexes: dict, table_scores: Optional[dict], features: torch.Tensor, top_k: int, rerank_type: str = "mmr",
                  lam: float = 0.5, chunk: int = 4096) -> dict:
    """table_indexes: data_id -> list of K pooled db indices (K >= top_k); table_scores: data_id -> their query
    scores (needed for "mmr").  Returns data_id -> re-ranked top_k db indices (rerank_indexes.py:127-143)."""
    if rerank_type not in ("mmr", "random"):
        raise NotImplementedError(rerank_type)
    ids = list(table_indexes.keys())
    out = {}
    if rerank_type == "random":
        for i in ids:
            pool = np.asarray(table_indexes[i])
            out[i] = pool[reranker_random(len(pool), top_k)].tolist()
        return out
    K = len(table_indexes[ids[0]])
    for lo in range(0, len(ids), chunk):
        part = ids[lo:lo + chunk]
        pools = torch.tensor([list(table_indexes[i]) for i in part], dtype=torch.int64)
        cos = pool_cosine(features, pools)
        for j, i in enumerate(part):
            local = maximal_marginal_relevance(score_di_q=np.asarray(table_scores[i])[:K], score_di_dj=cos[j], top_k=top_k,
                                               score_type="similarity", lam=lam)
            out[i] = np.asarray(table_indexes[i])[local].tolist()
    return out
