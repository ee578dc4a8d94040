"""Re-ranking of a retrieved pool (host NumPy, tiny): Maximal Marginal Relevance, top-k, random.
Same contracts as image2layout/train/models/retrieval/reranker.py:14-88 (argument names and meaning,
`score_type` in {"similarity", "distance"}, returned index order)."""
from __future__ import annotations

import numpy as np


def maximal_marginal_relevance(score_di_q: np.ndarray, score_di_dj: np.ndarray, lam: float, top_k: int, score_type: str) -> np.ndarray:
    """greedy MMR (Carbonell & Goldstein 1998): start from the best candidate, then repeatedly add the one
    maximising lam*relevance - (1-lam)*max-similarity-to-the-selected-set (argmin / min for distances)."""
    if score_type not in ("similarity", "distance"):
        raise ValueError("score_type must be one of ['similarity', 'distance'].")
    n = len(score_di_q)
    if not (0 < top_k <= n):
        raise ValueError(f"Number of iterations must be in (0, {n}].")
    if not (0 <= lam <= 1):
        raise ValueError("lambda must be in [0, 1].")
    sim = score_type == "similarity"
    chosen = [int(score_di_q.argmax() if sim else score_di_q.argmin())]
    for _ in range(top_k - 1):
        rest = np.setdiff1d(np.arange(n), chosen, assume_unique=True)   # ascending, like R[~isin(R, S)]
        pair = score_di_dj[np.asarray(chosen)[:, None], rest]
        crit = lam * score_di_q[rest] - (1 - lam) * (pair.max(axis=0) if sim else pair.min(axis=0))
        chosen.append(int(rest[np.argmax(crit) if sim else np.argmin(crit)]))
    return np.asarray(chosen)


def reranker_top_k(score_di_q: np.ndarray, top_k: int, score_type: str) -> np.ndarray:
    order = np.argsort(score_di_q)
    return (order[::-1] if score_type == "similarity" else order)[:top_k]


def reranker_random(input_len: int, top_k: int) -> np.ndarray:
    return np.random.choice(input_len, top_k, replace=False)
