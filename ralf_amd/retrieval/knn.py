"""HBM-resident exact inner-product index (the MI355X replacement for the faiss flat index the
reference builds in image2layout/train/models/retrieval/retriever.py:79-84 and searches one
query at a time at :200-202)."""
from __future__ import annotations

import ctypes
import os

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


def knn_rescore(index: torch.Tensor, queries: torch.Tensor, cand: torch.Tensor) -> torch.Tensor:
    """exact fp32 scores of per-query candidate rows: out[q, j] = <queries[q], index[cand[q, j]]> (bit-identical to knn_scores)."""
    N, D = index.shape
    nq, pool = cand.shape
    out = torch.empty(nq, pool, dtype=torch.float32, device=index.device)
    rc = _lib.lib().ralf_knn_rescore(_lib.ptr(index), N, D, _lib.ptr(queries.contiguous()), nq, _lib.ptr(cand.contiguous()), pool, _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "ralf_knn_rescore")
    return out


def knn_rownorms(x: torch.Tensor, xb: torch.Tensor, want_rows: bool = True, want_max: bool = False):
    """per row {|x|, |xb|, |x - xb|} of an fp32 matrix and its bf16 copy (ralf_knn_rownorms) -> ([R,3] or None, [3] maxima or None)"""
    R, D = x.shape
    norms = torch.empty(R, 3, dtype=torch.float32, device=x.device) if want_rows else None
    mx = torch.zeros(3, dtype=torch.float32, device=x.device) if want_max else None
    rc = _lib.lib().ralf_knn_rownorms(_lib.ptr(x), _lib.ptr(xb), R, D, _lib.ptr(norms), _lib.ptr(mx), _lib.stream_ptr())
    _lib.check(rc, "ralf_knn_rownorms")
    return norms, mx


def knn_select_cand(exact: torch.Tensor, cand: torch.Tensor, k: int, bound=None, qnorms=None, xnorms=None, D: int = 0):
    """k best of the re-scored candidates per query by (score desc, row index asc); with bound/qnorms/xnorms also the
    per-query certificate flags (int32 [nq], 1 = not certified)."""
    nq, pool = exact.shape
    idx = torch.empty(nq, k, dtype=torch.int64, device=exact.device)
    val = torch.empty(nq, k, dtype=torch.float32, device=exact.device)
    bad = torch.empty(nq, dtype=torch.int32, device=exact.device) if bound is not None else None
    import ctypes
    bptr = ctypes.c_void_p(bound.data_ptr()) if bound is not None else None
    rc = _lib.lib().ralf_knn_select_cand(_lib.ptr(exact), _lib.ptr(cand), nq, pool, k, _lib.ptr(idx), _lib.ptr(val), bptr,
                                         bound.stride(0) if bound is not None else 0, _lib.ptr(qnorms), _lib.ptr(xnorms), D, _lib.ptr(bad), _lib.stream_ptr())
    _lib.check(rc, "ralf_knn_select_cand")
    return val, idx, bad


FILTER_SAMPLE_ROWS = 4096   # rows of the index the threshold pass ranks (expected list length: (pool + 1) * N / this)
FILTER_TILE_SLOTS = 16      # candidate slots per (query, 128 columns of a tile of the index): ~2 hits expected, P(> 16) ~ 1e-10 for unordered data


def knn_topk_ip_two_stage(index: torch.Tensor, index_bf16: torch.Tensor, queries: torch.Tensor, k: int, pool: int = 0, index_norms: torch.Tensor | None = None,
                          filtered: bool | None = None):
    """Exact top-k for LARGE query batches (nq >> 32, where the fp32 scan is bound by the 157 TFLOP/s fp32 matrix rate,
    SURVEY section 7): a bf16-MFMA coarse pass ranks every row, the best `pool` candidates per query are re-scored exactly
    in fp32 (same ascending-d accumulation chain as the exhaustive scan, so the scores are bit-identical), and the result is
    CERTIFIED per query.  With qb / xb the bf16 roundings,
        |coarse(q, x) - exact(q, x)|  <=  |q - qb| |x| + |qb| |x - xb|  (operand rounding)  +  2 D 2^-24 |q| |x|  (fp32 accumulation of
        both passes)  =: eps,   evaluated with the index-wide maxima of |x| and |x - xb| (index_norms, built once per index),
    so   exact k-th score  >=  coarse (pool+1)-th score + eps   proves that no row outside the pool can enter the top-k.
    Queries that fail the test are re-run through the exhaustive fp32 path, so the returned (scores, indices) always equal
    knn_topk_ip's.  Every step is a HIP kernel (cast, GEMM, select, gathered re-score, candidate select + certificate).
    Returns (scores, idx, n_fallback)."""
    from .. import ops

    assert index.is_cuda and index.dtype == torch.float32 and index_bf16.dtype == torch.bfloat16 and index_bf16.shape == index.shape
    N, D = index.shape
    nq = queries.shape[0]
    pool = pool or min(max(4 * k, 64), N - 1, 1023)
    assert k <= pool < N
    q = queries.contiguous()
    qb = ops.cast(q, torch.bfloat16)
    if index_norms is None:
        _, index_norms = knn_rownorms(index, index_bf16, want_rows=False, want_max=True)
    qn, _ = knn_rownorms(q, qb)
    cval = cidx = over = None
    if filtered is None:
        # measured at BASELINE config 4 (tools/knn_two_stage_probe.py, profiles/r04_knn_two_stage_stages.txt): the filtered product itself is
        # faster (305 vs 352 us: no 252 MB score matrix), but the threshold pass (44 us) and the slot unpack + selection cost what it saves --
        # 664 us against 652 for the dense form.  Kept for indexes whose score matrix would not fit; off by default.
        filtered = False
    if filtered:
        # the coarse scores never reach memory: a first pass over a slice of the index gives every query a LOWER bound of its (pool+1)-th
        # best coarse score (the (pool+1)-th best of a subset cannot exceed that of the whole), the pass over the whole index then keeps only
        # the scores at or above that bound (RalfGemmDesc.flt_*: ~pool * N / slice candidates per query instead of N scores: 8 MB instead of
        # the 252 MB score matrix written and re-read at BASELINE config 4)
        ns = min(FILTER_SAMPLE_ROWS, N)
        sval, _ = knn_select(ops.gemm(qb, index_bf16[:ns], nq, ns, D, out_dtype=torch.float32), pool + 1)
        thresh = sval[:, pool].contiguous()
        tile = ops.gemm_filter_tile(nq, N, D)
        T = (N + tile - 1) // tile
        cnt = torch.empty(nq, T, dtype=torch.int32, device=q.device)                  # hits per (query, column tile): every entry is written
        slots = FILTER_TILE_SLOTS * max(tile // 128, 1)
        lst = torch.empty(nq, T, slots, 2, dtype=torch.int32, device=q.device)
        ops.gemm(qb, index_bf16, nq, N, D, flt=(thresh, cnt, lst))
        W = T * slots
        rows = torch.empty(nq, W, dtype=torch.int64, device=q.device)                 # row ids / scores of the slots, -inf where unused
        scores = torch.empty(nq, W, dtype=torch.float32, device=q.device)
        over = torch.zeros(nq, dtype=torch.int32, device=q.device)                    # queries with a flooded tile: redone exhaustively below
        _lib.check(_lib.lib().ralf_knn_list_unpack(_lib.ptr(lst), _lib.ptr(cnt), nq, T, slots, _lib.ptr(rows), _lib.ptr(scores), _lib.ptr(over), _lib.stream_ptr()), "ralf_knn_list_unpack")
        cval, pos = knn_select(scores, pool + 1)
        cidx = torch.empty(nq, pool + 1, dtype=torch.int64, device=q.device)
        _lib.check(_lib.lib().ralf_knn_gather_rows(_lib.ptr(rows), W, _lib.ptr(pos), nq, pool + 1, _lib.ptr(cidx), _lib.stream_ptr()), "ralf_knn_gather_rows")
    else:
        coarse = ops.gemm(qb, index_bf16, nq, N, D, out_dtype=torch.float32)         # [nq, N] = Qb Xb^T (bf16 MFMA, fp32 accumulate)
        cval, cidx = knn_select(coarse, pool + 1)                                     # sorted by coarse score
    # candidates = the pool+1 best coarse rows; every row outside them has a coarse score <= cval[:, pool]
    exact = knn_rescore(index, q, cidx)                                              # same fp32 MFMA chain as the exhaustive scan
    val, idx, bad = knn_select_cand(exact, cidx, k, bound=cval[:, pool], qnorms=qn, xnorms=index_norms, D=D)
    if over is not None:
        bad = bad | over
    bad = torch.nonzero(bad).flatten()                                               # (index plumbing; syncs with the host)
    if bad.numel():   # not certified (tiny gaps / mass ties): exhaustive fp32 scan for those queries only
        v2, i2 = knn_topk_ip(index, q[bad].contiguous(), k)
        val[bad], idx[bad] = v2, i2
    return val, idx, int(bad.numel())


_BAD_HOST: dict = {}   # page-locked mirrors of the certificate flags, by batch size
_HOST_FLAGS = os.environ.get("RALF_KNN_HOST_FLAGS", "1") != "0"   # the search's last kernel writes the flags straight into them


def knn_topk_ip_two_stage_fused(index: torch.Tensor, index_bf16: torch.Tensor, queries: torch.Tensor, k: int, index_norms: torch.Tensor, pool: int = 0,
                                workspace: torch.Tensor | None = None, filtered: bool = False):
    """knn_topk_ip_two_stage through ONE call of the library (ralf_knn_topk_ip_two_stage: every launch of the search back to back from C into one
    workspace) + one read of the certificate flags; queries that are not certified go through the exhaustive scan.  Same results as
    knn_topk_ip, bit for bit.  filtered: ralf_knn_topk_ip_two_stage_filtered -- the coarse scores are thresholded in the product's epilogue instead
    of written as an [nq, N] matrix (pays from ~640 queries at BASELINE config 4: profiles/r06_knn_filtered_ab.txt; a query with a flooded column tile counts as not certified).
    Returns (scores, idx, n_fallback, workspace)."""
    N, D = index.shape
    nq = queries.shape[0]
    # pool + 1 candidates per query: a multiple of 16 (whole blocks of the re-score kernel: 64 candidates are four waves, 65 were five)
    pool = pool or min(max(4 * k, 64) - 1, N - 1, 1023)
    L = _lib.lib()
    need = L.ralf_knn_two_stage_workspace_bytes(N, D, nq, pool)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=index.device)
    q = queries.contiguous()
    idx = torch.empty(nq, k, dtype=torch.int64, device=index.device)
    val = torch.empty(nq, k, dtype=torch.float32, device=index.device)
    # the certificate flags, the one host read of the search: the last kernel writes them STRAIGHT into page-locked host memory (device-visible, coherent) -- a
    # device buffer + copy was one more launch on the call's dependent chain (an `any` kernel + its read: two more); RALF_KNN_HOST_FLAGS=0: the copy
    host = _BAD_HOST.get(nq)
    if host is None:
        host = _BAD_HOST[nq] = torch.empty(nq, dtype=torch.int32, pin_memory=True)
    bad = host if _HOST_FLAGS else torch.empty(nq, dtype=torch.int32, device=index.device)
    entry = L.ralf_knn_topk_ip_two_stage_filtered if filtered else L.ralf_knn_topk_ip_two_stage
    rc = entry(_lib.ptr(index), _lib.ptr(index_bf16), N, D, _lib.ptr(q), nq, k, pool, _lib.ptr(index_norms), _lib.ptr(idx), _lib.ptr(val),
               ctypes.c_void_p(bad.data_ptr()), _lib.ptr(workspace), workspace.numel(), _lib.stream_ptr())
    _lib.check(rc, "ralf_knn_topk_ip_two_stage_filtered" if filtered else "ralf_knn_topk_ip_two_stage")
    nbad = 0
    if not _HOST_FLAGS:
        host.copy_(bad, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    if bool(host.any()):   # (almost never)
        rows = torch.nonzero(host).flatten().to(index.device)
        nbad = int(rows.numel())
        v2, i2 = knn_topk_ip(index, q[rows].contiguous(), k)
        val[rows], idx[rows] = v2, i2
    return val, idx, nbad, workspace


class FlatIPIndex:
    """Flat inner-product index kept in HBM; `search` mirrors faiss.IndexFlat.search(x, k) -> (D, I)."""

    def __init__(self, vectors: torch.Tensor, device: str = "cuda", two_stage_min_queries: int = 40, filtered_min_queries: int = 640):
        self.vectors = torch.as_tensor(vectors, dtype=torch.float32).to(device).contiguous()
        self._ws = None
        self._ws2 = None
        self._bf16 = None                 # bf16 shadow of the index for the coarse pass of query batches (built on first use)
        self._norms = None                # {max|x|, max|xb|, max|x - xb|} of the index rows: the certificate's constants
        self.two_stage_min_queries = two_stage_min_queries
        # from this many queries the coarse pass thresholds its scores instead of writing them (0 = never).  Switched off for good by the first batch
        # that floods tiles (more than 1 in 16 queries not certified: an index ordered by similarity), so such an index pays the fallback once.
        self.filtered_min_queries = int(os.environ.get("RALF_KNN_FILTERED_MIN", filtered_min_queries))
        self.max_queries_per_call = 16384
        self.last_fallbacks = 0

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
        if q.shape[0] > self.max_queries_per_call:
            # a whole split as one query set (retriever.py: table building hands over every sample of the split): in blocks -- the search's workspace grows
            # with the batch (score matrix or slot lists: 4 GB per 16 384 queries at BASELINE config 4's index) and its launches take the query count as a
            # grid dimension (<= 65 535); per-query results do not depend on the batch they travel in
            outs, fallbacks = [], 0
            for i in range(0, q.shape[0], self.max_queries_per_call):
                outs.append(self.search(q[i:i + self.max_queries_per_call], k))
                fallbacks += self.last_fallbacks
            self.last_fallbacks = fallbacks
            return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        # Batches of >= two_stage_min_queries queries take the two-stage search (identical results).  The exhaustive fp32 scan streams the index
        # once at ~0.7 of the HBM rate up to 16 queries per pass (92 us at BASELINE config 4's index) and is bound by the fp32 matrix rate
        # beyond: 108 / 179 / 304 us at nq = 32 / 64 / 128 against ~100-120 for the two-stage search (profiles/r06_knn_route_sweep.txt).
        if self.two_stage_min_queries and q.shape[0] >= self.two_stage_min_queries and 4 * k < self.ntotal // 8 and self.d % 64 == 0:
            from .. import ops
            if self._bf16 is None:
                self._bf16 = ops.cast(self.vectors, torch.bfloat16)
                _, self._norms = knn_rownorms(self.vectors, self._bf16, want_rows=False, want_max=True)
            flt = bool(self.filtered_min_queries) and q.shape[0] >= self.filtered_min_queries
            val, idx, self.last_fallbacks, self._ws2 = knn_topk_ip_two_stage_fused(self.vectors, self._bf16, q.contiguous(), k, self._norms, workspace=self._ws2, filtered=flt)
            if flt and self.last_fallbacks * 16 > q.shape[0]:
                self.filtered_min_queries = 0
            return val, idx
        self.last_fallbacks = 0
        need = _lib.lib().ralf_knn_topk_ip_workspace_bytes(self.ntotal, self.d, q.shape[0], k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.vectors.device)
        return knn_topk_ip(self.vectors, q, k, self._ws)
