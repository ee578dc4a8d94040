"""stage by stage timing of the two-stage top-k search at BASELINE config 4 (61548 x 1792, nq = 1024, k = 16)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _time_gpu  # noqa: E402
from ralf_amd import ops  # noqa: E402
from ralf_amd.retrieval.knn import knn_rescore, knn_rownorms, knn_select, knn_select_cand, knn_topk_ip_two_stage  # noqa: E402

N, D, k, nq = 61548, 1792, 16, 1024
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(N, D, device="cuda", generator=g); X /= X.norm(dim=1, keepdim=True)
Q = torch.randn(nq, D, device="cuda", generator=g); Q /= Q.norm(dim=1, keepdim=True)
Xb = ops.cast(X, torch.bfloat16)
_, xn = knn_rownorms(X, Xb, want_rows=False, want_max=True)
pool = 64
qb = ops.cast(Q, torch.bfloat16)
qn, _ = knn_rownorms(Q, qb)
coarse = ops.gemm(qb, Xb, nq, N, D, out_dtype=torch.float32)
cval, cidx = knn_select(coarse, pool + 1)
exact = knn_rescore(X, Q, cidx)
t = {}
t["cast + row norms of the queries"] = _time_gpu(lambda: (ops.cast(Q, torch.bfloat16), knn_rownorms(Q, qb)), 10, 2)
t["coarse GEMM bf16 -> fp32 scores [1024 x 61548]"] = _time_gpu(lambda: ops.gemm(qb, Xb, nq, N, D, out_dtype=torch.float32), 10, 2)
t["select top-65 of the coarse scores"] = _time_gpu(lambda: knn_select(coarse, pool + 1), 10, 2)
t["exact re-score of 65 candidates per query"] = _time_gpu(lambda: knn_rescore(X, Q, cidx), 10, 2)
t["candidate select + certificate"] = _time_gpu(lambda: knn_select_cand(exact, cidx, k, bound=cval[:, pool], qnorms=qn, xnorms=xn, D=D), 10, 2)
t["whole call, dense coarse pass"] = _time_gpu(lambda: knn_topk_ip_two_stage(X, Xb, Q, k, index_norms=xn, filtered=False), 10, 2)
from ralf_amd.retrieval import knn as K  # noqa: E402
ns = K.FILTER_SAMPLE_ROWS
t["threshold pass: GEMM on 4096 rows + select"] = _time_gpu(lambda: knn_select(ops.gemm(qb, Xb[:ns], nq, ns, D, out_dtype=torch.float32), pool + 1), 10, 2)
sval, _ = knn_select(ops.gemm(qb, Xb[:ns], nq, ns, D, out_dtype=torch.float32), pool + 1)
th = sval[:, pool].contiguous()
tile = ops.gemm_filter_tile(nq, N, D)
T = (N + tile - 1) // tile
cnt = torch.empty(nq, T, dtype=torch.int32, device="cuda")
slots = K.FILTER_TILE_SLOTS * max(tile // 128, 1)
lst = torch.empty(nq, T, slots, 2, dtype=torch.int32, device="cuda")
t["filtered coarse GEMM (no score matrix)"] = _time_gpu(lambda: ops.gemm(qb, Xb, nq, N, D, flt=(th, cnt, lst)), 10, 2)
print("hits per query: mean %.0f max %d; per (query, tile): max %d of %d slots" % (cnt.sum(1).float().mean().item(), int(cnt.sum(1).max()), int(cnt.max()), slots))
t["whole call, filtered coarse pass"] = _time_gpu(lambda: knn_topk_ip_two_stage(X, Xb, Q, k, index_norms=xn, filtered=True), 10, 2)
v0, i0, _ = knn_topk_ip_two_stage(X, Xb, Q, k, index_norms=xn, filtered=False)
v1, i1, nfb = knn_topk_ip_two_stage(X, Xb, Q, k, index_norms=xn, filtered=True)
print("filtered == dense:", bool(torch.equal(i0, i1) and torch.equal(v0, v1)), "fallback queries:", nfb)
for name, v in t.items():
    print(f"{name:52s} {v * 1e6:8.1f} us")
