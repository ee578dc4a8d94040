"""two-stage top-k (BASELINE config 4: 61548 x 1792, nq = 1024, k = 16): candidate pool size against time and certificate failures"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _time_gpu  # noqa: E402
from ralf_amd import ops  # noqa: E402
from ralf_amd.retrieval.knn import knn_rownorms, knn_topk_ip, knn_topk_ip_two_stage  # noqa: E402

N, D, k, nq = 61548, 1792, 16, 1024
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(N, D, device="cuda", generator=g); X /= X.norm(dim=1, keepdim=True)
Q = torch.randn(nq, D, device="cuda", generator=g); Q /= Q.norm(dim=1, keepdim=True)
Xb = ops.cast(X, torch.bfloat16)
_, xn = knn_rownorms(X, Xb, want_rows=False, want_max=True)
v_ref, i_ref = knn_topk_ip(X, Q, k)
for pool in (64, 48, 40, 32, 24, 20):
    v, i, nfb = knn_topk_ip_two_stage(X, Xb, Q, k, pool=pool, index_norms=xn, filtered=False)
    same = bool(torch.equal(i, i_ref) and torch.equal(v, v_ref))
    t = _time_gpu(lambda: knn_topk_ip_two_stage(X, Xb, Q, k, pool=pool, index_norms=xn, filtered=False), 10, 2)
    print(f"pool {pool:3d}: {t * 1e6:8.1f} us per call, {nfb:4d} of {nq} queries fell back to the exhaustive scan, identical to the exhaustive result: {same}")
