"""one-call two-stage search at BASELINE config 4's index (61548 x 1792, k = 16): dense coarse pass (ralf_knn_topk_ip_two_stage) against the
threshold-filtered one (ralf_knn_topk_ip_two_stage_filtered), interleaved on one box, whole calls incl. the read of the certificate flags"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _time_gpu  # noqa: E402
from ralf_amd import ops  # noqa: E402
from ralf_amd.retrieval.knn import knn_rownorms, knn_topk_ip_two_stage_fused  # noqa: E402

N, D, k = 61548, 1792, 16
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(N, D, device="cuda", generator=g); X /= X.norm(dim=1, keepdim=True)
Xb = ops.cast(X, torch.bfloat16)
_, xn = knn_rownorms(X, Xb, want_rows=False, want_max=True)
for nq in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048]:
    Q = torch.randn(nq, D, device="cuda", generator=g); Q /= Q.norm(dim=1, keepdim=True)
    v0, i0, f0, ws = knn_topk_ip_two_stage_fused(X, Xb, Q, k, xn)
    v1, i1, f1, ws = knn_topk_ip_two_stage_fused(X, Xb, Q, k, xn, workspace=ws, filtered=True)
    same = bool(torch.equal(i0, i1) and torch.equal(v0, v1))
    td, tf = [], []
    for _ in range(4):
        td.append(_time_gpu(lambda: knn_topk_ip_two_stage_fused(X, Xb, Q, k, xn, workspace=ws), 10, 2))
        tf.append(_time_gpu(lambda: knn_topk_ip_two_stage_fused(X, Xb, Q, k, xn, workspace=ws, filtered=True), 10, 2))
    print(f"nq = {nq:5d}: dense {min(td) * 1e6:7.1f} us (fallbacks {f0})   filtered {min(tf) * 1e6:7.1f} us (fallbacks {f1})   identical results: {same}   "
          f"[all runs: dense {' '.join('%.0f' % (t * 1e6) for t in td)} | filtered {' '.join('%.0f' % (t * 1e6) for t in tf)}]")
