"""kNN scan only (for rocprofv3 runs): 20 scans + 10 whole calls at nq = 16, 3 scans + 1 whole call at nq = 1024."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ralf_amd.retrieval.knn import knn_scores, knn_topk_ip
N, D = 61548, 1792
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(N, D, device="cuda", generator=g); X /= X.norm(dim=1, keepdim=True)
for nq in (16, 1024):
    Q = torch.randn(nq, D, device="cuda", generator=g)
    for _ in range(20 if nq == 16 else 3):   # (enough launches that the average is the steady-state duration)
        knn_scores(X, Q)
    for _ in range(10 if nq == 16 else 1):
        knn_topk_ip(X, Q, 16)
torch.cuda.synchronize()
