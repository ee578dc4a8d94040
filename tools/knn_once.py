"""kNN scan only (for rocprofv3 --pmc runs): a few launches of the nq=16 and nq=1024 scans."""
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
    for _ in range(3):
        knn_scores(X, Q)
    knn_topk_ip(X, Q, 16)
torch.cuda.synchronize()
