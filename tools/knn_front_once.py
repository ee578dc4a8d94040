"""a few calls of the k-NN front end for a profiler:  python tools/knn_front_once.py [nq=64] [calls=30]
(FlatIPIndex.search on BASELINE config 4's index: two-stage from 40 queries, exhaustive below)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ralf_amd.retrieval.knn import FlatIPIndex  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, D, k = 61548, 1792, 16
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(N, D, device="cuda", generator=g); X /= X.norm(dim=1, keepdim=True)
Q = torch.randn(nq, D, device="cuda", generator=g); Q /= Q.norm(dim=1, keepdim=True)
index = FlatIPIndex(X)
for _ in range(3):
    index.search(Q, k)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(calls):
    v, i = index.search(Q, k)
torch.cuda.synchronize()
print(f"nq={nq}: {(time.perf_counter() - t0) / calls * 1e6:.1f} us per call (host clock, {calls} calls)")
