"""kNN call phases for rocprofv3 --kernel-trace: whole fused call (scan + in-kernel selection, merge), and the two phases of the
unfused form (scores, select) at nq in {1, 16, 32} on the 61548 x 1792 index and nq = 16 on 61548 x 256."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ralf_amd import _lib  # noqa: E402
from ralf_amd.retrieval.knn import knn_scores, knn_select, knn_topk_ip  # noqa: E402

N = 61548
for D, nqs in ((1792, (1, 16, 32, 1024)), (256, (16,))):
    g = torch.Generator(device="cuda").manual_seed(0)
    X = torch.randn(N, D, device="cuda", generator=g)
    X /= X.norm(dim=1, keepdim=True)
    for nq in nqs:
        Q = torch.randn(nq, D, device="cuda", generator=g)
        ws = torch.empty(_lib.lib().ralf_knn_topk_ip_workspace_bytes(N, D, nq, 16), dtype=torch.uint8, device="cuda")
        for _ in range(6):
            knn_topk_ip(X, Q, 16, ws)
        torch.cuda.synchronize()
        for _ in range(6):
            S = knn_scores(X, Q)
            knn_select(S, 16)
        torch.cuda.synchronize()
