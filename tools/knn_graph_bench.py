"""whole-call and scan-only time of the exact top-k on hipGraph replays (no host launch cost): python tools/knn_graph_bench.py [nq ...]"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from ralf_amd import _lib  # noqa: E402
from ralf_amd.retrieval.knn import knn_scores, knn_topk_ip  # noqa: E402


def graph_time(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e-3


nqs = [int(a) for a in sys.argv[1:]] or [1, 16, 32]
N, k = 61548, 16
for D in (256, 1792):
    gen = torch.Generator(device="cuda").manual_seed(0)
    X = torch.randn(N, D, device="cuda", generator=gen)
    for nq in nqs:
        Q = torch.randn(nq, D, device="cuda", generator=gen)
        ws = torch.empty(_lib.lib().ralf_knn_topk_ip_workspace_bytes(N, D, nq, k), dtype=torch.uint8, device="cuda")
        t_all = graph_time(lambda: knn_topk_ip(X, Q, k, ws))
        t_sc = graph_time(lambda: knn_scores(X, Q))
        by = N * D * 4 + nq * D * 4 + nq * k * 12
        print(f"V32={os.environ.get('RALF_KNN_V32', '0')} D={D:5d} nq={nq:4d}: call {t_all * 1e6:7.1f} us ({by / t_all / 8e12 * 100:5.1f}% of 8 TB/s)  scan {t_sc * 1e6:7.1f} us ({N * D * 4 / t_sc / 8e12 * 100:5.1f}%)", flush=True)
