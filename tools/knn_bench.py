"""Micro-benchmark of the kNN scan (scores / select / whole call) across query-batch sizes."""
import sys
import time

import torch

sys.path.insert(0, ".")
from ralf_amd import _lib  # noqa: E402
from ralf_amd.retrieval.knn import knn_scores, knn_select, knn_topk_ip  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    N = 61548
    for D in (256, 1792):
        g = torch.Generator(device="cuda").manual_seed(0)
        X = torch.randn(N, D, device="cuda", generator=g)
        X /= X.norm(dim=1, keepdim=True)
        for nq in (1, 16, 32, 64, 128, 1024):
            Q = torch.randn(nq, D, device="cuda", generator=g)
            Q /= Q.norm(dim=1, keepdim=True)
            k = 16
            ws = torch.empty(_lib.lib().ralf_knn_topk_ip_workspace_bytes(N, D, nq, k), dtype=torch.uint8, device="cuda")
            t_all = timeit(lambda: knn_topk_ip(X, Q, k, ws))
            S = knn_scores(X, Q)
            t_sc = timeit(lambda: knn_scores(X, Q))
            t_sel = timeit(lambda: knn_select(S, k))
            by = N * D * 4 + nq * D * 4 + nq * k * 12
            fl = 2.0 * nq * N * D
            print(f"D={D:5d} nq={nq:5d}: total {t_all*1e6:9.1f} us  scores {t_sc*1e6:9.1f} us  select {t_sel*1e6:8.1f} us | "
                  f"QPS {nq/t_all:10.0f}  alg {by/t_all/1e9:8.1f} GB/s ({by/t_all/8e12*100:5.1f}% of 8 TB/s)  "
                  f"{fl/t_all/1e12:6.1f} TFLOP/s ({fl/t_all/157.3e12*100:5.1f}% of fp32 MFMA)", flush=True)


if __name__ == "__main__":
    main()
