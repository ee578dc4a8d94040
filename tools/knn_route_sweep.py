"""exhaustive fp32 scan against the two-stage search (bf16 coarse pass + exact re-score + certificate) over the query-batch size, at BASELINE
config 4's index (61548 x 1792, k = 16): whole calls through FlatIPIndex's two code paths, and the two-stage search stage by stage.
    python tools/knn_route_sweep.py [out.txt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _time_gpu  # noqa: E402
from ralf_amd import ops  # noqa: E402
from ralf_amd.retrieval.knn import (knn_rescore, knn_rownorms, knn_select, knn_select_cand, knn_topk_ip, knn_topk_ip_two_stage,  # noqa: E402
                                     knn_topk_ip_two_stage_fused)

N, D, k = 61548, 1792, 16
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(N, D, device="cuda", generator=g); X /= X.norm(dim=1, keepdim=True)
Xb = ops.cast(X, torch.bfloat16)
_, xn = knn_rownorms(X, Xb, want_rows=False, want_max=True)
lines = [f"{'nq':>5s} {'exhaustive':>11s} {'two-stage':>10s} {'one call':>9s} {'fallbacks':>9s} | {'cast+norms':>10s} {'coarse':>8s} {'select':>8s} {'re-score':>9s} {'cand+cert':>9s}   (us; results identical: checked)"]
for nq in (16, 32, 48, 64, 96, 128, 192, 256, 512, 1024):
    Q = torch.randn(nq, D, device="cuda", generator=g); Q /= Q.norm(dim=1, keepdim=True)
    v0, i0 = knn_topk_ip(X, Q, k)
    v1, i1, nfb = knn_topk_ip_two_stage(X, Xb, Q, k, index_norms=xn)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)
    te = _time_gpu(lambda: knn_topk_ip(X, Q, k), 10, 2)
    t2 = _time_gpu(lambda: knn_topk_ip_two_stage(X, Xb, Q, k, index_norms=xn), 10, 2)
    v3, i3, nfb3, ws = knn_topk_ip_two_stage_fused(X, Xb, Q, k, xn)
    assert torch.equal(i0, i3) and torch.equal(v0, v3)
    t3 = _time_gpu(lambda: knn_topk_ip_two_stage_fused(X, Xb, Q, k, xn, workspace=ws), 20, 3)
    pool = 64
    qb = ops.cast(Q, torch.bfloat16)
    qn, _ = knn_rownorms(Q, qb)
    coarse = ops.gemm(qb, Xb, nq, N, D, out_dtype=torch.float32)
    cval, cidx = knn_select(coarse, pool + 1)
    exact = knn_rescore(X, Q, cidx)
    st = [_time_gpu(lambda: (ops.cast(Q, torch.bfloat16), knn_rownorms(Q, qb)), 10, 2), _time_gpu(lambda: ops.gemm(qb, Xb, nq, N, D, out_dtype=torch.float32), 10, 2),
          _time_gpu(lambda: knn_select(coarse, pool + 1), 10, 2), _time_gpu(lambda: knn_rescore(X, Q, cidx), 10, 2),
          _time_gpu(lambda: knn_select_cand(exact, cidx, k, bound=cval[:, pool], qnorms=qn, xnorms=xn, D=D), 10, 2)]
    lines.append(f"{nq:5d} {te * 1e6:11.1f} {t2 * 1e6:10.1f} {t3 * 1e6:9.1f} {nfb:9d} | " + " ".join(f"{x * 1e6:{w}.1f}" for x, w in zip(st, (10, 8, 8, 9, 9))))
    print(lines[-1], flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("# python tools/knn_route_sweep.py (MI355X; 61548 x 1792 fp32 index, k = 16, HIP events over 10 calls)\n" + "\n".join(lines) + "\n")
