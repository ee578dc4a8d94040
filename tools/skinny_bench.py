"""The few-row products of a decode step (B = 256 rows): us per launch on a graph replay.   [RALF_GEMM_SKINNY_SPLIT=0|1] python tools/skinny_bench.py"""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from gemm_bench import timeit
from ralf_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(256, 256, 256), (256, 1024, 256), (256, 256, 1024), (256, 518, 256), (256, 256, 512), (256, 512, 2048)]:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(A, W, M, N, K, bias=b, res=r, out=out), iters=50)
    print(f"{M} x {N} x {K}: {t * 1e6:5.1f} us")
