"""split-K sweep of the weight-gradient (TN) products on the current tile choice (set RALF_GEMM_TILE=11 / 22 to pin one)"""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from gemm_bench import timeit
from ralf_amd import ops
dt = torch.bfloat16
for (R, N, K) in [(16384, 1024, 256), (16384, 256, 1024), (262144, 256, 64), (262144, 64, 256), (65536, 512, 128), (65536, 128, 512), (4096, 2048, 512), (3200, 1024, 256), (34048, 256, 256)]:
    A, B = torch.randn(R, N, device="cuda").to(dt), torch.randn(R, K, device="cuda").to(dt)
    out = torch.empty(N, K, device="cuda")
    res = []
    for sk in (1, 2, 4, 8, 16, 32, 64, 128):
        if R // sk < 64:
            continue
        t = timeit(lambda: ops.gemm(A, B, N, K, R, a_kcontig=False, b_kcontig=False, out=out, splitk=sk))
        res.append(f"sk{sk}:{t*1e6:6.1f}")
    print(f"TN R={R:7d} N={N:5d} K={K:5d}  " + " ".join(res), flush=True)
