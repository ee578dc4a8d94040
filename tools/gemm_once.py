"""a few launches of selected GEMM shapes (for rocprofv3 --pmc runs)."""
import sys, torch
sys.path.insert(0, ".")
from ralf_amd import ops
dt = torch.bfloat16
for (M, N, K) in [(16384, 1024, 256), (65536, 128, 1152)]:
    A, B = torch.randn(M, K, device="cuda").to(dt), torch.randn(N, K, device="cuda").to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    for _ in range(5):
        ops.gemm(A, B, M, N, K, out=out)
torch.cuda.synchronize()
