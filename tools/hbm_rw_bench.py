"""What a plain stream gets on this GPU (537 MB tensors, past the infinity cache): fill (write only), copy (read + write), sum (read only), the
library's own zero kernel and a residual-add (2 reads + 1 write).   python tools/hbm_rw_bench.py"""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from gemm_bench import timeit
from ralf_amd import ops
n = 1048576 * 256
a = torch.randn(n, device="cuda").bfloat16(); b = torch.empty_like(a); c = torch.randn(n, device="cuda").bfloat16()
nb = n * 2
for name, fn, byts in [("torch fill_ (write)", lambda: b.fill_(1.0), nb), ("torch copy_ (read + write)", lambda: b.copy_(a), 2 * nb),
                       ("torch sum (read)", lambda: a.sum(), nb), ("torch add (2 reads + write)", lambda: torch.add(a, c, out=b), 3 * nb),
                       ("ralf_zero (write)", lambda: ops.zero_(b.view(torch.float32)) if hasattr(ops, "zero_") else b.zero_(), nb)]:
    t = timeit(fn, iters=10)
    print(f"{name:30s} {t * 1e6:7.1f} us  {byts / t / 1e12:5.2f} TB/s")
