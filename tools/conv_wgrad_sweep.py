"""3x3 / 7x7 convolution weight gradients (dW = dy^T im2col(x), gather = 2): tile x split-K sweep.  RALF_GEMM_TILE=11|22 pins the tile."""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from gemm_bench import timeit
from ralf_amd import ops
dt = torch.bfloat16
Bn = 64
for (H, C, Co, s, k) in [(16, 256, 256, 1, 3), (64, 64, 64, 1, 3), (32, 128, 128, 1, 3), (8, 512, 512, 1, 3), (16, 512, 512, 2, 3), (256, 8, 64, 2, 7)]:
    p = k // 2
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn(Bn, H, H, C, device="cuda").to(dt)
    M = Bn * OH * OH
    dy = torch.randn(M, Co, device="cuda").to(dt)
    geom = dict(RH=OH, RW=OH, SH=H, SW=H, SC=C, KH=k, KW=k, stride=s, pad=p, mode=0)
    res = []
    for sk in (4, 8, 14, 28, 56, 112, 224):
        if M // sk < 64:
            continue
        t = timeit(lambda: ops.gemm(dy, x, Co, k * k * C, M, a_kcontig=False, b_kcontig=False, conv=geom, gather=2, out_dtype=torch.float32, splitk=sk))
        res.append(f"sk{sk}:{t*1e6:6.1f}")
    print(f"wgrad H={H:3d} C={C:4d} Co={Co:4d} s={s} k={k} (M={M}): " + " ".join(res) + f"   [{2*M*Co*k*k*C/1e9:.1f} GFLOP]", flush=True)
