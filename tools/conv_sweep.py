"""3x3 convolution (implicit GEMM) tile / split-K sweep; RALF_GEMM_TILE=11|22 pins the tile"""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from gemm_bench import timeit
from ralf_amd import ops
dt = torch.bfloat16
for (H, C, Co, s) in [(8, 512, 512, 1), (16, 256, 256, 1), (16, 512, 512, 2), (32, 128, 128, 1), (64, 64, 64, 1)]:
    Bn = 64
    OH = H // s
    x = torch.randn(Bn, H, H, C, device="cuda").to(dt)
    w = torch.randn(Co, 3, 3, C, device="cuda").to(dt)
    geom = dict(RH=OH, RW=OH, SH=H, SW=H, SC=C, KH=3, KW=3, stride=s, pad=1, mode=0)
    M = Bn * OH * OH
    out = torch.empty(M, Co, device="cuda", dtype=dt)
    res = []
    for sk in (1, 2, 3, 4, 6):
        t = timeit(lambda: ops.gemm(x, w, M, Co, 9 * C, conv=geom, gather=1, out=out, splitk=sk))
        res.append(f"sk{sk}:{t*1e6:6.1f}us")
    print(f"conv H={H:3d} C={C:4d} Co={Co:4d} s={s} (M={M}): " + " ".join(res) + f"   [{2*M*Co*9*C/1e9:.1f} GFLOP]", flush=True)
