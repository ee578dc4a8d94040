"""3x3 convolutions of the bottlenecks (implicit GEMM), B = 64: forward plain / with the BatchNorm statistics epilogue, data gradient plain /
with the BatchNorm-backward reductions epilogue -- what the epilogues cost per layer.   python tools/conv_epi_bench.py"""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from gemm_bench import timeit
from ralf_amd import ops
dt = torch.bfloat16
for (H, C) in [(64, 64), (32, 128), (16, 256), (8, 512)]:
    Bn = 64
    x = torch.randn(Bn, H, H, C, device="cuda").to(dt)
    w = (torch.randn(C, 3, 3, C, device="cuda") * 0.05).to(dt)
    M = Bn * H * H
    geom = dict(RH=H, RW=H, SH=H, SW=H, SC=C, KH=3, KW=3, stride=1, pad=1, mode=0)
    geomd = dict(RH=H, RW=H, SH=H, SW=H, SC=C, KH=3, KW=3, stride=1, pad=1, mode=1)
    out = torch.empty(M, C, device="cuda", dtype=dt)
    cst = ops.colstats_buffer(M, C, x.device)
    mean = torch.zeros(C, device="cuda")
    mask = torch.randint(0, 255, (M, C // 8), device="cuda", dtype=torch.uint8)
    part = torch.empty((M + 63) // 64, 2, C, dtype=torch.float32, device="cuda")
    xin = torch.randn(M, C, device="cuda").to(dt)
    sk = 4 if H == 8 else 1
    t0 = timeit(lambda: ops.gemm(x, w, M, C, 9 * C, conv=geom, gather=1, out=out, splitk=sk))
    t1 = timeit(lambda: ops.gemm(x, w, M, C, 9 * C, conv=geom, gather=1, out=out, colstats=cst)) if sk == 1 else float("nan")
    t2 = timeit(lambda: ops.gemm(x, w, M, C, 9 * C, conv=geomd, gather=1, out=out, splitk=sk))
    t3 = timeit(lambda: ops.gemm(x, w, M, C, 9 * C, conv=geomd, gather=1, out=out, bnb=(xin, mask, mean, part))) if sk == 1 else float("nan")
    print(f"H={H:3d} C={C:4d}: forward {t0*1e6:6.1f} us, + statistics {t1*1e6:6.1f} us | data gradient {t2*1e6:6.1f} us, + BatchNorm-backward sums {t3*1e6:6.1f} us", flush=True)
