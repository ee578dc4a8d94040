"""data gradients of the backbone's six stride-2 convolutions (layer2.0 / 3.0 / 4.0: conv2 3x3, downsample 1x1) at B = 64, 256 x 256 canvases:
RALF_GEMM_PARITY=0 python tools/stride2_dgrad_bench.py   vs   python tools/stride2_dgrad_bench.py   (the parity-class form, gemm_impl.h GATHER 14)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ralf_amd import ops  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dt = torch.bfloat16
print("RALF_GEMM_PARITY =", os.environ.get("RALF_GEMM_PARITY", "1"))
for (H, Ci, Co, k, p) in [(64, 128, 128, 3, 1), (32, 256, 256, 3, 1), (16, 512, 512, 3, 1), (64, 256, 512, 1, 0), (32, 512, 1024, 1, 0), (16, 1024, 2048, 1, 0)]:
    B, OH = 64, H // 2
    Mi = B * H * H
    dy = torch.randn(B, OH, OH, Co, device="cuda").to(dt)
    w = (torch.randn(Ci, k, k, Co, device="cuda") * 0.05).to(dt)
    skip = torch.randn(Mi, Ci, device="cuda").to(dt)
    x = torch.randn(Mi, Ci, device="cuda").to(dt)
    mean = torch.randn(Ci, device="cuda")
    bits = torch.randint(0, 256, (Mi * Ci // 8,), device="cuda", dtype=torch.uint8)
    part = torch.empty((Mi + 63) // 64, 2, Ci, device="cuda")
    out = torch.empty(Mi, Ci, device="cuda", dtype=dt)
    geom = dict(RH=H, RW=H, SH=OH, SW=OH, SC=Co, KH=k, KW=k, stride=2, pad=p, mode=1)
    t0 = timeit(lambda: ops.gemm(dy, w, Mi, Ci, k * k * Co, conv=geom, gather=1, out=out))
    t1 = timeit(lambda: ops.gemm(dy, w, Mi, Ci, k * k * Co, conv=geom, gather=1, res=skip, bnb=(x, bits, mean, part), out=out))
    useful = 2.0 * Mi * Ci * k * k * Co / 4
    print(f"{k}x{k} s2  {Ci:5d} <- {Co:5d} at {H}x{H}: plain {t0 * 1e6:7.1f} us ({useful / t0 / 1e12:6.1f} useful TFLOP/s)   + skip + BatchNorm-backward epilogue {t1 * 1e6:7.1f} us")
