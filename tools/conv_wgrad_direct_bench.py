"""3x3 / stride-1 weight gradient: direct form (ralf_conv3x3_wgrad) against the implicit-GEMM form (ralf_gemm gather = 2 + split-K reduce + permute)
on the four bottleneck geometries at B = 64"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _time_gpu  # noqa: E402
from ralf_amd import ops  # noqa: E402
from ralf_amd.functional import _splitk_for  # noqa: E402

B = 64
g = torch.Generator(device="cuda").manual_seed(0)
for H, C in ((64, 64), (32, 128), (16, 256), (8, 512)):
    x = torch.randn(B, H, H, C, device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn(B, H, H, C, device="cuda", generator=g).to(torch.bfloat16)
    M = B * H * H
    geom = dict(RH=H, RW=H, SH=H, SW=H, SC=C, KH=3, KW=3, stride=1, pad=1, mode=0)
    out = torch.empty(C, C, 3, 3, device="cuda")

    def old():
        gg = ops.gemm(dy.view(M, C), x, C, 9 * C, M, a_kcontig=False, b_kcontig=False, conv=geom, gather=2, out_dtype=torch.float32, splitk=_splitk_for(C, 9 * C, M))
        return ops.permute4(gg, (C, C, 3, 3), (9 * C, 1, 3 * C, C), 3, torch.float32, out=out)

    def new():
        return ops.conv3x3_wgrad(dy, x, out=out)
    a = old().clone()
    b = new().clone()
    err = ((a - b).abs().max() / a.abs().max()).item()
    to, tn = _time_gpu(old, 20, 3), _time_gpu(new, 20, 3)
    fl = 2.0 * M * 9 * C * C
    print(f"H={H:3d} C={C:4d}: implicit-GEMM {to * 1e6:6.1f} us ({fl / to / 1e12:5.0f} TFLOP/s)   direct {tn * 1e6:6.1f} us ({fl / tn / 1e12:5.0f} TFLOP/s)   max rel diff {err:.1e}")
