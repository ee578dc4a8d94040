"""a few launches of one implicit-GEMM 3x3 convolution (layer3 shape) and one short-K linear GEMM, for rocprofv3 --pmc runs."""
import sys, torch
sys.path.insert(0, ".")
from ralf_amd import ops
dt = torch.bfloat16
Bn, H, C, Co = 64, 16, 256, 256
x = torch.randn(Bn, H, H, C, device="cuda").to(dt)
w = torch.randn(Co, 3, 3, C, device="cuda").to(dt)
geom = dict(RH=H, RW=H, SH=H, SW=H, SC=C, KH=3, KW=3, stride=1, pad=1, mode=0)
M = Bn * H * H
out = torch.empty(M, Co, device="cuda", dtype=dt)
for _ in range(5):
    ops.gemm(x, w, M, Co, 9 * C, conv=geom, gather=1, out=out)
A, B = torch.randn(16384, 256, device="cuda").to(dt), torch.randn(1024, 256, device="cuda").to(dt)
o2 = torch.empty(16384, 1024, device="cuda", dtype=dt)
for _ in range(5):
    ops.gemm(A, B, 16384, 1024, 256, out=o2)
torch.cuda.synchronize()
