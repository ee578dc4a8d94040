"""the stem convolution's forward and weight gradient alone (B = 64, 256 x 256 canvases):  [RALF_HIP_LIB=other.so] python tools/lab/stem_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ralf_amd import ops
from tools.gemm_bench import timeit
B, H = 64, 256
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.zeros(B, H, H, 8, device="cuda", dtype=torch.bfloat16)
x[..., :4] = torch.randn(B, H, H, 4, device="cuda", generator=g).to(torch.bfloat16)
w = torch.zeros(64, 7, 7, 8, device="cuda", dtype=torch.bfloat16)
w[..., :4] = (torch.randn(64, 7, 7, 4, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
y, part = ops.stem7x7_fwd(x, w)
dy = torch.randn(B, H // 2, H // 2, 64, device="cuda", generator=g).to(torch.bfloat16)
dw = ops.stem7x7_wgrad(x, dy)
torch.cuda.synchronize()
t1 = timeit(lambda: ops.stem7x7_fwd(x, w))
t2 = timeit(lambda: ops.stem7x7_wgrad(x, dy))
print("fwd %.1f us  wgrad (+ reduce) %.1f us   checks %.4f %.4f" % (t1 * 1e6, t2 * 1e6, float(y.double().abs().sum()), float(dw.double().abs().sum())))
