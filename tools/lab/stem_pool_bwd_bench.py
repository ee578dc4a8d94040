import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ralf_amd import ops
from tools.gemm_bench import timeit
B, H, W, C = 64, 128, 128, 64
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
y = torch.randn(B, H, W, C, device="cuda", generator=g).to(dt)
dpool = torch.randn(B, H // 2, W // 2, C, device="cuda", generator=g).to(dt)
arg = torch.randint(0, 9, (B, H // 2, W // 2, C), device="cuda", dtype=torch.int8, generator=g)
scale = torch.rand(C, device="cuda") + 0.5; shift = torch.randn(C, device="cuda") * 0.1; mean = torch.randn(C, device="cuda") * 0.1
c1 = torch.rand(C, device="cuda"); c2 = torch.randn(C, device="cuda") * 0.01; c3 = torch.randn(C, device="cuda") * 0.01
part = torch.empty(1024, 2, C, device="cuda"); dy = torch.empty_like(y)
f1 = lambda: ops._call("ralf_bn_relu_maxpool_bwd_reduce", 1, ops._p(dpool), ops._p(arg), ops._p(y), ops._p(scale), ops._p(shift), ops._p(mean), ops._p(part), 1024, B, H, W, C)
f2 = lambda: ops._call("ralf_bn_relu_maxpool_bwd_apply", 1, ops._p(dpool), ops._p(arg), ops._p(y), ops._p(scale), ops._p(shift), ops._p(c1), ops._p(c2), ops._p(c3), ops._p(dy), B, H, W, C)
f1(); f2(); torch.cuda.synchronize()
print("reduce %.1f us  apply %.1f us   checks %.6f %.6f" % (timeit(f1) * 1e6, timeit(f2) * 1e6, float(part.double().sum()), float(dy.double().sum())))
