"""The BatchNorm apply / backward-apply passes at layer1's sizes (B = 64: [262144, 256] and [262144, 64] bf16) against torch's plain streams of
the same bytes.   python tools/bn_stream_bench.py"""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from gemm_bench import timeit
from ralf_amd import ops
from ralf_amd.ops import _call, _p, dtype_code
g = torch.Generator(device="cuda").manual_seed(0)
for (M, C) in [(262144, 256), (262144, 64), (65536, 512), (16384, 1024)]:
    x = torch.randn(M, C, device="cuda", generator=g).bfloat16()
    r = torch.randn(M, C, device="cuda", generator=g).bfloat16()
    dy = torch.randn(M, C, device="cuda", generator=g).bfloat16()
    y, dx = torch.empty_like(x), torch.empty_like(x)
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    mean, rstd, gam = torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5, torch.rand(C, device="cuda") + 0.5
    s1, s2 = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    mask = torch.empty(M * C // 8, dtype=torch.uint8, device="cuda")
    dt = dtype_code(x)
    nb = M * C * 2
    t_a = timeit(lambda: _call("ralf_bn_apply", dt, _p(x), _p(sc), _p(sh), _p(r), _p(y), _p(mask), M, C, 1), iters=20)
    t_a0 = timeit(lambda: _call("ralf_bn_apply", dt, _p(x), _p(sc), _p(sh), None, _p(y), _p(mask), M, C, 1), iters=20)
    t_add = timeit(lambda: torch.add(x, r, out=y), iters=20)
    t_cp = timeit(lambda: y.copy_(x), iters=20)
    print(f"[{M} x {C}] bn_apply + residual + mask {t_a * 1e6:6.1f} us ({(3 * nb + nb / 16) / t_a / 1e12:4.2f} TB/s); without residual {t_a0 * 1e6:6.1f} us ({(2 * nb + nb / 16) / t_a0 / 1e12:4.2f} TB/s) | "
          f"torch add {t_add * 1e6:6.1f} us ({3 * nb / t_add / 1e12:4.2f} TB/s), copy {t_cp * 1e6:6.1f} us ({2 * nb / t_cp / 1e12:4.2f} TB/s)")
