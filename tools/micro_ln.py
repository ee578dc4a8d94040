"""micro-benchmarks of the row-wise kernels at the model's shapes: LayerNorm backward with / without the dgamma/dbeta atomics,
dropout, column sums, LayerNorm forward"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ralf_amd import ops  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dt = torch.bfloat16
for rows in (256, 3200, 16384, 34048):
    x = torch.randn(rows, 256, device="cuda").to(dt)
    dy = torch.randn(rows, 256, device="cuda").to(dt)
    sk = torch.randn(rows, 256, device="cuda").to(dt)
    g, b = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    dg, db = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
    seed = torch.zeros(1, dtype=torch.int64, device="cuda")
    t_f = timeit(lambda: ops.layernorm_fwd(x, g, b))
    t_b = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, need_wgrad=True, into=(dg, db), skip=sk))
    t_bn = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, need_wgrad=False, skip=sk))
    t_d = timeit(lambda: ops.dropout(dy, 0.1, seed, 3))
    t_c = timeit(lambda: ops.colsum(dy, rows, 256, out=dg))
    h = torch.randn(rows, 1024, device="cuda").to(dt)
    dg4 = torch.zeros(1024, device="cuda")
    t_c4 = timeit(lambda: ops.colsum(h, rows, 1024, out=dg4))
    mb = rows * 256 * 2 / 1e6
    print(f"rows {rows:6d}: ln_fwd {t_f:6.1f} us | ln_bwd(+wgrad) {t_b:6.1f} us  ln_bwd(no wgrad) {t_bn:6.1f} us  [4 x {mb:.1f} MB -> {4 * mb / t_bn / 1e3 if t_bn else 0:.2f} TB/s] | "
          f"dropout {t_d:6.1f} us | colsum256 {t_c:6.1f} us | colsum1024 {t_c4:6.1f} us", flush=True)
