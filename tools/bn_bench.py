import sys, torch
sys.path.insert(0, ".")
from ralf_amd import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for (M, C) in [(1048576, 64), (262144, 64), (262144, 256), (65536, 128), (65536, 512), (16384, 1024), (4096, 2048)]:
    x = torch.randn(M, C, device="cuda").bfloat16(); dy = torch.randn_like(x); y = torch.relu(x)
    s = torch.zeros(2, C, device="cuda"); mean = torch.zeros(C, device="cuda"); rstd = torch.ones(C, device="cuda"); g = torch.ones(C, device="cuda")
    L = ops._lib.lib(); P = ops._p
    t0 = timeit(lambda: ops._call("ralf_bn_stats", 1, P(x), P(s[0]), P(s[1]), M, C, P(ops.workspace(1024*2*C*4, x.device))))
    t1 = timeit(lambda: ops._call("ralf_bn_bwd_reduce", 1, P(x), P(dy), P(y), None, P(mean), P(rstd), P(s[0]), P(s[1]), M, C, 1, P(ops.workspace(1024*2*C*4, x.device))))
    out = torch.empty_like(x)
    t2 = timeit(lambda: ops._call("ralf_bn_apply", 1, P(x), P(g), P(mean), None, P(out), None, M, C, 1))
    t3 = timeit(lambda: ops._call("ralf_bn_bwd_apply", 1, P(x), P(dy), P(y), None, P(mean), P(rstd), P(g), P(s[0]), P(s[1]), P(out), None, M, C, 1))
    by = M * C * 2
    print(f"M={M:8d} C={C:5d}: stats {t0*1e6:7.1f}us {by/t0/1e12:5.2f}TB/s | bwd_reduce {t1*1e6:7.1f}us {3*by/t1/1e12:5.2f}TB/s | apply {t2*1e6:7.1f}us {2*by/t2/1e12:5.2f}TB/s | bwd_apply {t3*1e6:7.1f}us {4*by/t3/1e12:5.2f}TB/s")
