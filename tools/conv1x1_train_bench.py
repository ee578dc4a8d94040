"""Output-heavy 1x1 convolutions of the train step at B = 64 (forward with the statistics epilogue; plain): us per launch and the rate of their
operand + output bytes.   [RALF_GEMM_TILE=11|22] python tools/conv1x1_train_bench.py"""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from gemm_bench import timeit
from ralf_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
B = 64
for (name, hw, K, N) in [("layer1 conv3 / downsample", 64, 64, 256), ("layer1 conv1 (first)", 64, 64, 64), ("layer1 conv1", 64, 256, 64), ("layer2 conv3", 32, 128, 512), ("layer2 conv1", 32, 512, 128),
                         ("layer3 conv3", 16, 256, 1024), ("layer4 conv3", 8, 512, 2048)]:
    M = B * hw * hw
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    cst = ops.colstats_buffer(M, N, x.device)
    t0 = timeit(lambda: ops.gemm(x, w, M, N, K, out=out), iters=20)
    t1 = timeit(lambda: ops.gemm(x, w, M, N, K, out=out, colstats=cst), iters=20)
    byts = 2 * (M * K + M * N + N * K)
    if ops.conv1x1_k64_ok(x, N, any_k=True):
        assert torch.equal(ops.conv1x1_k64(x, w), ops.gemm(x, w, M, N, K))
        t2 = timeit(lambda: ops.conv1x1_k64(x, w, out=out), iters=20)
        t3 = timeit(lambda: ops.conv1x1_k64(x, w, colstats=cst, out=out), iters=20)
        print(f"{'  conv1x1_k64':26s} {'':28s} plain {t2 * 1e6:6.1f} us ({byts / t2 / 1e12:4.2f} TB/s), + statistics {t3 * 1e6:6.1f} us ({byts / t3 / 1e12:4.2f} TB/s)")
    print(f"{name:26s} M={M:7d} K={K:4d} N={N:5d}: plain {t0 * 1e6:6.1f} us ({byts / t0 / 1e12:4.2f} TB/s), + statistics {t1 * 1e6:6.1f} us ({byts / t1 / 1e12:4.2f} TB/s)")
