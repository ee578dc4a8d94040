"""1x1 convolutions of the inference backbone at B = 256 (eval-mode BatchNorm + ReLU (+ residual) in the epilogue): us per launch and the
HBM rate of their operand / output bytes.   python tools/conv1x1_infer_bench.py"""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from gemm_bench import timeit
from ralf_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
B = 256
for (name, hw, K, N, res) in [("layer1 conv1 (first)", 64, 64, 64, False), ("layer1 conv1", 64, 256, 64, False), ("layer1 conv3", 64, 64, 256, True), ("layer1 downsample", 64, 64, 256, False),
                              ("layer2 conv1", 32, 512, 128, False), ("layer2 conv3", 32, 128, 512, True), ("layer3 conv1", 16, 1024, 256, False), ("layer3 conv3", 16, 256, 1024, True),
                              ("layer4 conv1", 8, 2048, 512, False), ("layer4 conv3", 8, 512, 2048, True)]:
    M = B * hw * hw
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    sc, sh = torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).bfloat16() if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(x, w, M, N, K, bias=sh, colscale=sc, act="relu_post" if res else "relu", res=r, out=out), iters=10)
    byts = 2 * (M * K + M * N * (2 if res else 1) + N * K)
    if ops.conv1x1_k64_ok(x, N):
        t2 = timeit(lambda: ops.conv1x1_k64(x, w, scale=sc, shift=sh, res=r, relu=2 if res else 1, out=out), iters=10)
        print(f"{'  ralf_conv1x1_k64':22s} {'':34s} {t2 * 1e6:7.1f} us  {byts / t2 / 1e12:5.2f} TB/s")
    print(f"{name:22s} M={M:8d} K={K:5d} N={N:5d}: {t * 1e6:7.1f} us  {byts / t / 1e12:5.2f} TB/s  {2 * M * N * K / t / 1e12:6.1f} TFLOP/s")
