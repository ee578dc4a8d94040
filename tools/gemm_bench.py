"""Micro-benchmark of ralf_gemm on the shapes of the RALF train step (bf16)."""
import sys

import torch

sys.path.insert(0, ".")
import os
import ralf_amd._lib as _L
if os.environ.get("RALF_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["RALF_LIB"])   # A/B runs of two builds
from ralf_amd import ops  # noqa: E402


def timeit(fn, iters=30):
    """seconds per call, measured on a hipGraph replay of `iters` back-to-back launches (eager launches of kernels
    shorter than ~18 us measure the Python/ctypes call rate, not the kernel)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dt = torch.bfloat16
    print("== linear-style NT (x[M,K] @ W[N,K]^T)")
    for (M, N, K) in [(128, 128, 64), (3200, 256, 256), (3200, 1024, 256), (16384, 768, 256), (16384, 1024, 256), (16384, 256, 1024), (34048, 512, 256), (33792, 1024, 256),
                      (262144, 64, 64), (262144, 256, 64), (262144, 64, 256), (65536, 128, 512), (65536, 512, 128), (16384, 1024, 256), (4096, 2048, 512), (4096, 512, 2048), (8192, 8192, 8192)]:
        A, B = torch.randn(M, K, device="cuda").to(dt), torch.randn(N, K, device="cuda").to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        t = timeit(lambda: ops.gemm(A, B, M, N, K, out=out))
        print(f"NT  M={M:7d} N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TFLOP/s")
    print("== dgrad-style NN (dy[M,N] @ W[N,K])")
    for (M, N, K) in [(16384, 256, 1024), (16384, 1024, 256), (262144, 64, 256)]:
        A, B = torch.randn(M, K, device="cuda").to(dt), torch.randn(K, N, device="cuda").to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        t = timeit(lambda: ops.gemm(A, B, M, N, K, b_kcontig=False, out=out))
        print(f"NN  M={M:7d} N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TFLOP/s")
    print("== wgrad-style TN (dy[R,N]^T @ x[R,K]) split-K")
    for (R, N, K, sk) in [(16384, 1024, 256, 2), (16384, 256, 1024, 2), (262144, 64, 64, 64), (262144, 256, 64, 32), (3200, 256, 256, 1)]:
        A, B = torch.randn(R, N, device="cuda").to(dt), torch.randn(R, K, device="cuda").to(dt)
        out = torch.empty(N, K, device="cuda")
        t = timeit(lambda: ops.gemm(A, B, N, K, R, a_kcontig=False, b_kcontig=False, out=out, splitk=sk))
        print(f"TN  R={R:7d} N={N:5d} K={K:5d} sk={sk:3d}: {t*1e6:8.1f} us  {2*R*N*K/t/1e12:7.1f} TFLOP/s")
    print("== conv 3x3 implicit GEMM (B=64)")
    for (H, C, Co, s) in [(64, 64, 64, 1), (32, 128, 128, 1), (16, 256, 256, 1), (8, 512, 512, 1), (64, 128, 128, 2)]:
        Bn = 64
        OH = H // s
        x = torch.randn(Bn, H, H, C, device="cuda").to(dt)
        w = torch.randn(Co, 3, 3, C, device="cuda").to(dt)
        geom = dict(RH=OH, RW=OH, SH=H, SW=H, SC=C, KH=3, KW=3, stride=s, pad=1, mode=0)
        M = Bn * OH * OH
        out = torch.empty(M, Co, device="cuda", dtype=dt)
        t = timeit(lambda: ops.gemm(x, w, M, Co, 9 * C, conv=geom, gather=1, out=out))
        print(f"conv H={H:3d} C={C:4d} Co={Co:4d} s={s}: {t*1e6:8.1f} us  {2*M*Co*9*C/t/1e12:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
