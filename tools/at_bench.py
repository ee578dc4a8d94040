"""A-operand transform (RalfGemmDesc.at_*) against the separate kernels it replaces, on the bottleneck shapes of layer1 / layer2 (B = 64, 256 x 256):
bit-identity of product, written-through operand and mask, and time per call (HIP events, back-to-back launches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ralf_amd import ops  # noqa: E402
from bench import _time_gpu  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
bf = torch.bfloat16


def fwd_case(name, M, K, N, res):
    """bn_apply(y, scale, shift (+ res), relu) -> 1x1 conv with column statistics   vs   the same product with at_mode 1"""
    y, W = rnd(M, K).to(bf), (rnd(N, K) * K ** -0.5).to(bf)
    r = rnd(M, K).to(bf) if res else None
    sc, sh = 0.5 + torch.rand(K, device=dev, generator=g), 0.3 * rnd(K)
    z, mask = torch.empty_like(y), torch.empty(M * K // 8, dtype=torch.uint8, device=dev)
    cst = ops.colstats_buffer(M, N, dev)

    def separate():
        ops._call("ralf_bn_apply", ops.dtype_code(y), ops._p(y), ops._p(sc), ops._p(sh), ops._p(r), ops._p(z), ops._p(mask), M, K, 1)
        return ops.gemm(z, W, M, N, K, colstats=cst)
    z2, mask2, cst2 = torch.empty_like(y), torch.zeros_like(mask), ops.colstats_buffer(M, N, dev)

    def fused():
        return ops.gemm(y, W, M, N, K, colstats=cst2, at=dict(mode=1, c1=sc, c2=sh, a2=r, out=z2, mask=mask2, relu=True))
    a, b = separate(), fused()
    torch.cuda.synchronize()
    same = torch.equal(a, b) and torch.equal(z, z2) and torch.equal(mask, mask2) and torch.equal(cst, cst2)
    ta = _time_gpu(separate, 20, 3)
    tb = _time_gpu(fused, 20, 3)
    tconv = _time_gpu(lambda: ops.gemm(z, W, M, N, K, colstats=cst), 20, 3)
    by = (M * K * 2 * (3 if res else 2) + M * N * 2) / 1e9
    print(f"{name:34s} M={M} K={K} N={N}: separate {ta * 1e6:7.1f} us (conv alone {tconv * 1e6:6.1f}), fused {tb * 1e6:7.1f} us = {by / tb / 1e3:.2f} TB/s of its own traffic; identical={same}")


def bwd_case(name, M, K, N, bnb):
    """bn_bwd_apply(dz, x -> dy) -> 1x1 data gradient (NN)   vs   the data gradient with at_mode 2"""
    dz, x, W = rnd(M, K).to(bf), rnd(M, K).to(bf), (rnd(K, N) * K ** -0.5).to(bf)
    c1, c2, c3 = 0.5 + torch.rand(K, device=dev, generator=g), 0.01 * rnd(K), 0.01 * rnd(K)
    dy = torch.empty_like(dz)
    extra = {}
    if bnb:
        bx, bm, mean = rnd(M, N).to(bf), torch.randint(0, 256, (M * N // 8,), dtype=torch.uint8, device=dev, generator=g), rnd(N)
        part, part2 = torch.empty((M + 63) // 64, 2, N, device=dev), torch.empty((M + 63) // 64, 2, N, device=dev)
    res = rnd(M, N).to(bf) if bnb else None

    def separate():
        ops._call("ralf_bn_bwd_apply_affine", ops.dtype_code(dz), ops._p(dz), ops._p(x), ops._p(c1), ops._p(c2), ops._p(c3), ops._p(dy), M, K)
        return ops.gemm(dy, W, M, N, K, b_kcontig=False, res=res, bnb=(bx, bm, mean, part) if bnb else None)
    dy2 = torch.empty_like(dz)

    def fused():
        return ops.gemm(dz, W, M, N, K, b_kcontig=False, res=res, bnb=(bx, bm, mean, part2) if bnb else None, at=dict(mode=2, c1=c1, c2=c2, c3=c3, a2=x, out=dy2))
    a, b = separate(), fused()
    torch.cuda.synchronize()
    same = torch.equal(a, b) and torch.equal(dy, dy2) and (not bnb or torch.equal(part, part2))
    ta, tb = _time_gpu(separate, 20, 3), _time_gpu(fused, 20, 3)
    print(f"{name:34s} M={M} K={K} N={N}: separate {ta * 1e6:7.1f} us, fused {tb * 1e6:7.1f} us; identical={same}")


fwd_case("layer1 conv1 <- bn3+res+relu", 262144, 256, 64, True)
fwd_case("layer2 conv1 <- bn3+res+relu", 65536, 512, 128, True)
fwd_case("layer2.0 conv1 <- layer1 bn3", 262144, 256, 128, True)
fwd_case("layer1 conv3 <- bn2+relu", 262144, 64, 256, False)
fwd_case("layer2 conv3 <- bn2+relu", 65536, 128, 512, False)
fwd_case("layer3 conv3 <- bn2+relu", 16384, 256, 1024, False)
bwd_case("layer1 conv3 dgrad <- bn3 bwd", 262144, 256, 64, True)
bwd_case("layer2 conv3 dgrad <- bn3 bwd", 65536, 512, 128, True)
bwd_case("layer1 conv1 dgrad <- bn1 bwd", 262144, 64, 256, True)
bwd_case("layer2 conv1 dgrad <- bn1 bwd", 65536, 128, 512, True)
bwd_case("layer3 conv1 dgrad <- bn1 bwd", 16384, 256, 1024, True)
