"""GPU numerics: ralf_gemm (through the C ABI) vs a plain torch fp32 reference of the same product."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: dict(atol=2e-5, rtol=1e-5), torch.bfloat16: dict(atol=3e-2, rtol=2e-2)}


def rnd(*shape, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


def ref_mm(a, b):
    return a.float() @ b.float()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 77, 100), (33, 518, 256), (1000, 256, 1024), (5, 3, 8)])
def test_layouts(dtype, ak, bk, M, N, K):
    from ralf_amd import ops

    A = rnd(M, K, seed=1, dtype=dtype)   # asymmetric operands catch transposed fragments
    B = rnd(K, N, seed=2, dtype=dtype)
    ref = ref_mm(A, B)
    Ad = (A if ak else A.t().contiguous()).cuda()
    Bd = (B.t().contiguous() if bk else B).cuda()
    C = ops.gemm(Ad, Bd, M, N, K, a_kcontig=ak, b_kcontig=bk)
    scale = K ** 0.5
    torch.testing.assert_close(C.float().cpu() / scale, ref / scale, **TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_epilogues_and_splitk(dtype):
    from ralf_amd import ops

    M, N, K = 300, 200, 512
    A, W = rnd(M, K, seed=3, dtype=dtype), rnd(N, K, seed=4, dtype=dtype) * 0.05
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6, dtype=dtype)
    z = A.float() @ W.float().t() * 0.5 + bias
    for act, fn in [("relu", torch.relu), ("gelu", F.gelu), (None, lambda v: v)]:
        pre = torch.empty(M, N, dtype=dtype, device="cuda")
        C = ops.gemm(A.cuda(), W.cuda(), M, N, K, bias=bias.cuda(), act=act, res=res.cuda(), alpha=0.5, out2=pre)
        torch.testing.assert_close(C.float().cpu(), fn(z) + res.float(), **TOL[dtype])
        torch.testing.assert_close(pre.float().cpu(), z, **TOL[dtype])
    # activation-gradient masks
    y = torch.relu(z).to(dtype)
    C = ops.gemm(A.cuda(), W.cuda(), M, N, K, aux=y.cuda(), aux_mode="relu_mask", aux_scale=2.0)
    torch.testing.assert_close(C.float().cpu(), (A.float() @ W.float().t()) * (y.float() > 0) * 2.0, **TOL[dtype])
    C = ops.gemm(A.cuda(), W.cuda(), M, N, K, aux=z.to(dtype).cuda(), aux_mode="gelu_grad")
    zz = z.to(dtype).float().requires_grad_(True)
    F.gelu(zz).sum().backward()
    torch.testing.assert_close(C.float().cpu(), (A.float() @ W.float().t()) * zz.grad, **TOL[dtype])
    # split-K (weight-gradient shape): dW[N,K'] = dY^T X, fp32 output, accumulate
    Mr = 4096
    dY, X = rnd(Mr, 96, seed=7, dtype=dtype), rnd(Mr, 160, seed=8, dtype=dtype)
    out = torch.ones(96, 160, device="cuda")
    ops.gemm(dY.cuda(), X.cuda(), 96, 160, Mr, a_kcontig=False, b_kcontig=False, out=out, splitk=8, accumulate=True)
    torch.testing.assert_close(out.cpu() / 64, (dY.float().t() @ X.float() + 1) / 64, **TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_strided(dtype):
    """two-level batch (b, h) with strided heads, as attention-style products use it."""
    from ralf_amd import ops

    Bn, H, S, dh = 3, 4, 50, 32
    q, k = rnd(Bn, S, H * dh, seed=9, dtype=dtype), rnd(Bn, S, H * dh, seed=10, dtype=dtype)
    ref = torch.einsum("bshd,bthd->bhst", q.float().view(Bn, S, H, dh), k.float().view(Bn, S, H, dh))
    out = ops.gemm(q.cuda(), k.cuda(), S, S, dh, lda=H * dh, ldb=H * dh, batch=(H, Bn), sA=(dh, S * H * dh), sB=(dh, S * H * dh))
    torch.testing.assert_close(out.float().cpu(), ref, **TOL[dtype])


CONVS = [  # (B, H, W, Cin, Cout, k, stride, pad)
    (2, 16, 12, 8, 64, 7, 2, 3), (2, 9, 7, 16, 24, 3, 1, 1), (2, 10, 8, 16, 32, 3, 2, 1), (3, 6, 5, 32, 16, 1, 1, 0), (2, 8, 8, 16, 40, 1, 2, 0),
    # channel counts that are multiples of the k-tile (one tap per k-tile: the scalar-tap loaders) on odd, non-square grids
    (2, 11, 9, 64, 128, 3, 1, 1), (2, 13, 10, 64, 64, 3, 2, 1), (1, 15, 15, 128, 64, 1, 2, 0), (3, 5, 23, 64, 72, 3, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_gather(dtype, cfg):
    """implicit-im2col forward / data-gradient / weight-gradient vs torch conv2d autograd (NCHW fp32)."""
    from ralf_amd import ops

    Bn, H, W, Ci, Co, k, s, p = cfg
    x = rnd(Bn, Ci, H, W, seed=11, dtype=dtype).float().requires_grad_(True)
    w = (rnd(Co, Ci, k, k, seed=12, dtype=dtype) * 0.1).float().requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    OH, OW = y.shape[2:]
    gy = rnd(*y.shape, seed=13, dtype=dtype).float()
    y.backward(gy)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(dtype).cuda()           # NHWC
    w_ohwi = w.detach().permute(0, 2, 3, 1).contiguous().to(dtype).cuda()        # [Co][kh][kw][ci]
    w_dgrad = w.detach().permute(1, 2, 3, 0).contiguous().to(dtype).cuda()       # [Ci][kh][kw][co]
    gyn = gy.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    Mo, Mi, Kf = Bn * OH * OW, Bn * H * W, k * k * Ci
    geom_f = dict(RH=OH, RW=OW, SH=H, SW=W, SC=Ci, KH=k, KW=k, stride=s, pad=p, mode=0)
    yo = ops.gemm(xn, w_ohwi, Mo, Co, Kf, conv=geom_f, gather=1)
    torch.testing.assert_close(yo.float().cpu().view(Bn, OH, OW, Co).permute(0, 3, 1, 2), y.detach(), **TOL[dtype])
    geom_d = dict(RH=H, RW=W, SH=OH, SW=OW, SC=Co, KH=k, KW=k, stride=s, pad=p, mode=1)
    gx = ops.gemm(gyn, w_dgrad, Mi, Ci, k * k * Co, conv=geom_d, gather=1)
    torch.testing.assert_close(gx.float().cpu().view(Bn, H, W, Ci).permute(0, 3, 1, 2), x.grad, **TOL[dtype])
    gw = ops.gemm(gyn, xn, Co, Kf, Mo, a_kcontig=False, b_kcontig=False, conv=geom_f, gather=2, splitk=3, out_dtype=torch.float32)
    tol = dict(atol=0.15, rtol=3e-2) if dtype == torch.bfloat16 else dict(atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(gw.cpu().view(Co, k, k, Ci).permute(0, 3, 1, 2), w.grad, **tol)


@pytest.mark.parametrize("cfg", [(8, 64, 64, 128, 128, 3, 1), (8, 64, 64, 128, 256, 1, 0), (16, 32, 32, 256, 128, 3, 1)])
@pytest.mark.parametrize("bnb", [False, True])
def test_stride2_data_gradient_by_parity_classes(cfg, bnb):
    """the data gradient of a STRIDE-2 convolution at sizes where ralf_gemm takes the parity-class form (gemm_impl.h GATHER 14: rows visited class by
    class, only the class's taps multiplied, none at all for the odd pixels of a 1 x 1 convolution) against torch's conv2d autograd -- plain with
    a skip gradient, and with the BatchNorm-backward epilogue (ReLU mask, dz stored, per-64-row partial sums: any row order gives the same column
    sums).  layer2.0 / layer3.0 / layer4.0 of the backbone: conv2 (3 x 3) and downsample (1 x 1), common/image.py:39-48."""
    from ralf_amd import ops

    dtype = torch.bfloat16
    Bn, H, W, Ci, Co, k, p = cfg
    x = rnd(Bn, Ci, H, W, seed=11, dtype=dtype).float().requires_grad_(True)
    w = (rnd(Co, Ci, k, k, seed=12, dtype=dtype) * 0.05).float().requires_grad_(True)
    y = F.conv2d(x, w, None, 2, p)
    OH, OW = y.shape[2:]
    gy = rnd(*y.shape, seed=13, dtype=dtype).float()
    y.backward(gy)
    w_dgrad = w.detach().permute(1, 2, 3, 0).contiguous().to(dtype).cuda()       # [Ci][kh][kw][co]
    gyn = gy.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    Mi = Bn * H * W
    assert (Mi // 4) % 128 == 0 and (Mi // 128) * ((Ci + 127) // 128) >= 192       # the parity form's conditions (whole tiles per class, the 128 x 128 tile rule)
    geom_d = dict(RH=H, RW=W, SH=OH, SW=OW, SC=Co, KH=k, KW=k, stride=2, pad=p, mode=1)
    skip = rnd(Mi, Ci, seed=14, dtype=dtype).cuda()
    want = x.grad.permute(0, 2, 3, 1).reshape(Mi, Ci) + skip.float().cpu()
    if not bnb:
        gx = ops.gemm(gyn, w_dgrad, Mi, Ci, k * k * Co, conv=geom_d, gather=1, res=skip)
        torch.testing.assert_close(gx.float().cpu(), want, **TOL[dtype])
        return
    xa = rnd(Mi, Ci, seed=15, dtype=dtype).cuda()
    mean = rnd(Ci, seed=16).cuda() * 0.3
    keep = torch.rand(Mi, Ci, generator=torch.Generator().manual_seed(17)) > 0.4
    bits = (keep.view(-1, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8).cuda()
    part = torch.full(((Mi + 63) // 64, 2, Ci), float("nan"), device="cuda")
    dz = ops.gemm(gyn, w_dgrad, Mi, Ci, k * k * Co, conv=geom_d, gather=1, res=skip, bnb=(xa, bits, mean, part))
    torch.testing.assert_close(dz.float().cpu(), (want * keep).to(dtype).float(), **TOL[dtype])
    assert bool((dz[~keep.cuda()] == 0).all())
    d = dz.float()
    torch.testing.assert_close(part[:, 0].sum(0), d.sum(0), atol=2e-2, rtol=1e-3)                       # (block order differs from the row order: the TOTALS are what the statistics use)
    torch.testing.assert_close(part[:, 1].sum(0), (d * (xa.float() - mean)).sum(0), atol=5e-2, rtol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (1000, 64, 128), (4101, 128, 2048), (70000, 256, 64), (130, 192, 64)])
def test_column_statistics_epilogue(dtype, M, N, K):
    """RalfGemmDesc.colstats: per-64-row column sums / sums of squares of the output AS STORED (BatchNorm batch statistics
    out of the convolution epilogue); ragged last block, both tile shapes, thousands of partial rows -> bn_stats_from_partials."""
    from ralf_amd import ops

    A, B = rnd(M, K, seed=31, dtype=dtype).cuda(), rnd(N, K, seed=32, dtype=dtype).cuda()
    st = torch.full(((M + 63) // 64, 2, N), float("nan"), device="cuda")
    out = ops.gemm(A, B, M, N, K, colstats=st)
    tol = dict(atol=1e-5 * K ** 0.5 * 2, rtol=1e-4) if dtype == torch.float32 else TOL[dtype]   # fp32: summation order over long K
    torch.testing.assert_close(out.float().cpu(), ref_mm(A.cpu(), B.cpu().t()), **tol)
    o = out.float()
    pad = (-M) % 64
    o64 = torch.cat([o, torch.zeros(pad, N, device="cuda")]).view(-1, 64, N)
    torch.testing.assert_close(st[:, 0], o64.sum(1), atol=1e-3 * K ** 0.5, rtol=1e-4)
    torch.testing.assert_close(st[:, 1], (o64 * o64).sum(1), atol=1e-2 * K, rtol=1e-4)
    # the BatchNorm entry that consumes them == statistics computed from the tensor itself
    g, b = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    res = torch.empty(4, N, device="cuda")
    ws = torch.empty(256 * 2 * N, device="cuda")
    cnt = torch.zeros((), dtype=torch.long, device="cuda")
    ops._call("ralf_bn_stats_from_partials", ops._p(st), st.shape[0], ops._p(g), ops._p(b), None, None, ops._p(cnt),
              ops._p(res[0]), ops._p(res[1]), ops._p(res[2]), ops._p(res[3]), M, N, 1e-5, 0.1, ops._p(ws))
    torch.testing.assert_close(res[0], o.mean(0), atol=1e-3, rtol=1e-3)
    torch.testing.assert_close(res[1], (o.var(0, unbiased=False) + 1e-5).rsqrt(), atol=1e-3, rtol=2e-3)
    assert int(cnt) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,relu", [(1000, 128, 192, True), (4101, 64, 64, True), (300, 256, 512, False)])
def test_batchnorm_backward_statistics_epilogue(dtype, M, N, K, relu):
    """RalfGemmDesc.bnb_*: dz = mask * (dy @ W + skip) stored in the operand dtype, and per 64-row block the column sums of dz and of
    dz * (x - mean); then ralf_bn_bwd_stats_from_partials == the sums ralf_bn_bwd_reduce computes from the tensors (ragged last block,
    both tile shapes via N)."""
    from ralf_amd import ops

    A, W = rnd(M, K, seed=41, dtype=dtype).cuda(), (rnd(K, N, seed=42, dtype=dtype) * 0.1).to(dtype).cuda()
    skip, x = rnd(M, N, seed=43, dtype=dtype).cuda(), rnd(M, N, seed=44, dtype=dtype).cuda()
    mean, rstd = rnd(N, seed=45).cuda() * 0.3, rnd(N, seed=46).abs().cuda() + 0.5
    keep = torch.rand(M, N, generator=torch.Generator().manual_seed(47)) > 0.4
    bits = (keep.view(-1, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8).cuda() if relu else None
    part = torch.full(((M + 63) // 64, 2, N), float("nan"), device="cuda")
    dz = ops.gemm(A, W, M, N, K, b_kcontig=False, res=skip, bnb=(x, bits, mean, part))
    want = A.float() @ W.float() + skip.float()
    if relu:
        want = want * keep.cuda()
    torch.testing.assert_close(dz.float().cpu(), want.to(dtype).float().cpu(), **TOL[dtype])
    assert not relu or bool((dz[~keep.cuda()] == 0).all())
    d = dz.float()
    pad = (-M) % 64
    z64 = lambda t: torch.cat([t, torch.zeros(pad, N, device="cuda")]).view(-1, 64, N)   # noqa: E731
    torch.testing.assert_close(part[:, 0], z64(d).sum(1), atol=1e-3, rtol=1e-4)
    torch.testing.assert_close(part[:, 1], z64(d * (x.float() - mean)).sum(1), atol=2e-3, rtol=1e-4)
    s = torch.ones(2, N, device="cuda")
    ops._call("ralf_bn_bwd_stats_from_partials", ops._p(part), part.shape[0], ops._p(rstd), ops._p(s[0]), ops._p(s[1]), N, ops._p(torch.empty(128 * 2 * N, device="cuda")), None, None, 0, None)
    torch.testing.assert_close(s[0] - 1, d.sum(0), atol=2e-3, rtol=1e-4)
    torch.testing.assert_close(s[1] - 1, (d * (x.float() - mean) * rstd).sum(0), atol=5e-3, rtol=2e-4)


@pytest.mark.parametrize("tile", ["11", "22"])
def test_every_case_on_a_forced_tile(tile):
    """the tile heuristic picks 64x64 (4 waves) or 128x128 (8 waves) by shape; RALF_GEMM_TILE pins one of them (read once per
    process), so the layout / epilogue / gather / column-statistics cases above are re-run in a child process on each."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, RALF_GEMM_TILE=tile)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k", "not forced_tile", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_grouped_weight_gradients_match_per_layer_products():
    """ralf_wgrad_grouped / ralf_colsum_grouped: the weight and bias gradients of several linear layers in one launch each (tiles that
    walk their whole reduction, and the slab + grouped-reduce form for long reductions) against fp32 matmuls of the same operands"""
    from ralf_amd import ops

    g = torch.Generator(device="cuda").manual_seed(7)
    shapes = [(16384, 768, 256, 1), (16384, 256, 1024, 1), (3200, 1024, 256, 1), (34048, 512, 256, 3), (256, 256, 256, 1), (1024, 128, 384, 2)]
    jobs, refs, bjobs, brefs = [], [], [], []
    flat = torch.randn(sum(n * k for _, n, k, _ in shapes) + 4096, device="cuda", generator=g)   # "the flat gradient buffer": accumulate into it
    before = flat.clone()
    off = 0
    for rows, n_out, n_in, sk in shapes:
        big = torch.randn(rows, n_out + 256, device="cuda", generator=g).bfloat16()
        dy = big[:, 128:128 + n_out]                                   # a column slice: leading dimension != width
        x = torch.randn(rows, n_in, device="cuda", generator=g).bfloat16()
        dw = flat[off:off + n_out * n_in].view(n_out, n_in)
        jobs.append((dy, x, dw, sk))
        refs.append((off, dy.float().t() @ x.float()))
        off += n_out * n_in
        if n_out % 256 == 0:
            db = torch.zeros(n_out, device="cuda")
            bjobs.append((dy, db))
            brefs.append(dy.float().sum(0))
    ops.wgrad_grouped(jobs)
    ops.colsum_grouped(bjobs)
    torch.cuda.synchronize()
    for (o, ref), (_, _, dw, _) in zip(refs, jobs):
        got = dw - before[o:o + dw.numel()].view_as(dw)
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, err                                         # fp32 accumulation of exact bf16 products: summation order only
    assert torch.equal(flat[off:], before[off:])                       # nothing written past the last job
    for (_, db), ref in zip(bjobs, brefs):
        assert ((db - ref).abs().max() / ref.abs().max()).item() < 2e-5
    # the bias gradients riding on the weight-gradient products (RalfWgradJob.db: column sums of the dy tiles the product reads anyway), accumulated
    # into a buffer that already holds something; whole reductions and the slab form
    flat2 = before.clone()
    jobs2, pre, off = [], [], 0
    for (dy, x, _, sk), (rows, n_out, n_in, _) in zip(jobs, shapes):
        db = torch.randn(n_out, device="cuda", generator=g) if n_out % 256 == 0 else None
        pre.append(db.clone() if db is not None else None)
        jobs2.append((dy, x, flat2[off:off + n_out * n_in].view(n_out, n_in), sk, db))
        off += n_out * n_in
    ops.wgrad_grouped(jobs2)
    torch.cuda.synchronize()
    assert torch.equal(flat2, flat)                                    # the weight gradients do not notice
    nb = 0
    for (dy, _, _, _, db), p0 in zip(jobs2, pre):
        if db is not None:
            ref = dy.float().sum(0)
            assert (((db - p0) - ref).abs().max() / ref.abs().max()).item() < 2e-5
            nb += 1
    assert nb == len(bjobs)


_FEW_ROW_SNIPPET = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from ralf_amd import ops
g = torch.Generator().manual_seed(7)
out = {}
for (M, N, K, act, res) in [(256, 256, 256, None, True), (256, 1024, 256, "gelu", False), (256, 256, 1024, None, True), (33, 518, 256, None, False), (500, 192, 64, "relu", False),
                            (33, 72, 512, "relu", True), (1, 256, 1024, None, False)]:
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda() if res else None
    out[f"{M}x{N}x{K}"] = ops.gemm(A, W, M, N, K, bias=b, act=act, res=r, few_row_split=True).cpu()
    assert torch.equal(ops.gemm(A, W, M, N, K, bias=b, act=act, res=r).cpu(), out[f"{M}x{N}x{K}"]) or (K % 512 == 0)   # without the flag: the single chain
    # the fp32 parity mode's few-row kernel (gemm_skinny_f32_kernel, K % 256 == 0): 16x16 tiles, the tiled kernel's bits; fp32 and bf16-to-fp32 outputs
    out[f"{M}x{N}x{K}xfp32"] = ops.gemm(A.float(), W.float(), M, N, K, bias=b, act=act, res=r.float() if res else None).cpu()
torch.save(out, sys.argv[2])
"""


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (4096, 256, 64), (16384, 64, 64), (1088 * 64, 256, 64), (128, 2048, 64), (8192, 512, 128), (192, 128, 128)])
def test_conv1x1_with_64_input_channels(M, N, K):
    """ralf_conv1x1_k64 (a wave per 64 x 64 tile, weights in registers) against ralf_gemm: the same output bits; the statistics partials of the
    same stored values (sums in another order); the inference epilogue (scale / shift / residual / ReLU before or after it) like the tiled one."""
    from ralf_amd import ops

    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.1).bfloat16()
    assert ops.conv1x1_k64_ok(x, N, any_k=True) and ops.conv1x1_k64_ok(x, N) == (K == 64)
    want = ops.gemm(x, w, M, N, K)
    assert torch.equal(ops.conv1x1_k64(x, w), want)
    cst, cst0 = ops.colstats_buffer(M, N, x.device), ops.colstats_buffer(M, N, x.device)
    want2 = ops.gemm(x, w, M, N, K, colstats=cst0)
    got2 = ops.conv1x1_k64(x, w, colstats=cst)
    assert torch.equal(got2, want2)
    ref = want.float().view(M // 64, 64, N)
    assert torch.allclose(cst[:, 0], ref.sum(1), rtol=1e-5, atol=1e-4) and torch.allclose(cst[:, 1], (ref * ref).sum(1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(cst, cst0, rtol=1e-5, atol=1e-4)
    sc, sh = torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    for relu, act, res in ((1, "relu", None), (2, "relu_post", r), (0, None, r), (0, None, None)):
        a = ops.conv1x1_k64(x, w, scale=sc, shift=sh, res=res, relu=relu)
        b = ops.gemm(x, w, M, N, K, bias=sh, colscale=sc, act=act, res=res)
        assert float((a.float() - b.float()).abs().max()) <= 2.0 ** -7 * float(b.float().abs().max()), (relu, act)
        assert float((a != b).float().mean()) < 0.01, (relu, act)      # (the same bits but for where the compiler contracts scale * x + shift)
    with pytest.raises(RuntimeError):
        ops.conv1x1_k64(x[:32], w)


@pytest.mark.parametrize("M,N,act,f32out", [(256, 1024, "relu", False), (256, 518, None, True), (33, 256, None, False), (512, 64, "gelu", False)])
def test_layernorm_prologue_of_the_few_row_product(M, N, act, f32out):
    """RalfGemmDesc.ln_*: LayerNorm of the A rows inside the few-row kernel against ralf_layernorm_fwd + the plain product (same formulas;
    the row sums are taken in another order, so a normalised element may round to the neighbouring bf16 value now and then)."""
    from ralf_amd import ops

    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = (torch.randn(M, 256, device="cuda", generator=g) * 2 + 0.3).bfloat16()
    W = (torch.randn(N, 256, device="cuda", generator=g) * 0.05).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    lg, lb = torch.rand(256, device="cuda", generator=g) + 0.5, torch.randn(256, device="cuda", generator=g) * 0.1
    assert ops.gemm_ln_ok(x, M, 256)
    odt = torch.float32 if f32out else None
    got = ops.gemm(x, W, M, N, 256, bias=b, act=act, ln=(lg, lb, 1e-5), out_dtype=odt)
    h, _, _ = ops.layernorm_fwd(x, lg, lb, save_stats=False)
    want = ops.gemm(h, W, M, N, 256, bias=b, act=act, out_dtype=odt)
    assert float((got.float() - want.float()).abs().max()) <= 0.02 * float(want.float().abs().max())
    assert float((got.float() != want.float()).float().mean()) < 0.2     # mostly the same bits
    ref = torch.nn.functional.layer_norm(x.float(), (256,), lg, lb).bfloat16().float() @ W.float().t() + b
    ref = torch.relu(ref) if act == "relu" else (torch.nn.functional.gelu(ref) if act == "gelu" else ref)
    assert float((got.float() - ref).abs().max()) <= 0.03 * float(ref.abs().max())
    with pytest.raises(RuntimeError):   # rows wider than the kernel keeps in registers
        ops.gemm(torch.zeros(M, 512, device="cuda", dtype=torch.bfloat16), torch.zeros(N, 512, device="cuda", dtype=torch.bfloat16), M, N, 512, ln=(lg, lb, 1e-5))


def test_few_row_kernel_is_bit_identical_to_the_tiled_kernel(tmp_path):
    """gemm_skinny_kernel (bf16 NT products with M <= 512: the decode step's linear layers) runs the tiled kernel's MFMA chain in the same
    order: the two builds of the same product agree bit for bit (RALF_GEMM_SKINNY_ROWS=0 switches the few-row path off in a child process)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for rows, split in (("512", "0"), ("0", "0"), ("512", "1")):
        f = str(tmp_path / f"o{rows}_{split}.pt")
        env = dict(os.environ, RALF_GEMM_SKINNY_ROWS=rows, RALF_GEMM_SKINNY_SPLIT=split)
        subprocess.run([sys.executable, "-c", _FEW_ROW_SNIPPET, root, f], check=True, env=env, timeout=600)
        outs.append(torch.load(f))
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
        # reductions of K % 512 == 0 (the feed-forward block's second product) run as four quarter chains summed in wave order: the same
        # product to bf16 rounding, no longer the same bits
        K = int(k.split("x")[2])
        if K % 512 or k.endswith("fp32"):
            assert torch.equal(outs[2][k], outs[1][k]), k
        else:
            torch.testing.assert_close(outs[2][k].float(), outs[1][k].float(), rtol=1.6e-2, atol=1e-2)


_GLDS_CASES = """
import sys, torch
sys.path.insert(0, {root!r})
from ralf_amd import ops
g = torch.Generator(device="cuda").manual_seed(11)
out = {{}}
seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
for M, N, K in [(16384, 1024, 256), (16384, 256, 1024), (3300, 768, 256), (4096, 512, 2048), (200, 256, 128), (33792, 264, 64), (8192, 128, 4096)]:
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    out[(M, N, K, "plain")] = ops.gemm(x, w, M, N, K)
    out[(M, N, K, "bias_relu")] = ops.gemm(x, w, M, N, K, bias=b, act="relu")
    out[(M, N, K, "bias_res_drop")] = ops.gemm(x, w, M, N, K, bias=b, res=r, drop_p=0.1, seed=seed, call_id=3)
    out[(M, N, K, "f32")] = ops.gemm(x, w, M, N, K, out_dtype=torch.float32)
    if N % 128 == 0 and M % 64 == 0:
        z = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        out[(M, N, K, "gelu_out2")] = ops.gemm(x, w, M, N, K, bias=b, act="gelu", out2=z)
        out[(M, N, K, "gelu_pre")] = z
        cst = ops.colstats_buffer(M, N, x.device)
        out[(M, N, K, "stats_y")] = ops.gemm(x, w, M, N, K, colstats=cst)
        out[(M, N, K, "stats")] = cst
# NN (data gradient of a linear layer: B = W [K][N]) and TN (weight gradient: both operands [rows][cols], reduction over the rows)
for M, N, K in [(16384, 256, 1024), (16384, 1024, 256), (3300, 256, 768), (4096, 2048, 512), (65536, 128, 512)]:
    dy = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(K, N, device="cuda", generator=g) * 0.05).bfloat16()
    r = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    out[(M, N, K, "nn")] = ops.gemm(dy, w, M, N, K, b_kcontig=False)
    out[(M, N, K, "nn_res_mask")] = ops.gemm(dy, w, M, N, K, b_kcontig=False, res=r, aux=r, aux_mode="relu_mask", aux_scale=1.25)
for rows, n_out, n_in, sk in [(16384, 256, 1024, 4), (16384, 768, 256, 2), (4096, 512, 2048, 1), (3200, 128, 256, 1), (1000, 64, 192, 1)]:
    dy = torch.randn(rows, n_out, device="cuda", generator=g).bfloat16()
    x = torch.randn(rows, n_in, device="cuda", generator=g).bfloat16()
    out[(rows, n_out, n_in, "tn")] = ops.gemm(dy, x, n_out, n_in, rows, a_kcontig=False, b_kcontig=False, out_dtype=torch.float32, splitk=sk)
# grouped weight gradients (ralf_wgrad_grouped)
jobs = []
flat = torch.zeros(256 * 1024 + 768 * 256 + 512 * 256 + 64, device="cuda")
off = 0
for rows, n_out, n_in, sk in [(16384, 256, 1024, 1), (16384, 768, 256, 2), (34048, 512, 256, 3)]:
    dy = torch.randn(rows, n_out, device="cuda", generator=g).bfloat16()
    x = torch.randn(rows, n_in, device="cuda", generator=g).bfloat16()
    jobs.append((dy, x, flat[off:off + n_out * n_in].view(n_out, n_in), sk))
    off += n_out * n_in
ops.wgrad_grouped(jobs)
out[("grouped",)] = flat
# weight gradients of k x k convolutions through the pixel gather (TN, split reduction)
for B, H, C, Co, k, stride in [(8, 32, 64, 128, 3, 1), (4, 30, 128, 64, 3, 2), (64, 16, 256, 256, 3, 1), (16, 16, 256, 512, 1, 2), (2, 64, 8, 64, 7, 2)]:
    pad = k // 2
    OH = (H + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, H, C, device="cuda", generator=g).bfloat16()
    dy = torch.randn(B * OH * OH, Co, device="cuda", generator=g).bfloat16()
    geom = dict(RH=OH, RW=OH, SH=H, SW=H, SC=C, KH=k, KW=k, stride=stride, pad=pad, mode=0)
    for sk in (1, 5):
        out[("conv_wgrad", B, H, C, Co, k, stride, sk)] = ops.gemm(dy, x, Co, k * k * C, B * OH * OH, a_kcontig=False, b_kcontig=False, conv=geom, gather=2,
                                                                  out_dtype=torch.float32, splitk=sk)
# k x k convolutions through the tap-uniform gather: forward (mode 0) and data gradient (mode 1, stride 1 and 2), with padding
for B, H, C, Co, k, stride in [(8, 32, 64, 128, 3, 1), (4, 30, 128, 64, 3, 2), (64, 16, 256, 256, 3, 1), (16, 16, 256, 512, 1, 2)]:
    pad = k // 2
    OH = (H + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, H, C, device="cuda", generator=g).bfloat16()
    w = (torch.randn(Co, k, k, C, device="cuda", generator=g) * 0.05).bfloat16()
    M = B * OH * OH
    geom = dict(RH=OH, RW=OH, SH=H, SW=H, SC=C, KH=k, KW=k, stride=stride, pad=pad, mode=0)
    cst = ops.colstats_buffer(M, Co, x.device) if Co % 64 == 0 and M % 64 == 0 else None
    out[("conv_fwd", B, H, C, Co, k, stride)] = ops.gemm(x, w, M, Co, k * k * C, conv=geom, gather=1, colstats=cst)
    if cst is not None:
        out[("conv_fwd_stats", B, H, C, Co, k, stride)] = cst
    dy = torch.randn(B, OH, OH, Co, device="cuda", generator=g).bfloat16()
    wt = (torch.randn(C, k, k, Co, device="cuda", generator=g) * 0.05).bfloat16()
    geom = dict(RH=H, RW=H, SH=OH, SW=OH, SC=Co, KH=k, KW=k, stride=stride, pad=pad, mode=1)
    r = torch.randn(B * H * H, C, device="cuda", generator=g).bfloat16()
    out[("conv_dgrad", B, H, C, Co, k, stride)] = ops.gemm(dy, wt, B * H * H, C, k * k * Co, conv=geom, gather=1, res=r)
torch.cuda.synchronize()
torch.save({{k: v.cpu() for k, v in out.items()}}, {path!r})
"""


def test_direct_to_lds_kernels_are_bit_identical_to_register_staged(tmp_path):
    """the aligned NT products on 128x128 tiles and the grouped weight gradients run on the direct-to-LDS kernels (global_load_lds ring,
    gemm_impl.h GATHER 5 / 6) by default and on the register-staged kernels with RALF_GEMM_GLDS=0 RALF_GEMM_GLDS_GROUPED=0 (read once
    per process): same MFMA chain in the same order, so every output -- plain, fused epilogues, fp32 logits, the pre-activation copy,
    the BatchNorm column statistics, the grouped fp32 accumulations -- is the same bits (the other layouts / gathers are the same
    kernels in both runs).  Shapes: both tile sizes, the 3-stage ring (<= 256 tiles, K >= 1024), ragged M and N, a single k-tile."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flag in ("1", "0"):
        path = str(tmp_path / f"glds{flag}.pt")
        env = dict(os.environ, RALF_GEMM_GLDS=flag, RALF_GEMM_GLDS_GROUPED=flag)
        r = subprocess.run([sys.executable, "-c", _GLDS_CASES.format(root=root, path=path)], env=env,
                           capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res[flag] = torch.load(path)
    assert set(res["1"]) == set(res["0"]) and len(res["1"]) >= 70
    for k in res["1"]:
        assert torch.equal(res["1"][k], res["0"][k]), k
    x = res["1"][(16384, 1024, 256, "plain")]
    assert torch.isfinite(x.float()).all() and x.float().abs().max() > 0.1


_PATCH_CASES = """
import sys, torch
sys.path.insert(0, {root!r})
from ralf_amd import ops
dt = torch.bfloat16
out = {{}}
for (B, H, C) in [(16, 64, 64), (64, 32, 128), (32, 32, 128), (64, 16, 256), (2, 16, 256), (16, (32, 64), 128)]:
    H, W = H if isinstance(H, tuple) else (H, H)
    M = B * H * W
    g = torch.Generator(device="cuda").manual_seed(H + B)
    x = torch.randn(M, C, device="cuda", generator=g).to(dt)
    w = (torch.randn(C, 3, 3, C, device="cuda", generator=g) * 0.05).to(dt)
    bias = torch.randn(C, device="cuda", generator=g)
    skip = torch.randn(M, C, device="cuda", generator=g).to(dt)
    xa = torch.randn(M, C, device="cuda", generator=g).to(dt)
    mean = torch.randn(C, device="cuda", generator=g)
    bits = torch.randint(0, 256, (M * C // 8,), device="cuda", dtype=torch.uint8, generator=g)
    gf = dict(RH=H, RW=W, SH=H, SW=W, SC=C, KH=3, KW=3, stride=1, pad=1, mode=0)
    gd = dict(gf, mode=1)
    st = torch.full(((M + 63) // 64, 2, C), float("nan"), device="cuda")
    part = torch.full(((M + 63) // 64, 2, C), float("nan"), device="cuda")
    k = (B, H if H == W else (H, W), C)
    out[k + ("variant",)] = torch.tensor([ops.gemm_patch_variant(M, C, 9 * C, conv=gf), ops.gemm_patch_variant(M, C, 9 * C, conv=gd)])
    out[k + ("fwd",)] = ops.gemm(x, w, M, C, 9 * C, conv=gf, gather=1, colstats=st)
    out[k + ("fwd_stats",)] = st
    out[k + ("fwd_eval",)] = ops.gemm(x, w, M, C, 9 * C, conv=gf, gather=1, bias=bias, act="relu")
    out[k + ("dgrad",)] = ops.gemm(x, w, M, C, 9 * C, conv=gd, gather=1, res=skip)
    out[k + ("dgrad_bnb",)] = ops.gemm(x, w, M, C, 9 * C, conv=gd, gather=1, res=skip, bnb=(xa, bits, mean, part))
    out[k + ("dgrad_bnb_part",)] = part
torch.cuda.synchronize()
torch.save({{k: v.cpu() for k, v in out.items()}}, {path!r})
"""


def test_conv3x3_with_the_input_patch_in_lds_is_bit_identical_to_the_tap_gather(tmp_path):
    """3 x 3 / stride-1 convolutions (timm Bottleneck.conv2 of layer1..3, common/image.py:39-48) take the patch form (gemm_impl.h GATHER 15: the tile's
    halo patch resident in LDS, only the weights stream) where whole image rows make a tile and enough tiles exist: 256 x 64 tiles (layer1), 256 x 128
    (layer2 at B = 64), 128 x 128 (layer2 at B = 32, layer3); RALF_GEMM_PATCH=0 (read once per process) keeps the tap gather.  Same k-tiles in the same
    order on the same MFMA chain: forward with column statistics, the eval form (bias + ReLU), data gradient with a skip gradient, data gradient with
    the BatchNorm-backward epilogue -- the same bits.  The smallest case (B = 2) has too few tiles and stays on the tap gather in both runs."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flag in ("1", "0"):
        path = str(tmp_path / f"patch{flag}.pt")
        r = subprocess.run([sys.executable, "-c", _PATCH_CASES.format(root=root, path=path)], env=dict(os.environ, RALF_GEMM_PATCH=flag),
                           capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res[flag] = torch.load(path)
    want = {(16, 64, 64): 3, (64, 32, 128): 2, (32, 32, 128): 1, (64, 16, 256): 1, (2, 16, 256): 0, (16, (32, 64), 128): 1}   # (the last: 32 x 64-pixel maps)
    for k, v in want.items():
        assert res["1"][k + ("variant",)].tolist() == [v, v] and res["0"][k + ("variant",)].tolist() == [0, 0], k
    for k in res["1"]:
        if k[-1] != "variant":
            assert torch.equal(res["1"][k], res["0"][k]), k
    # ... and the convolution itself against torch (one case per variant, forward and data gradient)
    for (B, H, C) in [(16, 64, 64), (64, 32, 128), (64, 16, 256), (16, (32, 64), 128)]:
        key = (B, H, C)
        H, W = H if isinstance(H, tuple) else (H, H)
        M = B * H * W
        g = torch.Generator(device="cuda").manual_seed(H + B)   # (the subprocess's operands: x, w, bias, skip in its order)
        x = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(C, 3, 3, C, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        torch.randn(C, device="cuda", generator=g)
        skip = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
        xi = x.float().view(B, H, W, C).permute(0, 3, 1, 2)
        y = F.conv2d(xi, w.float().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1).reshape(M, C)               # weights [co][kh][kw][ci]
        torch.testing.assert_close(res["1"][key + ("fwd",)].float(), y.cpu(), atol=6e-2, rtol=2e-2)
        dx = F.conv_transpose2d(xi, w.float().permute(3, 0, 1, 2), None, 1, 1).permute(0, 2, 3, 1).reshape(M, C)   # weights [ci][kh][kw][co]
        torch.testing.assert_close(res["1"][key + ("dgrad",)].float(), (dx + skip.float()).cpu(), atol=6e-2, rtol=2e-2)


# ---- operand transform with write-through (RalfGemmDesc.at_*): BatchNorm apply / backward apply inside the loader of a 1x1 convolution ----
@pytest.mark.parametrize("M,K,N,res", [(4096 + 40, 256, 64, True), (4096, 512, 128, True), (8192 + 8, 64, 256, False), (2048, 128, 512, False), (2048, 256, 1024, False)])
def test_operand_transform_forward_equals_bn_apply_then_product(M, K, N, res):
    """at_mode 1: relu(A * scale + shift (+ res)) in the A loader, written through with its ReLU bits by the first column tile, product with
    column statistics -- everything bit-identical to ralf_bn_apply followed by the plain product (64x64 and 128x128 tiles, ragged M,
    one and several column tiles)"""
    from ralf_amd import ops

    y, W = rnd(M, K, seed=1).to(torch.bfloat16).cuda(), (rnd(N, K, seed=2) * K ** -0.5).to(torch.bfloat16).cuda()
    r = rnd(M, K, seed=3).to(torch.bfloat16).cuda() if res else None
    sc, sh = (0.5 + torch.rand(K, generator=torch.Generator().manual_seed(4))).cuda(), (0.3 * rnd(K, seed=5)).cuda()
    z, mask = torch.empty_like(y), torch.empty(M * K // 8, dtype=torch.uint8, device="cuda")
    stats = N % 64 == 0
    cst = ops.colstats_buffer(M, N, y.device) if stats else None
    ops._call("ralf_bn_apply", ops.dtype_code(y), ops._p(y), ops._p(sc), ops._p(sh), ops._p(r), ops._p(z), ops._p(mask), M, K, 1)
    want = ops.gemm(z, W, M, N, K, colstats=cst)
    z2, mask2 = torch.full_like(y, 7.0), torch.zeros_like(mask)
    cst2 = ops.colstats_buffer(M, N, y.device) if stats else None
    got = ops.gemm(y, W, M, N, K, colstats=cst2, at=dict(mode=1, c1=sc, c2=sh, a2=r, out=z2, mask=mask2, relu=True))
    assert torch.equal(got, want) and torch.equal(z2, z) and torch.equal(mask2, mask)
    if stats:
        assert torch.equal(cst2, cst)
    torch.testing.assert_close(z.float().cpu(), torch.relu(y.float().cpu() * sc.cpu() + sh.cpu() + (r.float().cpu() if res else 0)), atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("M,K,N,bnb", [(4096 + 40, 256, 64, True), (4096, 512, 128, True), (4096, 64, 256, True), (2048, 128, 512, False)])
def test_operand_transform_backward_equals_bn_bwd_apply_then_data_gradient(M, K, N, bnb):
    """at_mode 2: dy = c1 dz + c2 x + c3 in the A loader of the data-gradient product (NN), written through; with the BatchNorm-backward
    reductions of the NEXT layer in the epilogue (bnb_*): bit-identical to ralf_bn_bwd_apply_affine followed by the product.  The
    coefficients come from ralf_bn_bwd_stats_from_partials and reproduce torch's batch-norm backward."""
    from ralf_amd import ops

    dev = "cuda"
    dz, x, W = rnd(M, K, seed=1).to(torch.bfloat16).cuda(), rnd(M, K, seed=2).to(torch.bfloat16).cuda(), (rnd(K, N, seed=3) * K ** -0.5).to(torch.bfloat16).cuda()
    gamma, mean = (0.5 + torch.rand(K, generator=torch.Generator().manual_seed(4))).cuda(), x.float().mean(0)
    rstd = (x.float().var(0, unbiased=False) + 1e-5).rsqrt()
    # the partial rows a data-gradient epilogue would have written: per 64-row block, sum dz and sum dz (x - mean)
    nb = (M + 63) // 64
    pad = nb * 64 - M
    dzp, xp = torch.nn.functional.pad(dz.float(), (0, 0, 0, pad)), torch.nn.functional.pad(x.float() - mean, (0, 0, 0, pad))
    part = torch.stack([dzp.view(nb, 64, K).sum(1), (dzp * xp).view(nb, 64, K).sum(1)], dim=1).contiguous()
    s1, s2, coef = torch.zeros(K, device=dev), torch.zeros(K, device=dev), torch.empty(3, K, device=dev)
    ops._call("ralf_bn_bwd_stats_from_partials", ops._p(part), nb, ops._p(rstd), ops._p(s1), ops._p(s2), K, ops._p(torch.empty(128 * 2 * K, device=dev)),
              ops._p(gamma), ops._p(mean), M, ops._p(coef))
    dy = torch.empty_like(dz)
    ops._call("ralf_bn_bwd_apply_affine", ops.dtype_code(dz), ops._p(dz), ops._p(x), ops._p(coef[0]), ops._p(coef[1]), ops._p(coef[2]), ops._p(dy), M, K)
    xh = (x.float() - mean) * rstd
    ref = gamma * rstd * (dz.float() - dz.float().mean(0) - xh * (dz.float() * xh).mean(0))          # torch's native_batch_norm_backward (training)
    torch.testing.assert_close(dy.float(), ref, atol=3e-2, rtol=2e-2)
    extra, extra2 = {}, {}
    if bnb:
        bx, bm = rnd(M, N, seed=6).to(torch.bfloat16).cuda(), torch.randint(0, 256, (M * N // 8,), dtype=torch.uint8, generator=torch.Generator().manual_seed(7)).cuda()
        bmean, res = rnd(N, seed=8).cuda(), rnd(M, N, seed=9).to(torch.bfloat16).cuda()
        p1, p2 = torch.empty(nb, 2, N, device=dev), torch.empty(nb, 2, N, device=dev)
        extra, extra2 = dict(res=res, bnb=(bx, bm, bmean, p1)), dict(res=res, bnb=(bx, bm, bmean, p2))
    want = ops.gemm(dy, W, M, N, K, b_kcontig=False, **extra)
    dy2 = torch.full_like(dz, 7.0)
    got = ops.gemm(dz, W, M, N, K, b_kcontig=False, at=dict(mode=2, c1=coef[0], c2=coef[1], c3=coef[2], a2=x, out=dy2), **extra2)
    assert torch.equal(got, want) and torch.equal(dy2, dy)
    if bnb:
        assert torch.equal(p1, p2)


def test_operand_transform_refuses_what_it_does_not_cover():
    from ralf_amd import _lib, ops

    y, W = rnd(256, 1024, seed=1).to(torch.bfloat16).cuda(), rnd(64, 1024, seed=2).to(torch.bfloat16).cuda()
    c = torch.ones(1024, device="cuda")
    with pytest.raises(_lib.RalfHipError, match="K <= 512"):
        ops.gemm(y, W, 256, 64, 1024, at=dict(mode=1, c1=c, c2=c))


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 64, 64, 64, 64), (3, 32, 32, 128, 128), (4, 16, 16, 256, 256), (5, 8, 8, 512, 512), (2, 16, 32, 64, 128)])
def test_conv3x3_weight_gradient_direct_form(B, H, W, Ci, Co):
    """ralf_conv3x3_wgrad (dy tile and halo patch staged once for all nine taps; OIHW fp32 output) against torch's conv2d weight gradient on
    the same bf16 operands and against the implicit-GEMM form (ralf_gemm gather = 2) it replaces -- the four bottleneck geometries (one, two,
    four and eight image rows per 64-pixel tile), image borders, accumulate"""
    from ralf_amd import ops

    x = rnd(B, H, W, Ci, seed=1).to(torch.bfloat16).cuda()
    dy = rnd(B, H, W, Co, seed=2).to(torch.bfloat16).cuda()
    got = ops.conv3x3_wgrad(dy, x)
    ref = torch.nn.grad.conv2d_weight(x.float().cpu().permute(0, 3, 1, 2), (Co, Ci, 3, 3), dy.float().cpu().permute(0, 3, 1, 2), stride=1, padding=1)
    scale = (B * H * W) ** 0.5
    torch.testing.assert_close(got.cpu() / scale, ref / scale, atol=2e-3, rtol=2e-3)
    M = B * H * W
    geom = dict(RH=H, RW=W, SH=H, SW=W, SC=Ci, KH=3, KW=3, stride=1, pad=1, mode=0)
    g = ops.gemm(dy.view(M, Co), x, Co, 9 * Ci, M, a_kcontig=False, b_kcontig=False, conv=geom, gather=2, out_dtype=torch.float32, splitk=4)
    old = g.view(Co, 3, 3, Ci).permute(0, 3, 1, 2)
    torch.testing.assert_close(got / scale, old / scale, atol=1e-4, rtol=1e-4)       # same products, fp32 sums in another order
    # stride 2 (the first block of layer2-4): output grid [H/2, W/2] over the same input
    if W >= 16:
        dy2 = rnd(B, H // 2, W // 2, Co, seed=3).to(torch.bfloat16).cuda()
        got2 = ops.conv3x3_wgrad(dy2, x, stride=2)
        ref2 = torch.nn.grad.conv2d_weight(x.float().cpu().permute(0, 3, 1, 2), (Co, Ci, 3, 3), dy2.float().cpu().permute(0, 3, 1, 2), stride=2, padding=1)
        s2 = (B * H * W / 4) ** 0.5
        torch.testing.assert_close(got2.cpu() / s2, ref2 / s2, atol=2e-3, rtol=2e-3)
    acc = torch.ones(Co, Ci, 3, 3, device="cuda")
    ops.conv3x3_wgrad(dy, x, out=acc, accumulate=True)
    torch.testing.assert_close(acc, got + 1.0, atol=1e-5, rtol=1e-6)
    assert torch.equal(ops.conv3x3_wgrad(dy, x), got)                                   # deterministic


@pytest.mark.parametrize("B,IH,IW", [(2, 64, 64), (1, 350, 240), (3, 32, 96), (1, 30, 300)])
def test_stem_convolution_direct_form(B, IH, IW):
    """ralf_stem7x7_fwd (7x7 / 2 / pad 3 on 8-channel pixels, one output row per tile, statistics from its epilogue) against torch's conv2d on the
    same bf16 operands and against the implicit-GEMM form it replaces; odd sizes, rows wider than one tile (OW > 128), image borders"""
    from ralf_amd import ops

    x = torch.zeros(B, IH, IW, 8)
    x[..., :4] = rnd(B, IH, IW, 4, seed=1)
    x = x.to(torch.bfloat16).cuda()
    w = torch.zeros(64, 7, 7, 8)
    w[..., :4] = rnd(64, 7, 7, 4, seed=2) * 0.1
    w = w.to(torch.bfloat16).cuda()
    y, part = ops.stem7x7_fwd(x, w)
    OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
    ref = F.conv2d(x[..., :4].float().cpu().permute(0, 3, 1, 2), w[..., :4].float().cpu().permute(0, 3, 1, 2), None, 2, 3).permute(0, 2, 3, 1)
    assert tuple(y.shape) == (B, OH, OW, 64)
    torch.testing.assert_close(y.float().cpu(), ref, atol=2e-2, rtol=2e-2)
    M = B * OH * OW
    geom = dict(RH=OH, RW=OW, SH=IH, SW=IW, SC=8, KH=7, KW=7, stride=2, pad=3, mode=0)
    old = ops.gemm(x, w, M, 64, 7 * 7 * 8, conv=geom, gather=1).view(B, OH, OW, 64)
    assert (y.float() - old.float()).abs().max().item() <= 2.0 ** -7 * old.float().abs().max().item()   # same products, another fp32 summation order: <= 1 bf16 ulp
    # ... and its weight gradient against torch's (fp32 accumulation of bf16 products)
    dyt = rnd(B, OH, OW, 64, seed=3).to(torch.bfloat16).cuda()
    dW = ops.stem7x7_wgrad(x, dyt)
    refw = torch.nn.grad.conv2d_weight(x[..., :4].float().cpu().permute(0, 3, 1, 2), (64, 4, 7, 7), dyt.float().cpu().permute(0, 3, 1, 2), stride=2, padding=3)
    sc = (B * OH * OW) ** 0.5
    torch.testing.assert_close(dW.cpu() / sc, refw / sc, atol=2e-3, rtol=2e-3)
    acc = torch.ones(64, 4, 7, 7, device="cuda")
    ops.stem7x7_wgrad(x, dyt, out=acc, accumulate=True)
    torch.testing.assert_close(acc, dW + 1.0, atol=1e-4, rtol=1e-6)
    assert torch.equal(ops.stem7x7_wgrad(x, dyt), dW)
    s = part.sum(0)
    yf = y.float().view(-1, 64)
    torch.testing.assert_close(s[0], yf.sum(0), atol=1e-2, rtol=1e-4)
    torch.testing.assert_close(s[1], (yf * yf).sum(0), atol=1e-2, rtol=1e-4)
