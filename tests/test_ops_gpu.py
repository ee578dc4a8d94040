"""GPU numerics of the non-GEMM kernels (through the C ABI) vs plain torch fp32 references on CPU."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: dict(atol=2e-5, rtol=2e-5), torch.bfloat16: dict(atol=3e-2, rtol=3e-2)}


def rnd(*shape, seed=0, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def close(a, b, dtype, frac=1.0, **kw):
    tol = dict(TOL[dtype]); tol.update(kw)
    a, b = a.float().cpu(), b.float().cpu()
    if frac < 1.0:  # bf16 ReLU masks may flip where the fp32 reference output is ~0: allow a few outliers
        ok = (a - b).abs() <= tol["atol"] + tol["rtol"] * b.abs()
        assert ok.float().mean().item() >= frac, f"only {ok.float().mean().item():.5f} within tolerance"
        return
    torch.testing.assert_close(a, b, **tol)


@pytest.mark.parametrize("dtype", DT)
def test_layernorm(dtype):
    from ralf_amd import ops

    x = rnd(37, 5, 256, seed=1, dtype=dtype).float().requires_grad_(True)
    g, b = (1 + 0.1 * rnd(256, seed=2)).requires_grad_(True), rnd(256, seed=3).requires_grad_(True)
    y = F.layer_norm(x, (256,), g, b)
    go = rnd(37, 5, 256, seed=4, dtype=dtype).float()
    y.backward(go)
    yd, mean, rstd = ops.layernorm_fwd(x.detach().to(dtype).cuda(), g.detach().cuda(), b.detach().cuda())
    close(yd, y.detach(), dtype)
    dx, dg, db = ops.layernorm_bwd(go.to(dtype).cuda(), x.detach().to(dtype).cuda(), g.detach().cuda(), mean, rstd)
    close(dx, x.grad, dtype)
    close(dg, g.grad, dtype, atol=5e-2 if dtype == torch.bfloat16 else 1e-4)
    close(db, b.grad, dtype, atol=5e-2 if dtype == torch.bfloat16 else 1e-4)
    close(ops.colsum(go.to(dtype).view(-1, 256).cuda(), 185, 256), go.view(-1, 256).sum(0), dtype, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,cols", [(1, 256), (33, 512), (2051, 256), (16384 + 3, 256)])
def test_layernorm_bwd_shapes(dtype, rows, cols):
    """odd row counts (two-rows-in-flight tail), the 512-wide variant, residual-skip addend, accumulation into dgamma/dbeta"""
    from ralf_amd import ops

    x = rnd(rows, cols, seed=11, dtype=dtype).float().requires_grad_(True)
    g, b = (1 + 0.1 * rnd(cols, seed=12)).requires_grad_(True), rnd(cols, seed=13).requires_grad_(True)
    go, sk = rnd(rows, cols, seed=14, dtype=dtype).float(), rnd(rows, cols, seed=15, dtype=dtype).float()
    F.layer_norm(x, (cols,), g, b).backward(go)
    _, mean, rstd = ops.layernorm_fwd(x.detach().to(dtype).cuda(), g.detach().cuda(), b.detach().cuda())
    gg, gb = torch.ones(cols, device="cuda"), torch.full((cols,), 2.0, device="cuda")
    dx, _, _ = ops.layernorm_bwd(go.to(dtype).cuda(), x.detach().to(dtype).cuda(), g.detach().cuda(), mean, rstd, need_wgrad=True, into=(gg, gb), skip=sk.to(dtype).cuda())
    close(dx, x.grad + sk, dtype)
    tol = dict(atol=0.5, rtol=2e-2) if dtype == torch.bfloat16 else dict(atol=2e-3, rtol=1e-4)
    close(gg, g.grad + 1.0, torch.float32, **tol)
    close(gb, b.grad + 2.0, torch.float32, **tol)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,cols,ld", [(1, 4, 4), (185, 256, 256), (3200, 518, 518), (16384, 768, 768), (5000, 256, 768), (777, 1024, 1024), (100, 30, 30)])
def test_colsum_shapes(dtype, rows, cols, ld):
    from ralf_amd import ops

    x = rnd(rows, ld, seed=21, dtype=dtype)
    out = torch.full((cols,), 3.0, device="cuda")
    ops.colsum(x.cuda(), rows, cols, ld=ld, out=out)
    close(out, x.float()[:, :cols].sum(0) + 3.0, torch.float32, atol=2e-2 if dtype == torch.bfloat16 else 2e-3, rtol=1e-3)


@pytest.mark.parametrize("dtype", DT)
def test_embedding_scalar_xent(dtype):
    from ralf_amd import ops

    W, pe = rnd(50, 256, seed=5), rnd(20, 256, seed=6)
    idx = torch.randint(0, 50, (7, 12), generator=torch.Generator().manual_seed(7))
    out = ops.embed_fwd(idx.cuda(), W.cuda(), pe.cuda(), 12, 16.0, dtype)
    close(out, W[idx] * 16.0 + pe[:12], dtype, atol=0.15 if dtype == torch.bfloat16 else 1e-5)
    dy = rnd(7, 12, 256, seed=8, dtype=dtype)
    ref = torch.zeros(50, 256).index_add_(0, idx.view(-1), dy.float().view(-1, 256) * 16.0)
    close(ops.embed_bwd(idx.cuda(), dy.cuda(), 50, 16.0), ref, torch.float32, atol=1e-3, rtol=1e-4)
    x, s = rnd(9, 256, seed=9, dtype=dtype), torch.tensor([0.37, -1.0])
    close(ops.add_scalar(x.cuda(), s.cuda()[1:]), x.float() - 1.0, dtype)
    close(ops.sum_all(x.cuda()), x.float().sum().view(1), dtype, atol=1e-2, rtol=1e-3)
    # cross entropy with label smoothing and ignore_index
    logits = rnd(64, 518, seed=10, scale=2.0).requires_grad_(True)
    tgt = torch.randint(0, 518, (64,), generator=torch.Generator().manual_seed(11))
    tgt[::5] = 515
    loss = F.cross_entropy(logits, tgt, ignore_index=515, label_smoothing=0.1)
    loss.backward()
    cl, dl = ops.xent(logits.detach().cuda(), tgt.cuda(), 515, 0.1, dtype)
    assert cl[0].item() == (tgt != 515).sum().item()
    close(cl[1], loss.detach(), torch.float32, atol=1e-5, rtol=1e-5)
    close(dl, logits.grad, dtype, atol=1e-6 if dtype == torch.float32 else 2e-4)


def test_dropout_is_counter_based():
    from ralf_amd import ops

    x = torch.ones(1 << 20, device="cuda")
    seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
    y1, y2, y3 = ops.dropout(x, 0.1, seed, 7), ops.dropout(x, 0.1, seed, 7), ops.dropout(x, 0.1, seed, 8)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    keep = (y1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 2e-3 and abs(y1.mean().item() - 1.0) < 5e-3
    u = sorted(y1.unique().tolist())
    assert len(u) == 2 and u[0] == 0.0 and abs(u[1] - 1 / 0.9) < 1e-6


def test_dropout_mask_statistics():
    """the element mask (csrc/common.h drop_hash4: one hash per 4 consecutive elements, 16-bit fields): keep rate at several p, and no
    correlation between the four fields of a group, neighbouring groups, rows 1024 apart, consecutive streams (call ids) and seeds"""
    from ralf_amd import ops

    n = 1 << 24
    x = torch.ones(n + (1 << 26), device="cuda")[-n:]   # (a view: the mask depends on the element INDEX only)
    seed = torch.tensor([987654321], dtype=torch.int64, device="cuda")
    for p in (0.1, 0.5, 0.9):
        k = (ops.dropout(x, p, seed, 3) != 0).float()
        assert abs(k.mean().item() - (1 - p)) < 6e-4, (p, k.mean().item())

    def corr(a, b):
        a, b = a - a.mean(), b - b.mean()
        return ((a * b).mean() / (a.std() * b.std())).item()

    k = (ops.dropout(x, 0.5, seed, 3) != 0).float()
    g = k.view(-1, 4)
    for i in range(4):
        for j in range(i + 1, 4):
            assert abs(corr(g[:, i], g[:, j])) < 1.5e-3, (i, j)
    assert abs(corr(k[:-4], k[4:])) < 1.5e-3 and abs(corr(k[:-1024], k[1024:])) < 1.5e-3 and abs(corr(k[:-1], k[1:])) < 1.5e-3
    k2 = (ops.dropout(x, 0.5, seed, 4) != 0).float()
    k3 = (ops.dropout(x, 0.5, seed + 1, 3) != 0).float()
    assert abs(corr(k, k2)) < 1.5e-3 and abs(corr(k, k3)) < 1.5e-3
    # group sums are binomial(4, 0.5): variance 1 (a mask that repeated a field would show 2 or 4)
    assert abs(g.sum(1).var().item() - 1.0) < 5e-3
    # tensors beyond 2^26 elements use the high index bits too
    big = torch.ones((1 << 26) + (1 << 22), device="cuda")
    kb = (ops.dropout(big, 0.5, seed, 3) != 0).float()
    assert abs(corr(kb[: 1 << 22], kb[1 << 26:])) < 3e-3 and abs(kb[1 << 26:].mean().item() - 0.5) < 2e-3


@pytest.mark.parametrize("dtype", DT)
def test_pool_upsample_permute(dtype):
    from ralf_amd import ops

    x = torch.relu(rnd(2, 16, 9, 11, seed=12, dtype=dtype).float()).requires_grad_(True)  # many exact-zero ties
    y = F.max_pool2d(x, 3, 2, 1)
    go = rnd(*y.shape, seed=13, dtype=dtype).float()
    y.backward(go)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    yd, arg = ops.maxpool_fwd(xn)
    close(yd.permute(0, 3, 1, 2), y.detach(), dtype, atol=0, rtol=0)
    dx = ops.maxpool_bwd(go.permute(0, 2, 3, 1).contiguous().to(dtype).cuda(), arg, tuple(xn.shape))
    close(dx.permute(0, 3, 1, 2), x.grad, dtype)
    for (ih, iw, oh, ow) in [(4, 4, 8, 8), (11, 8, 22, 15)]:
        src = rnd(2, 8, ih, iw, seed=14, dtype=dtype).float().requires_grad_(True)
        lat = rnd(2, 8, oh, ow, seed=15, dtype=dtype).float()
        up = F.interpolate(src, size=(oh, ow), mode="nearest")
        g1, g2 = rnd(*up.shape, seed=16, dtype=dtype).float(), rnd(*up.shape, seed=17, dtype=dtype).float()
        (up * g1 + (up + lat) * g2).sum().backward()
        nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
        upd, sd = ops.upsample_add(nh(src), nh(lat))
        close(upd.permute(0, 3, 1, 2), up.detach(), dtype, atol=0, rtol=0)
        close(sd.permute(0, 3, 1, 2), (up + lat).detach(), dtype)
        dsrc = ops.upsample_bwd(nh(g1), nh(g2), (2, ih, iw, 8))
        close(dsrc.permute(0, 3, 1, 2), src.grad, dtype, atol=6e-2 if dtype == torch.bfloat16 else 1e-5)
    w = rnd(6, 4, 7, 7, seed=18)
    ohwi = ops.permute4(w.cuda(), (6, 7, 7, 8), (4 * 49, 7, 1, 49), 4, dtype)
    ref = torch.zeros(6, 7, 7, 8); ref[..., :4] = w.permute(0, 2, 3, 1)
    close(ohwi, ref, dtype)
    # image packing NCHW fp32 -> NHWC with channels padded to 8 (the pixel-per-thread path), odd sizes; and a general permute (d3 != 8)
    img = rnd(3, 4, 35, 24, seed=19)
    nhwc = ops.permute4(img.cuda(), (3, 35, 24, 8), (4 * 35 * 24, 24, 1, 35 * 24), 4, dtype)
    ref = torch.zeros(3, 35, 24, 8); ref[..., :4] = img.permute(0, 2, 3, 1)
    close(nhwc, ref, dtype)
    gen = ops.permute4(img.cuda(), (3, 24, 35, 4), (4 * 35 * 24, 1, 24, 35 * 24), 4, dtype)
    close(gen, img.permute(0, 3, 2, 1), dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C,M", [(64, 1000), (256, 333), (2048, 70)])
def test_batchnorm(dtype, C, M):
    from ralf_amd import ops

    x = (rnd(M, C, seed=19, dtype=dtype).float() * 2 + 0.5).requires_grad_(True)
    res = rnd(M, C, seed=20, dtype=dtype).float().requires_grad_(True)
    g, b = (1 + 0.1 * rnd(C, seed=21)).requires_grad_(True), (0.1 * rnd(C, seed=22)).requires_grad_(True)
    rm, rv = 0.1 * rnd(C, seed=23), 1 + 0.1 * rnd(C, seed=24).abs()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = torch.relu(F.batch_norm(x, rm_ref, rv_ref, g, b, True, 0.1, 1e-5) + res)
    go = rnd(M, C, seed=25, dtype=dtype).float()
    y.backward(go)
    rmd, rvd = rm.cuda(), rv.cuda()
    xd, resd = x.detach().to(dtype).cuda(), res.detach().to(dtype).cuda()
    nbt = torch.tensor(3, dtype=torch.long, device="cuda")
    yd, mean, rstd = ops.bn_forward(xd, g.detach().cuda(), b.detach().cuda(), rmd, rvd, True, True, resd, counter=nbt)
    assert int(nbt) == 4  # num_batches_tracked is advanced by the statistics kernel
    close(yd, y.detach(), dtype)
    close(rmd, rm_ref, torch.float32, atol=1e-3); close(rvd, rv_ref, torch.float32, atol=2e-3, rtol=2e-3)
    dx, dg, db, dres = ops.bn_backward(xd, go.to(dtype).cuda(), yd, g.detach().cuda(), mean, rstd, True, True)
    btol = dict(atol=6e-2, rtol=6e-2) if dtype == torch.bfloat16 else dict(atol=2e-4, rtol=1e-3)
    fr = 0.999 if dtype == torch.bfloat16 else 1.0
    close(dx, x.grad, dtype, frac=fr, **btol); close(dres, res.grad, dtype, frac=fr)
    close(dg, g.grad, dtype, frac=0.99 if dtype == torch.bfloat16 else 1.0, atol=0.5 if dtype == torch.bfloat16 else 2e-3, rtol=5e-2)
    close(db, b.grad, dtype, frac=0.99 if dtype == torch.bfloat16 else 1.0, atol=0.5 if dtype == torch.bfloat16 else 2e-3, rtol=5e-2)
    # the 1-bit ReLU mask written by the forward replaces y in the backward: identical gradients, bit for bit
    y2, _, _, mask = ops.bn_forward(xd, g.detach().cuda(), b.detach().cuda(), rm.cuda(), rv.cuda(), True, True, resd, want_mask=True)
    assert torch.equal(y2, yd) and mask.numel() == M * C // 8
    bits = ((mask.view(-1, 1) >> torch.arange(8, device="cuda", dtype=torch.uint8)) & 1).view(M, C).bool()
    assert torch.equal(bits, yd > 0)
    dx2, dg2, db2, dres2 = ops.bn_backward(xd, go.to(dtype).cuda(), None, g.detach().cuda(), mean, rstd, True, True, mask=mask)
    assert torch.equal(dx2, dx) and torch.equal(dres2, dres)
    close(dg2, dg, torch.float32, atol=1e-4, rtol=1e-5); close(db2, db, torch.float32, atol=1e-4, rtol=1e-5)
    # eval mode uses the running statistics
    ye = F.batch_norm(x.detach(), rm_ref, rv_ref, g.detach(), b.detach(), False, 0.1, 1e-5)
    yed, _, _ = ops.bn_forward(xd, g.detach().cuda(), b.detach().cuda(), rm_ref.cuda(), rv_ref.cuda(), False, False, None)
    close(yed, ye, dtype)


def ref_attention(q, k, v, H, causal, kpm, scale):
    B, Sq, _ = q.shape
    Sk = k.shape[1]
    dh = q.shape[-1] // H
    qh, kh, vh = (t.view(B, -1, H, dh).transpose(1, 2) for t in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        s = s + torch.triu(torch.full((Sq, Sk), float("-inf")), 1)
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Sq, H * dh)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,Sq,Sk,dh,causal,pad", [(2, 8, 50, 50, 32, True, True), (2, 8, 70, 300, 32, False, False), (3, 4, 11, 11, 64, False, True),
                                                      (2, 8, 256, 16, 64, False, False), (1, 8, 130, 130, 32, True, False),
                                                      (2, 8, 51, 532, 32, False, False), (3, 8, 64, 300, 32, False, True), (2, 8, 17, 129, 32, False, False)])
def test_attention(dtype, B, H, Sq, Sk, dh, causal, pad):
    from ralf_amd import ops

    d = H * dh
    q, k, v = (rnd(B, S, d, seed=s, dtype=dtype).float().requires_grad_(True) for S, s in ((Sq, 30), (Sk, 31), (Sk, 32)))
    kpm = None
    if pad:
        kpm = torch.zeros(B, Sk, dtype=torch.bool); kpm[0, Sk // 2:] = True; kpm[-1, -1] = True
    scale = dh ** -0.5
    o = ref_attention(q, k, v, H, causal, kpm, scale)
    go = rnd(B, Sq, d, seed=33, dtype=dtype).float()
    o.backward(go)
    dev = lambda t: t.detach().to(dtype).cuda()
    kp = kpm.to(torch.uint8).cuda() if pad else None
    qd, kd, vd = dev(q), dev(k), dev(v)
    od, lse = ops.attention_fwd(qd, kd, vd, B, H, Sq, Sk, dh, causal=causal, kpm=kp)
    close(od, o.detach(), dtype)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.attention_bwd(dev(go), qd, kd, vd, od, lse, dq, dk, dv, B, H, Sq, Sk, dh, causal=causal, kpm=kp)
    tol = dict(atol=5e-2, rtol=5e-2) if dtype == torch.bfloat16 else dict(atol=5e-5, rtol=1e-4)
    close(dq, q.grad, dtype, **tol); close(dk, k.grad, dtype, **tol); close(dv, v.grad, dtype, **tol)


@pytest.mark.parametrize("Sq,Sk", [(256, 256), (130, 300), (51, 532)])
def test_attention_backward_padded_key_with_a_huge_score(Sq, Sk):
    """a PADDED key whose score exceeds the row's log-sum-exp by more than 128 in the log2 domain makes exp2 overflow; its probability must be an
    exact zero in every backward kernel (0 * inf would be NaN in dQ / dK / dV of the whole head): the one-pass kernel (S = 256: the image
    encoder's shape), the per-query + per-key pair (130 x 300) and the one-pass cross-attention kernel (51 x 532) -- all through the select in
    attn::bwd_step_keys / bwd_step_queries (csrc/attn_core.h)"""
    from ralf_amd import ops

    B, H, dh = 2, 8, 32
    d = H * dh
    q, k, v = (rnd(B, S, d, seed=s, dtype=torch.bfloat16).float() for S, s in ((Sq, 60), (Sk, 61), (Sk, 62)))
    k[0, 7] = q[0].mean(0) * 0 + 60.0 * torch.sign(q[0, 3])     # key 7 of sample 0: score ~ 60 * |q|_1 * dh^-0.5 >> every other score
    q, k, v = (t.requires_grad_(True) for t in (q, k, v))
    kpm = torch.zeros(B, Sk, dtype=torch.bool); kpm[0, 7] = True; kpm[1, Sk - 56:] = True
    o = ref_attention(q, k, v, H, False, kpm, dh ** -0.5)
    go = rnd(B, Sq, d, seed=63, dtype=torch.bfloat16).float()
    o.backward(go)
    dev = lambda t: t.detach().to(torch.bfloat16).cuda()
    qd, kd, vd = dev(q), dev(k), dev(v)
    kp = kpm.to(torch.uint8).cuda()
    od, lse = ops.attention_fwd(qd, kd, vd, B, H, Sq, Sk, dh, kpm=kp)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.attention_bwd(dev(go), qd, kd, vd, od, lse, dq, dk, dv, B, H, Sq, Sk, dh, kpm=kp)
    for t in (od, dq, dk, dv):
        assert torch.isfinite(t.float()).all()
    tol = dict(atol=5e-2, rtol=5e-2)
    close(od, o.detach(), torch.bfloat16)
    close(dq, q.grad, torch.bfloat16, **tol); close(dk, k.grad, torch.bfloat16, **tol); close(dv, v.grad, torch.bfloat16, **tol)


@pytest.mark.parametrize("Sq,Sk", [(256, 256), (50, 300)])
def test_attention_with_every_key_of_a_sample_masked_is_defined(Sq, Sk):
    """a sample whose keys are ALL padded (an empty constraint sequence): l == 0 in the forward's epilogue.  attention_mfma.hip / tlayer.hip are
    built with -fno-honor-nans, so the kernel defines the case instead of leaving 0 * inf to the flag: output 0, lse -inf, zero gradients for
    that sample; the other sample is untouched"""
    from ralf_amd import ops

    B, H, dh = 2, 8, 32
    d = H * dh
    q, k, v = (rnd(B, S, d, seed=s, dtype=torch.bfloat16).float() for S, s in ((Sq, 70), (Sk, 71), (Sk, 72)))
    kpm = torch.zeros(B, Sk, dtype=torch.bool); kpm[0, :] = True
    dev = lambda t: t.detach().to(torch.bfloat16).cuda()
    qd, kd, vd = dev(q), dev(k), dev(v)
    od, lse = ops.attention_fwd(qd, kd, vd, B, H, Sq, Sk, dh, kpm=kpm.to(torch.uint8).cuda())
    assert torch.equal(od[0].float(), torch.zeros_like(od[0].float())) and bool(torch.isinf(lse.view(B, -1)[0]).all()) and bool((lse.view(B, -1)[0] < 0).all())
    want = ref_attention(q[1:], k[1:], v[1:], H, False, None, dh ** -0.5)
    close(od[1:], want, torch.bfloat16)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    go = dev(rnd(B, Sq, d, seed=73, dtype=torch.bfloat16))
    ops.attention_bwd(go, qd, kd, vd, od, lse, dq, dk, dv, B, H, Sq, Sk, dh, kpm=kpm.to(torch.uint8).cuda())
    for t in (dq, dk, dv):
        assert torch.isfinite(t.float()).all() and float(t[0].float().abs().max()) == 0.0


def test_attention_packed_qkv_and_dropout():
    """self-attention on a packed [B,S,3d] buffer; dropout backward consistent with forward (finite differences)."""
    from ralf_amd import ops

    B, H, S, dh = 2, 8, 40, 32
    d = H * dh
    qkv = rnd(B, S, 3 * d, seed=40).cuda()
    o1, _ = ops.attention_fwd(qkv, qkv, qkv, B, H, S, S, dh, q_off=0, k_off=d, v_off=2 * d)
    q, k, v = qkv[..., :d].contiguous(), qkv[..., d:2 * d].contiguous(), qkv[..., 2 * d:].contiguous()
    o2, _ = ops.attention_fwd(q, k, v, B, H, S, S, dh)
    assert torch.equal(o1, o2)
    seed = torch.tensor([99], dtype=torch.int64, device="cuda")
    od, lse = ops.attention_fwd(q, k, v, B, H, S, S, dh, p_drop=0.3, seed=seed, call_id=5)
    assert not torch.equal(od, o2)
    go = rnd(B, S, d, seed=41).cuda()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attention_bwd(go, q, k, v, od, lse, dq, dk, dv, B, H, S, S, dh, p_drop=0.3, seed=seed, call_id=5)
    # directional derivative check of L = <go, O(v + t*dv_dir)> (O is linear in V for a fixed mask)
    dirv = rnd(B, S, d, seed=42).cuda()
    o_plus, _ = ops.attention_fwd(q, k, v + dirv, B, H, S, S, dh, p_drop=0.3, seed=seed, call_id=5)
    lhs = ((o_plus - od) * go).sum().item()
    rhs = (dv * dirv).sum().item()
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))
    eps = 1e-2
    dirq = rnd(B, S, d, seed=43).cuda()
    op, _ = ops.attention_fwd(q + eps * dirq, k, v, B, H, S, S, dh, p_drop=0.3, seed=seed, call_id=5)
    om, _ = ops.attention_fwd(q - eps * dirq, k, v, B, H, S, S, dh, p_drop=0.3, seed=seed, call_id=5)
    lhs = (((op - om) * go).sum() / (2 * eps)).item()
    rhs = (dq * dirq).sum().item()
    assert abs(lhs - rhs) < 2e-2 * max(1.0, abs(lhs))


def test_adamw_and_clip():
    from ralf_amd import ops

    n = 10007
    p0, g = rnd(n, seed=50), rnd(n, seed=51)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p_ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    pd, m, v = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    shadow = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    for step in range(1, 4):
        gs = g * step
        p_ref.grad = gs.clone()
        total = torch.nn.utils.clip_grad_norm_([p_ref], 0.1)
        opt.step()
        ss, coef, nrm = torch.zeros(1, device="cuda"), torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
        ops.sumsq(gs.cuda(), ss); ops.clip_coef(ss, 0.1, coef, nrm)
        torch.testing.assert_close(nrm.cpu()[0], total, rtol=1e-5, atol=1e-6)
        # the deterministic pair the train step uses: same norm, and bit-identical from call to call
        parts, c2, n2 = torch.empty(ops.SUMSQ_PARTS, device="cuda"), torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
        ops.sumsq_partials(gs.cuda(), parts); ops.clip_coef_partials(parts, 0.1, c2, n2)
        torch.testing.assert_close(n2.cpu()[0], total, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(c2, coef, rtol=1e-5, atol=0)
        big = rnd(3_000_001, seed=52).cuda()
        outs = []
        for _ in range(3):
            ops.sumsq_partials(big, parts); ops.clip_coef_partials(parts, 0.1, c2, n2)
            outs.append((float(c2), float(n2)))
        assert outs[0] == outs[1] == outs[2]
        if step % 2:   # bias corrections from the device-resident step counter (graph-replay path)
            sd = torch.tensor([step], dtype=torch.int32, device="cuda")
            ops.adamw(pd, gs.cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1, coef, shadow, sd)
        else:          # ... or from the host step argument
            ops.adamw(pd, gs.cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step, coef, shadow)
        torch.testing.assert_close(pd.cpu(), p_ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(shadow.float().cpu(), pd.cpu(), rtol=1e-2, atol=1e-2)


def test_mask_sample():
    from ralf_amd import ops

    g = torch.Generator().manual_seed(60)
    B, V = 300, 518
    logits = torch.randn(B, V, generator=g) * 3
    allowed = torch.rand(V, generator=g) > 0.3
    forced = torch.full((B,), -1, dtype=torch.int64); forced[::7] = 517; forced[3] = 5
    ref = logits.masked_fill(~allowed, float("-inf")).argmax(1)
    ref = torch.where(forced >= 0, forced, ref)
    out = ops.mask_sample(logits.cuda(), allowed.to(torch.uint8).cuda(), forced.cuda(), mode=0)
    assert torch.equal(out.cpu(), ref)
    # ties -> first maximum, like torch.argmax
    t = torch.zeros(4, V); t[:, 100] = 1.0; t[:, 300] = 1.0
    assert ops.mask_sample(t.cuda()).tolist() == [100] * 4
    # top-k multinomial: draws stay inside the top-k set and follow softmax(x/T) of it
    seed = torch.tensor([7], dtype=torch.int64, device="cuda")
    row = torch.randn(V, generator=g)
    x = row.repeat(20000, 1).cuda()
    draws = ops.mask_sample(x, allowed.to(torch.uint8).cuda(), None, mode=1, top_k=5, temperature=0.7, seed=seed, call_id=3).cpu()
    masked = row.masked_fill(~allowed, float("-inf"))
    top = masked.topk(5)
    assert set(draws.tolist()) <= set(top.indices.tolist())
    want = torch.softmax(top.values / 0.7, 0)
    got = torch.stack([(draws == i).float().mean() for i in top.indices])
    assert (got - want).abs().max() < 0.015, (got, want)
    d2 = ops.mask_sample(x, allowed.to(torch.uint8).cuda(), None, mode=1, top_k=5, temperature=0.7, seed=seed, call_id=3).cpu()
    assert torch.equal(draws, d2)   # counter-based: same seed/call -> same draws
    # decode-loop form: the token also lands in a column of the sequence buffer, its pad flag in a column of the mask buffer
    seqbuf = torch.full((B, 9), -7, dtype=torch.int64, device="cuda")
    padbuf = torch.full((B, 9), 9, dtype=torch.uint8, device="cuda")
    out2 = ops.mask_sample(logits.cuda(), allowed.to(torch.uint8).cuda(), forced.cuda(), mode=0, seq_col=seqbuf[:, 4], pad_flag_col=padbuf[:, 6], pad_id=517)
    assert torch.equal(out2.cpu(), ref) and torch.equal(seqbuf[:, 4].cpu(), ref) and torch.equal(padbuf[:, 6].cpu(), (ref == 517).to(torch.uint8))
    assert bool((seqbuf[:, [0, 1, 2, 3, 5, 6, 7, 8]] == -7).all()) and bool((padbuf[:, [0, 1, 2, 3, 4, 5, 7, 8]] == 9).all())
    d3 = ops.mask_sample(x, allowed.to(torch.uint8).cuda(), None, mode=1, top_k=5, temperature=0.7, seed=seed, call_id=3,
                         seq_col=torch.empty(20000, 2, dtype=torch.int64, device="cuda")[:, 1]).cpu()
    assert torch.equal(draws, d3)


def test_mask_sample_top_p_random_gumbel():
    """ralf_mask_sample modes 2-4 against image2layout/train/helpers/sampling.py:18-71 (ralf_amd.helpers.sampling.sample restates it with torch
    ops): the nucleus is the reference's kept SET (sort + inclusive cumulative sum; here a bisection, no sort), the draws follow the
    renormalised softmax; `random` and `gumbel` follow softmax(x / T) (a Gumbel-perturbed softmax is checked through its argmax-free mean)"""
    import torch.nn.functional as F

    from ralf_amd import ops

    g = torch.Generator().manual_seed(61)
    V, n = 518, 40000
    allowed = torch.rand(V, generator=g) > 0.3
    seed = torch.tensor([11], dtype=torch.int64, device="cuda")
    al = allowed.to(torch.uint8).cuda()
    for T, top_p, spread in ((1.0, 0.9, 1.0), (0.7, 0.5, 2.0), (1.3, 0.999, 0.5), (1.0, 0.05, 3.0)):
        row = torch.randn(V, generator=g) * spread
        masked = row.masked_fill(~allowed, float("-inf"))
        scaled = masked / T
        srt, order = torch.sort(scaled, descending=True)
        cum = torch.cumsum(F.softmax(srt, 0), 0)
        keep_sorted = ~((cum > top_p) & (torch.arange(V) > 0))
        kept = set(order[keep_sorted & (srt > float("-inf"))].tolist())
        x = row.repeat(n, 1).cuda()
        draws = ops.mask_sample(x, al, None, mode=ops.SAMPLING_MODES["top_p"], temperature=T, top_p=top_p, seed=seed, call_id=5).cpu()
        got = set(draws.tolist())
        pk = F.softmax(scaled, 0)
        # the kept set: nothing outside it is ever drawn; everything in it that is likely enough to show up in n draws does
        boundary = {int(order[i]) for i in range(V) if abs(float(cum[i]) - top_p) < 1e-5}       # (an element ON the threshold may round either way)
        assert got <= kept | boundary, (T, top_p, sorted(got - kept))
        likely = {i for i in kept if float(pk[i]) * n / sum(float(pk[j]) for j in kept) > 30}
        assert likely <= got | boundary, (T, top_p)
        want = torch.zeros(V); idx = torch.tensor(sorted(kept)); want[idx] = pk[idx] / pk[idx].sum()
        freq = torch.bincount(draws, minlength=V).float() / n
        assert (freq - want).abs().max() < 0.012, (T, top_p, (freq - want).abs().max())
    row = torch.randn(V, generator=g) * 2
    masked = row.masked_fill(~allowed, float("-inf"))
    x = row.repeat(n, 1).cuda()
    draws = ops.mask_sample(x, al, None, mode=ops.SAMPLING_MODES["random"], temperature=0.8, seed=seed, call_id=6).cpu()
    freq = torch.bincount(draws, minlength=V).float() / n
    assert (freq - F.softmax(masked / 0.8, 0)).abs().max() < 0.012 and not bool(freq[~allowed].any())
    # gumbel: softmax(x / T + G) drawn once per row = a draw from a RANDOM distribution; its mean over rows is compared with a torch simulation
    draws = ops.mask_sample(x, al, None, mode=ops.SAMPLING_MODES["gumbel"], temperature=0.8, seed=seed, call_id=7).cpu()
    freq = torch.bincount(draws, minlength=V).float() / n
    u = torch.rand(4000, V, generator=g)
    sim = F.softmax(masked / 0.8 - torch.log(-torch.log(u + 1e-30) + 1e-30), 1).mean(0)
    assert (freq - sim).abs().max() < 0.02 and not bool(freq[~allowed].any())
    d2 = ops.mask_sample(x, al, None, mode=ops.SAMPLING_MODES["gumbel"], temperature=0.8, seed=seed, call_id=7).cpu()
    assert torch.equal(draws, d2)


@pytest.mark.parametrize("B,Sk,pad", [(3, 1, False), (5, 51, True), (4, 532, False), (2, 700, True)])
def test_attention_decode_step(B, Sk, pad):
    """Sq = 1 streaming kernel (bf16, need_lse=False) on the KV-cache layout of nn.decoder_step: packed [K|V] rows,
    cache longer than the valid prefix (kv_rows), key-padding mask."""
    from ralf_amd import ops

    H, dh = 8, 32
    d = H * dh
    L = Sk + 7                                       # cache rows beyond the prefix hold garbage that must not be read
    q = rnd(B, 1, d, seed=50, dtype=torch.bfloat16)
    kv = rnd(B, L, 2 * d, seed=51, dtype=torch.bfloat16)
    kv[:, Sk:] = float("nan")
    kpm = None
    if pad:
        kpm = torch.zeros(B, Sk, dtype=torch.bool); kpm[0, Sk // 2:] = True; kpm[-1, 0] = True
    ref = ref_attention(q.float(), kv[:, :Sk, :d].float(), kv[:, :Sk, d:].float(), H, False, kpm, dh ** -0.5)
    kp = kpm.to(torch.uint8).cuda() if pad else None
    o, lse = ops.attention_fwd(q.cuda(), kv.cuda(), kv.cuda(), B, H, 1, Sk, dh, 0, 0, d, causal=False, kpm=kp, need_lse=False, kv_rows=L)
    assert lse is None
    close(o, ref, torch.bfloat16)
    o2, _ = ops.attention_fwd(q.cuda(), kv.cuda(), kv.cuda(), B, H, 1, Sk, dh, 0, 0, d, causal=False, kpm=kp, need_lse=True, kv_rows=L)   # tiled kernel
    close(o, o2.float().cpu(), torch.bfloat16)
    if pad:   # the decode loop's form of the mask: ONE [B, max_len] buffer, rows longer than the prefix (RalfAttnDesc.kpm_bs)
        wide = torch.ones(B, Sk + 5, dtype=torch.uint8)
        wide[:, :Sk] = kpm.to(torch.uint8)
        for need_lse in (False, True):
            o3, _ = ops.attention_fwd(q.cuda(), kv.cuda(), kv.cuda(), B, H, 1, Sk, dh, 0, 0, d, causal=False, kpm=wide.cuda(), need_lse=need_lse, kv_rows=L, kpm_stride=Sk + 5)
            assert torch.equal(o3, o if not need_lse else o2)


@pytest.mark.parametrize("B,H,Sk,pad", [(3, 8, 1, False), (5, 8, 51, True), (4, 8, 532, False), (2, 4, 700, True), (3, 8, 3000, True)])
def test_attention_decode_step_fp32(B, H, Sk, pad):
    """Sq = 1 in the fp32 parity mode (attn_decode_f32_kernel: a workgroup per batch element, lanes across the 256-wide row): against
    torch at 1e-5, and against the general one-lane-per-query kernel (need_lse=True keeps that one)."""
    from ralf_amd import ops

    dh = 256 // H
    d = H * dh
    L = Sk + 7
    q = rnd(B, 1, d, seed=52)
    kv = rnd(B, L, 2 * d, seed=53)
    kv[:, Sk:] = float("nan")
    kpm = None
    if pad:
        kpm = torch.zeros(B, Sk, dtype=torch.bool); kpm[0, Sk // 2:] = True; kpm[-1, 0] = True
    ref = ref_attention(q, kv[:, :Sk, :d], kv[:, :Sk, d:], H, False, kpm, dh ** -0.5)
    kp = kpm.to(torch.uint8).cuda() if pad else None
    o, lse = ops.attention_fwd(q.cuda(), kv.cuda(), kv.cuda(), B, H, 1, Sk, dh, 0, 0, d, causal=False, kpm=kp, need_lse=False, kv_rows=L)
    assert lse is None and bool(torch.isfinite(o).all())
    assert float((o.cpu() - ref).abs().max()) < 2e-5
    o2, _ = ops.attention_fwd(q.cuda(), kv.cuda(), kv.cuda(), B, H, 1, Sk, dh, 0, 0, d, causal=False, kpm=kp, need_lse=True, kv_rows=L)
    assert float((o - o2).abs().max()) < 2e-5
    if pad:
        wide = torch.ones(B, Sk + 5, dtype=torch.uint8)
        wide[:, :Sk] = kpm.to(torch.uint8)
        o3, _ = ops.attention_fwd(q.cuda(), kv.cuda(), kv.cuda(), B, H, 1, Sk, dh, 0, 0, d, causal=False, kpm=wide.cuda(), need_lse=False, kv_rows=L, kpm_stride=Sk + 5)
        assert torch.equal(o3, o)


@pytest.mark.parametrize("B,pos,pad", [(3, 0, False), (5, 1, False), (6, 37, True), (9, 70, True), (4, 159, False)])
def test_decode_attn_self_block(B, pos, pad):
    """ralf_decode_attn, self-attention form: LayerNorm + q/k/v projections + cache append + attention over pos+1 keys in one launch,
    against the same arithmetic in torch fp32 with the kernel's rounding points (h, q/k/v, o in bf16); B not a multiple of the 4
    elements a workgroup shares, more than 64 cached keys, masked keys"""
    from ralf_amd import ops

    d, H, L = 256, 8, 200
    dh = d // H
    x = rnd(B, d, seed=60, dtype=torch.bfloat16)
    g, be = rnd(d, seed=61) * 0.2 + 1.0, rnd(d, seed=62) * 0.1
    W = rnd(3 * d, d, seed=63, scale=d ** -0.5).bfloat16()
    bias = rnd(3 * d, seed=64) * 0.1
    kv = rnd(B, L, 2 * d, seed=65, dtype=torch.bfloat16)
    kv[:, pos:] = float("nan")                      # rows the kernel must not read (row `pos` is written by it)
    kpm = torch.zeros(B, L, dtype=torch.uint8)
    if pad:
        kpm[0, 1:pos:3] = 1; kpm[-1, pos // 2] = 1
    h = F.layer_norm(x.float(), (d,), g, be, 1e-5).bfloat16().float()
    qkv = (h @ W.float().t() + bias).bfloat16().float()
    q, k, v = (t.contiguous() for t in (qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]))
    K = torch.cat([kv[:, :pos, :d].float(), k[:, None]], 1)
    V = torch.cat([kv[:, :pos, d:].float(), v[:, None]], 1)
    ref = ref_attention(q[:, None].contiguous(), K, V, H, False, kpm[:, :pos + 1].bool() if pad else None, dh ** -0.5)[:, 0]
    kvd = kv.cuda()
    o = ops.decode_attn(x.cuda(), g.cuda(), be.cuda(), W.cuda(), bias.cuda(), kvd, pos, H, True, kpm=kpm.cuda(), kpm_stride=L)
    close(o, ref, torch.bfloat16)
    torch.testing.assert_close(kvd[:, pos, :d].float().cpu(), k.bfloat16().float(), atol=2e-2, rtol=2e-2)     # the appended row
    torch.testing.assert_close(kvd[:, pos, d:].float().cpu(), v.bfloat16().float(), atol=2e-2, rtol=2e-2)
    assert bool(torch.isnan(kvd[:, pos + 1:].float()).all())                                                    # nothing else was touched


@pytest.mark.parametrize("B,M", [(3, 1), (5, 130), (2, 578)])
def test_decode_attn_cross_block(B, M):
    """ralf_decode_attn, cross-attention form (q projection only, K/V of the memory precomputed) against torch fp32"""
    from ralf_amd import ops

    d, H = 256, 8
    x = rnd(B, d, seed=70, dtype=torch.bfloat16)
    g, be = rnd(d, seed=71) * 0.2 + 1.0, rnd(d, seed=72) * 0.1
    W = rnd(3 * d, d, seed=73, scale=d ** -0.5).bfloat16()
    bias = rnd(3 * d, seed=74) * 0.1
    kv = rnd(B, M, 2 * d, seed=75, dtype=torch.bfloat16)
    h = F.layer_norm(x.float(), (d,), g, be, 1e-5).bfloat16().float()
    q = (h @ W[:d].float().t() + bias[:d]).bfloat16().float()
    ref = ref_attention(q[:, None].contiguous(), kv[:, :, :d].float().contiguous(), kv[:, :, d:].float().contiguous(), H, False, None, (d // H) ** -0.5)[:, 0]
    o = ops.decode_attn(x.cuda(), g.cuda(), be.cuda(), W.cuda(), bias.cuda(), kv.cuda(), M, H, False)
    close(o, ref, torch.bfloat16)


@pytest.mark.parametrize("dtype", DT)
def test_conv_weight_relayout_and_folded_batchnorm_epilogue(dtype):
    """ralf_conv_relayout_batched (both GEMM layouts of k x k weights from one read, incl. the zero-padded 4-channel stem and a 7x7
    kernel) and the inference epilogue colscale / bias / ReLU-after-residual of ralf_gemm with ralf_bn_fold_batched's scale / shift"""
    from ralf_amd import ops

    ws = [rnd(16, 4, 7, 7, seed=80), rnd(24, 72, 3, 3, seed=81), rnd(128, 64, 3, 3, seed=82), rnd(8, 200, 3, 3, seed=83)]
    jobs, outs = [], []
    for w in ws:
        Co, Ci, kh, kw = w.shape
        cip = (Ci + 7) // 8 * 8
        o1 = torch.full((Co, kh, kw, cip), float("nan"), dtype=dtype, device="cuda")
        o2 = torch.full((Ci, kh, kw, Co), float("nan"), dtype=dtype, device="cuda")
        jobs.append((w.cuda(), o1, o2)); outs.append((o1, o2))
    table, n, blocks = ops.conv_relayout_table(jobs, torch.device("cuda", 0))
    ops.conv_relayout_batched(table, n, blocks)
    for w, (o1, o2) in zip(ws, outs):
        Ci = w.shape[1]
        ref1 = torch.zeros(o1.shape)
        ref1[..., :Ci] = w.permute(0, 2, 3, 1)
        assert torch.equal(o1.float().cpu(), ref1.to(dtype).float())
        assert torch.equal(o2.float().cpu(), w.permute(1, 2, 3, 0).to(dtype).float())
    # folded BatchNorm: y = relu(x W^T * scale + shift + res)
    M, K, N = 200, 64, 128
    bns = [(rnd(N, seed=84) * 0.2 + 1.0, rnd(N, seed=85) * 0.1, rnd(N, seed=86) * 0.3, torch.rand(N, generator=torch.Generator().manual_seed(87)) + 0.5)]
    buf = torch.empty(2 * N, dtype=torch.float32, device="cuda")
    tb, views = ops.bn_fold_table([tuple(t.cuda() for t in bns[0])], buf, torch.device("cuda", 0))
    ops.bn_fold_batched(tb, 1, 1e-5)
    gam, bet, mu, var = bns[0]
    sc = gam / torch.sqrt(var + 1e-5)
    torch.testing.assert_close(views[0][0].cpu(), sc, atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(views[0][1].cpu(), bet - mu * sc, atol=1e-6, rtol=1e-6)
    x, W, res = rnd(M, K, seed=88).to(dtype), rnd(N, K, seed=89, scale=K ** -0.5).to(dtype), rnd(M, N, seed=90).to(dtype)
    y = ops.gemm(x.cuda(), W.cuda(), M, N, K, colscale=views[0][0], bias=views[0][1], act="relu_post", res=res.cuda())
    ref = torch.relu(x.float() @ W.float().t() * sc + (bet - mu * sc) + res.float())
    close(y, ref, dtype)
    y2 = ops.gemm(x.cuda(), W.cuda(), M, N, K, colscale=views[0][0], bias=views[0][1], act="relu")
    close(y2, torch.relu(x.float() @ W.float().t() * sc + (bet - mu * sc)), dtype)


def test_library_streams_are_not_pool_streams():
    """ops.own_stream: created by ralf_stream_create, stable per slot, distinct from every stream of torch's pool"""
    from ralf_amd import ops

    a, b = ops.own_stream("t1"), ops.own_stream("t2")
    assert a.cuda_stream != b.cuda_stream and ops.own_stream("t1").cuda_stream == a.cuda_stream
    pool = {torch.cuda.Stream().cuda_stream for _ in range(40)}
    assert a.cuda_stream not in pool and b.cuda_stream not in pool
    x = torch.ones(1 << 20, device="cuda")
    with torch.cuda.stream(a):
        y = x * 2
    a.synchronize()
    assert float(y.sum()) == 2 * (1 << 20)


def test_fanout_alias_with_an_extra_consumer_of_the_shared_input():
    """Runtime.fanout_alias: the linear layers that read the alias sum their data gradients in the GEMM epilogue (one buffer, in place);
    another consumer of the ORIGINAL tensor whose backward runs between them must still be summed in (the alias is an autograd node
    of its own).  Reference: the same graph on plain torch ops."""
    from ralf_amd import functional as RF
    from ralf_amd.functional import Runtime

    g = torch.Generator(device="cuda").manual_seed(5)
    rt = Runtime(torch.float32).to(torch.device("cuda"))
    x = torch.randn(192, 64, device="cuda", generator=g, requires_grad=True)
    Ws = [torch.randn(128, 64, device="cuda", generator=g, requires_grad=True) for _ in range(3)]
    c = torch.randn(192, 64, device="cuda", generator=g)
    up = [torch.randn(192, 128, device="cuda", generator=g) for _ in range(3)]

    def graph(lin, shared):
        y0 = (lin(shared, Ws[0]) * up[0]).sum()
        extra = (x * c).sum()                       # a consumer of x itself, recorded between the projections
        y1 = (lin(shared, Ws[1]) * up[1]).sum()
        extra2 = (x.sin() * c).sum()
        y2 = (lin(shared, Ws[2]) * up[2]).sum()
        return y0 + extra + y1 + extra2 + y2
    rt.begin_step()
    a = rt.fanout_alias(x)
    assert a.data_ptr() == x.data_ptr() and a is not x
    graph(lambda t, W: RF.linear(t, W, rt=rt), a).backward()
    got = [x.grad.clone()] + [W.grad.clone() for W in Ws]
    x.grad = None
    for W in Ws:
        W.grad = None
    graph(lambda t, W: t @ W.t(), x).backward()
    want = [x.grad] + [W.grad for W in Ws]
    for u, v in zip(got, want):
        torch.testing.assert_close(u, v, rtol=1e-4, atol=1e-4)


_ATTN_BWD_CASES = """
import sys
sys.path.insert(0, {root!r})
import torch
from ralf_amd import ops
g = torch.Generator(device="cuda").manual_seed(11)
out = {{}}
# (B, H, Sq, Sk, packed self-attention?, causal, padded keys, p)
for B, H, Sq, Sk, packed, causal, pad, p in [(4, 8, 256, 256, True, False, False, 0.1), (3, 8, 50, 50, True, True, True, 0.1), (2, 8, 50, 200, False, False, False, 0.0),
                                             (2, 8, 256, 77, False, False, True, 0.1), (5, 8, 1, 33, False, False, False, 0.0), (2, 8, 33, 256, False, True, False, 0.1),
                                             (2, 8, 40, 300, False, False, False, 0.1)]:
    dh = 32
    seed = torch.tensor([99], dtype=torch.int64, device="cuda")
    if packed:
        qkv = torch.randn(B, Sq, 3 * H * dh, device="cuda", generator=g).bfloat16()
        q = k = v = qkv
        offs = (0, H * dh, 2 * H * dh)
    else:
        q = torch.randn(B, Sq, H * dh, device="cuda", generator=g).bfloat16()
        k = v = torch.randn(B, Sk, 2 * H * dh, device="cuda", generator=g).bfloat16()
        offs = (0, 0, H * dh)
    kpm = None
    if pad:
        kpm = torch.zeros(B, Sk, dtype=torch.uint8, device="cuda")
        kpm[:, Sk - 5:] = 1
        kpm[0, 1] = 1
    o, lse = ops.attention_fwd(q, k, v, B, H, Sq, Sk, dh, *offs, causal=causal, kpm=kpm, p_drop=p, seed=seed, call_id=5)
    do = torch.randn(B, Sq, H * dh, device="cuda", generator=g).bfloat16()
    if packed:
        da = torch.zeros_like(qkv)
        ops.attention_bwd(do, q, k, v, o, lse, da, da, da, B, H, Sq, Sk, dh, *offs, *offs, causal=causal, kpm=kpm, p_drop=p, seed=seed, call_id=5)
        out[(B, Sq, Sk, causal, pad, p, "dqkv")] = da
    else:
        dq, dkv = torch.zeros_like(q), torch.zeros_like(k)
        ops.attention_bwd(do, q, k, v, o, lse, dq, dkv, dkv, B, H, Sq, Sk, dh, *offs, *offs, causal=causal, kpm=kpm, p_drop=p, seed=seed, call_id=5)
        out[(B, Sq, Sk, causal, pad, p, "dq")] = dq
        out[(B, Sq, Sk, causal, pad, p, "dkv")] = dkv
torch.cuda.synchronize()
torch.save({{k: v.cpu() for k, v in out.items()}}, {path!r})
"""


def test_one_pass_attention_backward_equals_the_two_kernel_backward(tmp_path):
    """sequences of up to 256 queries / keys (dh = 32) can take dQ, dK, dV from ONE pass over the scores (attn_bwd_fused16_mfma, or the 8-wave
    attn_bwd_fused_mfma with RALF_ATTN_BWD_FUSED16=0: the default from 128 x 128 on, everywhere it fits with RALF_ATTN_BWD_FUSED=2 as
    here); RALF_ATTN_BWD_FUSED=0 (read once per process) keeps the per-query + per-key kernels.  Same P, dropout mask and dS arithmetic;
    delta is summed in another order and the 8-wave form adds eight fp32 partial dQ, so results may differ in the last bf16 bit.
    The last case (Sk = 300) runs the two-kernel path in both processes."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flag, w16 in (("2", "1"), ("2", "0"), ("0", "1")):
        path = str(tmp_path / f"attn_bwd{flag}{w16}.pt")
        r = subprocess.run([sys.executable, "-c", _ATTN_BWD_CASES.format(root=root, path=path)],
                           env=dict(os.environ, RALF_ATTN_BWD_FUSED=flag, RALF_ATTN_BWD_FUSED16=w16, RALF_ATTN_BWD_CROSS="0"), capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res[flag + w16] = torch.load(path)
    assert set(res["21"]) == set(res["01"]) == set(res["20"]) and len(res["21"]) == 12
    for k, form in [(k, form) for k in res["21"] for form in ("21", "20")]:
        a, b = res[form][k].float(), res["01"][k].float()
        assert torch.isfinite(a).all(), k
        tol = 2.0 ** -7 * b.abs().max().item()   # one bf16 ulp at the tensor's scale
        bad = ((a - b).abs() > tol).float().mean().item()
        assert (a - b).abs().max().item() <= 4 * tol and bad < 1e-3, (k, (a - b).abs().max().item(), tol, bad)
        assert b.abs().max().item() > 1e-3, k


_ATTN_CROSS_CASES = _ATTN_BWD_CASES.replace("""for B, H, Sq, Sk, packed, causal, pad, p in [(4, 8, 256, 256, True, False, False, 0.1), (3, 8, 50, 50, True, True, True, 0.1), (2, 8, 50, 200, False, False, False, 0.0),
                                             (2, 8, 256, 77, False, False, True, 0.1), (5, 8, 1, 33, False, False, False, 0.0), (2, 8, 33, 256, False, True, False, 0.1),
                                             (2, 8, 40, 300, False, False, False, 0.1)]:""", """for B, H, Sq, Sk, packed, causal, pad, p in [(64, 8, 51, 532, False, False, False, 0.1), (3, 8, 64, 300, False, False, True, 0.1), (2, 8, 20, 129, False, False, False, 0.0),
                                             (2, 8, 33, 40, False, False, True, 0.1), (1, 8, 64, 1000, False, False, False, 0.1), (2, 8, 5, 31, False, False, False, 0.0),
                                             (2, 8, 50, 50, True, False, True, 0.1)]:""")
assert _ATTN_CROSS_CASES != _ATTN_BWD_CASES


def test_cross_attention_backward_in_one_pass_equals_the_two_kernel_backward(tmp_path):
    """the decoder's cross-attention (<= 64 queries over a long memory, dh = 32, not causal) takes dQ, dK, dV from ONE pass (attn_bwd_cross_mfma:
    one workgroup per (batch, head), each wave owns 32-key blocks; the default from 128 keys on, everywhere it fits with RALF_ATTN_BWD_CROSS=2 as
    here); RALF_ATTN_BWD_CROSS=0 keeps the per-query + per-key kernels.  Same P, dropout mask and dS arithmetic; dQ sums per-wave partial
    products in a fixed order instead of one accumulator chain: last-bit differences at most."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flag in ("2", "0"):
        path = str(tmp_path / f"attn_cross{flag}.pt")
        r = subprocess.run([sys.executable, "-c", _ATTN_CROSS_CASES.format(root=root, path=path)],
                           env=dict(os.environ, RALF_ATTN_BWD_CROSS=flag, RALF_ATTN_BWD_FUSED="0"), capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res[flag] = torch.load(path)
    assert set(res["2"]) == set(res["0"]) and len(res["2"]) == 13
    for k in res["2"]:
        a, b = res["2"][k].float(), res["0"][k].float()
        assert torch.isfinite(a).all(), k
        assert b.abs().max().item() > 1e-3, k
        tol = 2.0 ** -7 * b.abs().max().item()   # one bf16 ulp at the tensor's scale
        bad = ((a - b).abs() > tol).float().mean().item()
        assert (a - b).abs().max().item() <= 4 * tol and bad < 1e-3, (k, (a - b).abs().max().item(), tol, bad)


@pytest.mark.parametrize("dtype", DT)
def test_stem_bn_relu_maxpool_in_one_pass_equals_the_three_kernels(dtype):
    """ralf_bn_relu_maxpool_fwd / _bwd_reduce / _bwd_apply (the stem: conv1 -> bn1 -> relu -> maxpool, common/image.py:39-48) against
    ralf_bn_apply -> ralf_maxpool3x3s2 and their backward: pooled values and argmax bit for bit (odd sizes, negative gammas, dead channels),
    gradients to rounding, and against torch's own batch_norm / relu / max_pool2d"""
    from ralf_amd import ops

    B, H, W, C = 3, 37, 50, 64
    y = rnd(B, H, W, C, seed=70, dtype=dtype).cuda()
    gamma = (0.5 + torch.rand(C, generator=torch.Generator().manual_seed(71)))
    gamma[::7] *= -1.0                       # negative scales: relu(BN(.)) is not monotone in y, every tap must be normalised before the comparison
    beta = 0.3 * rnd(C, seed=72)
    beta[5] = -50.0                          # a channel the ReLU kills everywhere
    gamma, beta = gamma.cuda(), beta.cuda()
    rm, rv, cnt = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros((), dtype=torch.long, device="cuda")
    stats = ops.bn_train_stats(y.view(-1, C), gamma, beta, rm, rv, cnt, None)
    pooled, arg = ops.bn_relu_maxpool_fwd(y, stats[2], stats[3])
    # the three-kernel chain with the same statistics
    z, mask = torch.empty_like(y), torch.empty(B * H * W * C // 8, dtype=torch.uint8, device="cuda")
    ops._call("ralf_bn_apply", ops.dtype_code(y), ops._p(y), ops._p(stats[2]), ops._p(stats[3]), None, ops._p(z), ops._p(mask), B * H * W, C, 1)
    pooled2, arg2 = ops.maxpool_fwd(z)
    assert torch.equal(pooled, pooled2) and torch.equal(arg, arg2)
    dpool = rnd(*pooled.shape, seed=73, dtype=dtype).cuda()
    dy, dg, db = ops.bn_relu_maxpool_bwd(dpool, arg, y, stats, gamma, None)
    dz = ops.maxpool_bwd(dpool, arg2, tuple(y.shape))
    dx2, dg2, db2, _ = ops.bn_backward(y.view(-1, C), dz.view(-1, C), None, gamma, stats[0], stats[1], True, False, True, mask=mask)
    tol = dict(atol=2e-2, rtol=2e-2) if dtype == torch.bfloat16 else dict(atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(dy.view(-1, C).float(), dx2.float(), **tol)
    torch.testing.assert_close(dg, dg2, atol=2e-3 * float(dg2.abs().max()), rtol=1e-3)
    torch.testing.assert_close(db, db2, atol=2e-3 * float(db2.abs().max()), rtol=1e-3)
    if dtype == torch.float32:   # torch's own chain on the CPU
        yc = y.cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
        g, bta = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
        ref = F.max_pool2d(torch.relu(F.batch_norm(yc, None, None, g, bta, True, 0.1, 1e-5)), 3, 2, 1)
        torch.testing.assert_close(pooled.cpu().permute(0, 3, 1, 2), ref.detach(), atol=2e-5, rtol=1e-5)
        ref.backward(dpool.cpu().permute(0, 3, 1, 2))
        torch.testing.assert_close(dy.cpu().permute(0, 3, 1, 2), yc.grad, atol=3e-5, rtol=1e-3)
        torch.testing.assert_close(dg.cpu(), g.grad, atol=1e-3, rtol=1e-3)
        torch.testing.assert_close(db.cpu(), bta.grad, atol=1e-3, rtol=1e-3)
