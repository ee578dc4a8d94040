"""The one-launch transformer layer (ralf_tlayer_fwd, ralf_amd/csrc/tlayer.hip) against the launches it replaces: same bits for every
tensor it writes, and -- through the modules -- the same output and the same gradients as the unfused layers
(nn.TransformerDecoderLayer / nn.TransformerEncoderLayer, norm_first: image2layout/train/models/common/common.py:25-34,84-135,216-226)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
D, H, FF = 256, 8, 1024


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def make_weights(cross, seed=0):
    w = {}
    names = ["ln1", "ln3"] + (["ln2"] if cross else [])
    for i, n in enumerate(names):
        w[n] = ((1 + 0.1 * rnd(D, seed=seed + i)).cuda(), (0.1 * rnd(D, seed=seed + 10 + i)).cuda())
    shapes = {"sa_in": (3 * D, D), "sa_out": (D, D), "ffn1": (FF, D), "ffn2": (D, FF)}
    if cross:
        shapes.update({"ca_in": (3 * D, D), "ca_out": (D, D)})
    for i, (n, (no, ni)) in enumerate(sorted(shapes.items())):
        w[n] = ((rnd(no, ni, seed=seed + 20 + i) * ni ** -0.5).to(torch.bfloat16).cuda(), (0.1 * rnd(no, seed=seed + 40 + i)).cuda())
    return w


def unfused(x, w, kv, kpm, causal, p, seed, calls):
    """the launches the fused kernel replaces, in their order (functional.LayerNormSkipFn / LinearFn / AttnFn / FFNFn forward)"""
    from ralf_amd import ops

    B, S, _ = x.shape
    rows = B * S
    t = {}
    sd = seed if p > 0 else None
    t["h1"], t["mean1"], t["rstd1"] = ops.layernorm_fwd(x, *w["ln1"])
    t["qkv"] = ops.gemm(t["h1"].view(rows, D), w["sa_in"][0], rows, 3 * D, D, bias=w["sa_in"][1]).view(B, S, 3 * D)
    t["o1"], t["lse1"] = ops.attention_fwd(t["qkv"], t["qkv"], t["qkv"], B, H, S, S, D // H, 0, D, 2 * D, causal=causal, kpm=kpm, p_drop=p, seed=seed, call_id=calls[0])
    t["x1"] = ops.gemm(t["o1"].view(rows, D), w["sa_out"][0], rows, D, D, bias=w["sa_out"][1], res=x.view(rows, D), drop_p=p, seed=sd, call_id=calls[1]).view(B, S, D)
    r = t["x1"]
    if kv is not None:
        t["h2"], t["mean2"], t["rstd2"] = ops.layernorm_fwd(r, *w["ln2"])
        t["q"] = ops.gemm(t["h2"].view(rows, D), w["ca_in"][0][:D], rows, D, D, bias=w["ca_in"][1][:D]).view(B, S, D)
        t["o2"], t["lse2"] = ops.attention_fwd(t["q"], kv, kv, B, H, S, kv.shape[1], D // H, 0, 0, D, p_drop=p, seed=seed, call_id=calls[2])
        t["x2"] = ops.gemm(t["o2"].view(rows, D), w["ca_out"][0], rows, D, D, bias=w["ca_out"][1], res=r.view(rows, D), drop_p=p, seed=sd, call_id=calls[3]).view(B, S, D)
        r = t["x2"]
    t["h3"], t["mean3"], t["rstd3"] = ops.layernorm_fwd(r, *w["ln3"])
    t["hid"] = ops.gemm(t["h3"].view(rows, D), w["ffn1"][0], rows, FF, D, bias=w["ffn1"][1], act="relu", drop_p=p, seed=sd, call_id=calls[4]).view(B, S, FF)
    t["out"] = ops.gemm(t["hid"].view(rows, FF), w["ffn2"][0], rows, D, FF, bias=w["ffn2"][1], res=r.view(rows, D), drop_p=p, seed=sd, call_id=calls[5]).view(B, S, D)
    return t


def same_bits(a, b, name):
    a, b = a.contiguous().view(-1), b.contiguous().view(-1)
    ia = a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32)
    ib = b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32)
    bad = (ia != ib).sum().item()
    if bad:
        diff = (a.float() - b.float()).abs()
        raise AssertionError(f"{name}: {bad} of {a.numel()} elements differ (max |diff| {diff.max().item():.3e}, first at {int((ia != ib).nonzero()[0])})")


CASES = [
    # B, S, M (None = encoder layer), causal, key padding, p
    (3, 50, 532, True, True, 0.1),     # decoder layer of the training step: 5 x 10 tokens, 2 x 14 x 19 memory rows
    (2, 64, 64, True, False, 0.0),     # full strip, one memory tile
    (2, 1, 7, True, False, 0.1),       # a single token; a memory shorter than a tile
    (5, 17, 130, True, True, 0.0),
    (4, 33, 611, False, True, 0.1),    # not causal + padded keys over three 16-query groups
    (3, 40, None, False, True, 0.1),   # encoder layer (constraint encoder)
    (2, 64, None, False, False, 0.0),
    (1, 16, None, True, True, 0.1),
]


@pytest.mark.parametrize("B,S,M,causal,padded,p", CASES)
def test_fused_layer_writes_the_bits_of_the_unfused_launches(B, S, M, causal, padded, p):
    from ralf_amd import ops

    cross = M is not None
    w = make_weights(cross, seed=S)
    x = rnd(B, S, D, seed=1).to(torch.bfloat16).cuda()
    kv = rnd(B, M, 2 * D, seed=2).to(torch.bfloat16).cuda() if cross else None
    kpm = None
    if padded:
        kpm = torch.zeros(B, S, dtype=torch.uint8)
        for b in range(B):
            kpm[b, max(1, S - 1 - 3 * b):] = 1 if S > 1 else 0    # a padded tail of varying length, key 0 always visible
        kpm = kpm.cuda()
    seed = torch.tensor([1234567], dtype=torch.int64, device="cuda")
    calls = (3, 4, 5, 6, 7, 8)
    ref = unfused(x, w, kv, kpm, causal, p, seed, calls)
    names = ["sa_in", "sa_out"] + (["q_proj", "out2"] if cross else []) + ["ffn1", "ffn2"]
    rowmajor = {"sa_in": w["sa_in"], "sa_out": w["sa_out"], "ffn1": w["ffn1"], "ffn2": w["ffn2"]}
    if cross:
        rowmajor.update({"q_proj": (w["ca_in"][0][:D], w["ca_in"][1][:D].contiguous()), "out2": w["ca_out"]})
    packed = ops.tlayer_pack([rowmajor[n][0] for n in names])
    W = {n: (pk, rowmajor[n][1]) for n, pk in zip(names, packed)}
    W.update({k: w[k] for k in ("ln1", "ln2", "ln3") if k in w})
    got = ops.tlayer_fwd(x, W, causal=causal, kpm=kpm, kv=kv, p_attn=p, p_res=p, seed=seed, calls=calls)
    torch.cuda.synchronize()
    assert set(got) == set(ref)
    order = ["h1", "mean1", "rstd1", "qkv", "o1", "lse1", "x1", "h2", "mean2", "rstd2", "q", "o2", "lse2", "x2", "h3", "mean3", "rstd3", "hid", "out"]
    for k in order:
        if k in ref:
            same_bits(got[k], ref[k], k)


def _layer(cross, seed):
    from ralf_amd import nn as RN

    torch.manual_seed(seed)
    layer = RN.TransformerDecoderLayer(D, H, FF, 0.1) if cross else RN.TransformerEncoderLayer(D, H, FF, 0.1, True)
    for prm in layer.parameters():
        torch.nn.init.normal_(prm, std=0.05)
    for n in ("norm1", "norm2", "norm3"):
        if hasattr(layer, n):
            getattr(layer, n).weight.data.add_(1.0)
    return layer.cuda()


@pytest.mark.parametrize("cross,training,B", [(True, True, 4), (True, False, 4), (False, True, 4), (True, True, 32), (False, True, 32), (True, False, 32)])
def test_fused_layer_module_matches_unfused_module_forward_and_backward(cross, training, B):
    """same output and same input / memory gradients bit for bit (with the dropout masks of the training mode); same parameter gradients.
    B = 32 (B S = 1600 rows = 25 strips of 64): the backward also takes the strip-wise launches (ops.tlayer_bwd / tlayer_bwd_lnqkv)"""
    from ralf_amd.functional import Runtime

    S, M = 50, 300
    layer = _layer(cross, 5)
    x0 = rnd(B, S, D, seed=1).to(torch.bfloat16).cuda()
    mem0 = rnd(B, M, D, seed=2).to(torch.bfloat16).cuda()
    kpm = torch.zeros(B, S, dtype=torch.uint8, device="cuda")
    kpm[1, 40:] = 1
    go = rnd(B, S, D, seed=3).to(torch.bfloat16).cuda()
    res = {}
    for fused in (True, False):
        rt = Runtime(torch.bfloat16, seed=11)
        rt.to(torch.device("cuda"))
        rt.training = training
        rt.fused_layers = fused
        rt.fused_ffn_bwd = fused
        rt.begin_step()
        x, mem = x0.clone().requires_grad_(True), mem0.clone().requires_grad_(True)
        layer.zero_grad(set_to_none=True)
        y = layer(x, mem, rt, kpm) if cross else layer(x, rt, kpm)
        y.backward(go)
        rt.flush_wgrads()
        rt.join_side()
        torch.cuda.synchronize()
        res[fused] = {"y": y.detach().clone(), "dx": x.grad.clone(), **({"dmem": mem.grad.clone()} if cross else {}),
                      **{n: prm.grad.clone() for n, prm in layer.named_parameters()}}
    assert set(res[True]) == set(res[False])
    for k in res[True]:
        if k in ("y", "dx", "dmem"):
            same_bits(res[True][k], res[False][k], k)
        else:   # parameter gradients: fp32 reductions over the rows with split / atomic partial sums (order not fixed between two runs)
            torch.testing.assert_close(res[True][k], res[False][k], rtol=1e-5, atol=1e-5 * res[False][k].abs().max().item(), msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("R,strip", [(256, None), (64, None), (48, None), (5, None), (64, 64), (96, 48)])
def test_decode_tail_equals_the_unfused_chain(R, strip):
    """part 2 on strips of batch rows, nothing kept (the KV-cached decode step's tail: nn.decoder_step): the bits of
    out-projection + residual -> LayerNorm -> linear1 + ReLU -> linear2 + residual as four launches"""
    from ralf_amd import ops

    w = make_weights(True, seed=R)
    o2 = rnd(R, D, seed=1).to(torch.bfloat16).cuda()
    x1 = rnd(R, D, seed=2).to(torch.bfloat16).cuda()
    r = ops.gemm(o2, w["ca_out"][0], R, D, D, bias=w["ca_out"][1], res=x1)
    h, _, _ = ops.layernorm_fwd(r, *w["ln3"], save_stats=False)
    f = ops.gemm(h, w["ffn1"][0], R, FF, D, bias=w["ffn1"][1], act="relu")
    want = ops.gemm(f, w["ffn2"][0], R, D, FF, bias=w["ffn2"][1], res=r)
    pk = ops.tlayer_pack([w["ca_out"][0], w["ffn1"][0], w["ffn2"][0]])
    got = ops.tlayer_tail(o2, x1, {"out2": (pk[0], w["ca_out"][1]), "ln3": w["ln3"], "ffn1": (pk[1], w["ffn1"][1]), "ffn2": (pk[2], w["ffn2"][1])}, rows_per_strip=strip)
    torch.cuda.synchronize()
    same_bits(got, want, "out")


def test_decode_step_with_the_one_launch_tail_equals_the_separate_launches():
    """nn.decoder_step with Runtime.fused_decode_tail (off by default: slower): same logits, bit for bit, over a few positions"""
    from ralf_amd import functional as RF, nn as RN

    torch.manual_seed(3)
    dec = RN.BaseDecoder(137, D, 2, H, FF).cuda()
    for prm in dec.parameters():
        torch.nn.init.normal_(prm, std=0.05)
    mem = rnd(64, 90, D, seed=4).to(torch.bfloat16).cuda()
    outs = {}
    for tail in (True, False, "default"):
        rt = RF.Runtime(torch.bfloat16, seed=1)
        rt.to(torch.device("cuda"))
        rt.fused_decode_tail = tail is True
        rt.fused_decode_token = tail == "default"   # (the one-launch-per-token step, csrc/decode_token.hip: the default; its bits are not the chain's)
        if tail != "default":   # the separate launches whose bits the one-launch tail writes: LayerNorm as its own kernel, single-chain products
            rt.decode_ln_gemm = rt.decode_few_row_split = False
        cache = RN.decoder_init_cache(dec, mem, rt, 8)
        assert (cache.packed is not None) == (tail is True)
        kpm = torch.zeros(64, 8, dtype=torch.uint8, device="cuda")
        toks = torch.randint(0, 137, (4, 64), generator=torch.Generator().manual_seed(11)).cuda()   # (teacher-forced: the same tokens for every variant)
        outs[tail] = torch.stack([RN.decoder_step(dec, toks[pos].contiguous(), pos, cache, rt, kpm[:, :pos + 1].contiguous()) for pos in range(4)])
    same_bits(outs[True], outs[False], "logits")
    # the default step (the whole stack in one launch per token, a workgroup per sample: other summation orders, the same rounding points): the
    # same logits to bf16 accumulation noise
    assert float((outs["default"] - outs[False]).abs().max()) <= 2e-2 * float(outs[False].abs().max())
    assert bool((outs["default"].argmax(-1) == outs[False].argmax(-1)).float().mean() > 0.97)


@pytest.mark.parametrize("layers,M", [(6, 540), (1, 33), (3, 1000)])
def test_one_launch_decode_token_equals_the_per_kernel_step(layers, M):
    """ralf_decode_token (csrc/decode_token.hip: embedding, every decoder layer and the head of one generated token in ONE launch, a workgroup per
    sample) against the per-kernel KV-cached step (ralf_embed_fwd / ralf_decode_attn / ralf_gemm chain, the round-5 path) on the same weights
    and caches: logits to bf16 accumulation noise over ten teacher-forced positions, padded prefix tokens masked, the caches they leave behind
    equal to rounding; then per-element positions (pos_vec): bit-identical to the all-at-one-position step of the same kernel."""
    from ralf_amd import functional as RF, nn as RN

    torch.manual_seed(5)
    B, T, V = 96, 12, 137
    dec = RN.BaseDecoder(V, D, layers, H, FF).cuda()
    for prm in dec.parameters():
        torch.nn.init.normal_(prm, std=0.05)
    for l in dec.transformer.layers:
        for nm in (l.norm1, l.norm2, l.norm3):
            torch.nn.init.normal_(nm.weight, mean=1.0, std=0.1)
    torch.nn.init.normal_(dec.head[0].weight, mean=1.0, std=0.1)
    mem = rnd(B, M, D, seed=6).to(torch.bfloat16).cuda()
    outs, caches = {}, {}
    g = torch.Generator().manual_seed(7)
    toks = torch.randint(0, V, (T, B), generator=g).cuda()
    kpm = torch.zeros(B, T, dtype=torch.uint8, device="cuda")
    kpm[3, 2] = 1; kpm[5, 1] = 1; kpm[7, 1:4] = 1       # (never position 0: a prefix whose keys are ALL masked has no softmax)
    for one in (True, False):
        rt = RF.Runtime(torch.bfloat16, seed=1)
        rt.to(torch.device("cuda"))
        rt.fused_decode_token = one
        cache = RN.decoder_init_cache(dec, mem, rt, T)
        assert RN._decode_token_ok(dec, cache, rt, toks[0]) == one
        outs[one] = torch.stack([RN.decoder_step(dec, toks[pos].contiguous(), pos, cache, rt, kpm, kpm_stride=T) for pos in range(10)])
        caches[one] = (cache, rt)
    torch.cuda.synchronize()
    a, b = outs[True], outs[False]
    assert torch.isfinite(a).all()
    assert float((a - b).abs().max()) <= 1e-2 * float(b.abs().max()), (float((a - b).abs().max()), float(b.abs().max()))
    assert bool((a.argmax(-1) == b.argmax(-1)).float().mean() > 0.97)
    for ka, kb in zip(caches[True][0].self_kv, caches[False][0].self_kv):
        assert float((ka[:, :10].float() - kb[:, :10].float()).abs().max()) <= 0.02 * (1 + float(kb.float().abs().max()))
    # per-element positions: every element re-steps a position of its own on the cache the teacher-forced decode left
    cache, rt = caches[True]
    pos = torch.randint(0, 10, (B,), generator=g)
    pos[0], pos[-1] = 0, 9
    tok = toks[pos.cuda(), torch.arange(B, device="cuda")].contiguous()
    kb = kpm.clone()
    for e in range(B):
        kb[e, int(pos[e]) + 1:] = 1
    got = RN.decoder_step(dec, tok, int(pos.max()), cache, rt, kb, kpm_stride=T, pos_vec=pos.to(torch.int32).cuda())
    want = torch.stack([outs[True][int(pos[e])][e] for e in range(B)])
    assert torch.equal(got, want)


@pytest.mark.parametrize("mode,top_k,top_p", [(0, 1, 1.0), (1, 5, 1.0), (2, 1, 0.9), (3, 1, 1.0), (4, 1, 1.0)])
def test_token_choice_inside_the_one_launch_step_equals_the_sampling_kernel(mode, top_k, top_p):
    """ralf_decode_token with its s_* arguments (the decode-space mask + token choice of helpers/sampling.py in the SAME launch as the decoder step: one
    wave per sample runs sample_core.h's row function on the logits it finds in LDS) against ralf_mask_sample_step on the logits the same launch wrote:
    the same tokens, sequence column and pad flags for every mode (deterministic, top_k, top_p, random, gumbel), with an allowed-token mask, forced
    tokens and a row offset."""
    from ralf_amd import functional as RF, nn as RN, ops

    torch.manual_seed(11)
    B, T, V, M = 160, 8, 518, 70
    dec = RN.BaseDecoder(V, D, 2, H, FF).cuda()
    for prm in dec.parameters():
        torch.nn.init.normal_(prm, std=0.08)
    mem = rnd(B, M, D, seed=12).to(torch.bfloat16).cuda()
    rt = RF.Runtime(torch.bfloat16, seed=3)
    rt.to(torch.device("cuda"))
    g = torch.Generator().manual_seed(13)
    allowed = (torch.rand(V, generator=g) > 0.3).to(torch.uint8).cuda()
    allowed[:4] = 1
    forced = torch.full((B,), -1, dtype=torch.int64)
    forced[5], forced[17] = 3, 0
    forced = forced.cuda()
    kpm = torch.zeros(B, T + 1, dtype=torch.uint8, device="cuda")
    res = {}
    for fused in (True, False):
        rt.fused_decode_sample = fused
        cache = RN.decoder_init_cache(dec, mem, rt, T)
        assert RN._decode_token_ok(dec, cache, rt, forced)
        seqbuf = torch.full((B, T + 1), 7, dtype=torch.int64, device="cuda")
        pad = kpm.clone()
        tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(14)).cuda()
        toks = []
        for pos in range(4):
            tok = RN.decoder_step(dec, tok, pos, cache, rt, pad, kpm_stride=T + 1,
                                  sample=dict(allowed=allowed, forced=forced if pos == 1 else None, mode=mode, top_k=top_k, temperature=0.8, seed=rt.seed, call_id=1000 + pos,
                                              seq_col=seqbuf[:, pos + 1], pad_flag_col=pad[:, pos + 1], pad_id=3, top_p=top_p, row0=32))
            toks.append(tok.clone())
        res[fused] = (torch.stack(toks), seqbuf.clone(), pad.clone())
    torch.cuda.synchronize()
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    t = res[True][0]
    assert bool(allowed[t[0]].all()) and int(t[1][5]) == 3 and int(t[1][17]) == 0 and bool(res[True][2][5, 2] == 1)
    assert mode == 0 or len(torch.unique(t)) > 8


def test_weight_packing_is_the_documented_permutation():
    from ralf_amd import ops

    for n, k, ld in ((64, 32, 32), (256, 1024, 1024), (256, 256, 768)):
        src = torch.arange(n * ld, dtype=torch.float32).remainder(4099).to(torch.bfloat16).view(n, ld).cuda()[:, :k]
        (pk,) = ops.tlayer_pack([src])
        torch.cuda.synchronize()
        # dst[((t * K/16 + i) * 64 + lane) * 8 + e] = src[32 t + (lane & 31)][16 i + 8 (lane >> 5) + e]
        want = src.view(n // 32, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
        assert torch.equal(pk, want)


def test_fused_layer_is_what_the_decoder_runs():
    """a decoder layer on 5N = 50 tokens takes the fused path (and falls back when the runtime is fp32 or the sequence is long)"""
    from ralf_amd import functional as RF, ops

    seen = []
    orig = ops.tlayer_fwd

    def spy(*a, **k):
        seen.append(a[0].shape)
        return orig(*a, **k)

    ops.tlayer_fwd = spy
    try:
        layer = _layer(True, 6)
        rt = RF.Runtime(torch.bfloat16, seed=1)
        rt.to(torch.device("cuda"))
        x = rnd(2, 50, D, seed=1).to(torch.bfloat16).cuda()
        mem = rnd(2, 100, D, seed=2).to(torch.bfloat16).cuda()
        layer(x, mem, rt, None)
        assert seen == [x.shape]
        long = rnd(2, 65, D, seed=1).to(torch.bfloat16).cuda()
        layer(long, mem, rt, None)
        assert len(seen) == 1
        rt32 = RF.Runtime(torch.float32, seed=1)
        rt32.to(torch.device("cuda"))
        layer(x.float(), mem.float(), rt32, None)
        assert len(seen) == 1
    finally:
        ops.tlayer_fwd = orig


@pytest.mark.parametrize("rows,p", [(16384, 0.1), (128, 0.0), (64 * 7, 0.1)])
def test_feed_forward_block_on_strips_writes_the_bits_of_the_unfused_launches(rows, p):
    """part 3: LayerNorm -> linear1 + ReLU + dropout -> linear2 + dropout + residual on 64-row strips of any [rows, 256] tensor"""
    from ralf_amd import ops

    w = make_weights(False, seed=rows % 97)
    x = rnd(rows, D, seed=1).to(torch.bfloat16).cuda()
    seed = torch.tensor([4242], dtype=torch.int64, device="cuda")
    sd = seed if p > 0 else None
    h, mean, rstd = ops.layernorm_fwd(x, *w["ln3"])
    hid = ops.gemm(h, w["ffn1"][0], rows, FF, D, bias=w["ffn1"][1], act="relu", drop_p=p, seed=sd, call_id=7)
    out = ops.gemm(hid, w["ffn2"][0], rows, D, FF, bias=w["ffn2"][1], res=x, drop_p=p, seed=sd, call_id=8)
    pk = ops.tlayer_pack([w["ffn1"][0], w["ffn2"][0]])
    t = ops.tlayer_ffn(x, {"ln3": w["ln3"], "ffn1": (pk[0], w["ffn1"][1]), "ffn2": (pk[1], w["ffn2"][1])}, p=p, seed=seed, calls=(0, 7, 8))
    torch.cuda.synchronize()
    for name, ref in (("h3", h), ("mean3", mean), ("rstd3", rstd), ("hid", hid), ("out", out)):
        same_bits(t[name], ref, name)


@pytest.mark.parametrize("rows", [16384, 192])
def test_layernorm_qkv_on_strips_writes_the_bits_of_the_unfused_launches(rows):
    """part 4: h1 = LayerNorm(x), qkv = h1 Win^T + bin on 64-row strips"""
    from ralf_amd import ops

    w = make_weights(False, seed=5)
    x = rnd(rows, D, seed=2).to(torch.bfloat16).cuda()
    h, mean, rstd = ops.layernorm_fwd(x, *w["ln1"])
    qkv = ops.gemm(h, w["sa_in"][0], rows, 3 * D, D, bias=w["sa_in"][1])
    (pk,) = ops.tlayer_pack([w["sa_in"][0]])
    t = ops.tlayer_lnqkv(x, {"ln1": w["ln1"], "sa_in": (pk, w["sa_in"][1])})
    torch.cuda.synchronize()
    for name, ref in (("h1", h), ("mean1", mean), ("rstd1", rstd), ("qkv", qkv)):
        same_bits(t[name], ref, name)


def test_feed_forward_tail_with_the_out_projection_on_strips():
    """part 2 on 64-row strips of many rows: r = x + drop(o Wo^T + bo); out = r + FFN(LN(r))"""
    from ralf_amd import ops

    rows, p = 4096, 0.1
    w = make_weights(True, seed=8)
    x = rnd(rows, D, seed=1).to(torch.bfloat16).cuda()
    o = rnd(rows, D, seed=2).to(torch.bfloat16).cuda()
    seed = torch.tensor([777], dtype=torch.int64, device="cuda")
    r = ops.gemm(o, w["ca_out"][0], rows, D, D, bias=w["ca_out"][1], res=x, drop_p=p, seed=seed, call_id=3)
    h, mean, rstd = ops.layernorm_fwd(r, *w["ln3"])
    hid = ops.gemm(h, w["ffn1"][0], rows, FF, D, bias=w["ffn1"][1], act="relu", drop_p=p, seed=seed, call_id=4)
    out = ops.gemm(hid, w["ffn2"][0], rows, D, FF, bias=w["ffn2"][1], res=r, drop_p=p, seed=seed, call_id=5)
    pk = ops.tlayer_pack([w["ca_out"][0], w["ffn1"][0], w["ffn2"][0]])
    t = ops.tlayer_ffn(x, {"out": (pk[0], w["ca_out"][1]), "ln3": w["ln3"], "ffn1": (pk[1], w["ffn1"][1]), "ffn2": (pk[2], w["ffn2"][1])}, o=o, p=p,
                       seed=seed, calls=(3, 4, 5))
    torch.cuda.synchronize()
    for name, ref in (("x2", r), ("h3", h), ("mean3", mean), ("rstd3", rstd), ("hid", hid), ("out", out)):
        same_bits(t[name], ref, name)


def test_long_sequence_encoder_layer_with_the_one_launch_feed_forward_half():
    """nn.TransformerEncoderLayer on 256 tokens per sample (the image encoder's layers): attention per operation, then out-projection +
    residual + LayerNorm + feed-forward + residual in one launch (functional.TFFNFn): output and input gradient bit for bit equal to the
    per-operation layer, parameter gradients equal; likewise with LayerNorm 1 + the q | k | v projection as one launch (three launches per layer)"""
    from ralf_amd.functional import Runtime

    layer = _layer(False, 9)
    x0 = rnd(4, 256, D, seed=1).to(torch.bfloat16).cuda()
    go = rnd(4, 256, D, seed=3).to(torch.bfloat16).cuda()
    res = {}
    for fused in (True, "tail only", "forward only", False):
        rt = Runtime(torch.bfloat16, seed=11)
        rt.to(torch.device("cuda"))
        rt.training = True
        rt.fused_ffn = bool(fused)
        rt.fused_lnqkv = fused is True or fused == "forward only"
        rt.fused_ffn_bwd = fused is True or fused == "tail only"
        rt.begin_step()
        x = x0.clone().requires_grad_(True)
        layer.zero_grad(set_to_none=True)
        y = layer(x, rt)
        y.backward(go)
        rt.flush_wgrads()
        rt.join_side()
        torch.cuda.synchronize()
        res[fused] = {"y": y.detach().clone(), "dx": x.grad.clone(), **{n: prm.grad.clone() for n, prm in layer.named_parameters()}}
    for variant in (True, "tail only", "forward only"):
        for k in res[variant]:
            if k in ("y", "dx"):
                same_bits(res[variant][k], res[False][k], k)
            else:
                torch.testing.assert_close(res[variant][k], res[False][k], rtol=1e-5, atol=1e-5 * res[False][k].abs().max().item(), msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("rows", [33792, 1024])
def test_gelu_feed_forward_on_strips_and_its_module(rows):
    """part 3 with GELU, the pre-activation kept, no residual (nn.FeedForward: LN -> Linear -> GELU -> Linear): the bits of the three launches;
    through the module: same output and input gradient as the per-operation path"""
    from ralf_amd import nn as RN, ops
    from ralf_amd.functional import Runtime

    w = make_weights(False, seed=13)
    x = rnd(rows, D, seed=1).to(torch.bfloat16).cuda()
    h, mean, rstd = ops.layernorm_fwd(x, *w["ln3"])
    z = torch.empty(rows, FF, dtype=torch.bfloat16, device="cuda")
    hid = ops.gemm(h, w["ffn1"][0], rows, FF, D, bias=w["ffn1"][1], act="gelu", out2=z)
    out = ops.gemm(hid, w["ffn2"][0], rows, D, FF, bias=w["ffn2"][1])
    pk = ops.tlayer_pack([w["ffn1"][0], w["ffn2"][0]])
    t = ops.tlayer_ffn(x, {"ln3": w["ln3"], "ffn1": (pk[0], w["ffn1"][1]), "ffn2": (pk[1], w["ffn2"][1])}, act="gelu", residual=False)
    torch.cuda.synchronize()
    for name, ref in (("h3", h), ("mean3", mean), ("rstd3", rstd), ("z", z), ("hid", hid), ("out", out)):
        same_bits(t[name], ref, name)
    if rows > 2048:
        return
    torch.manual_seed(2)
    ffm = RN.FeedForward(D, FF).cuda()
    for prm in ffm.parameters():
        torch.nn.init.normal_(prm, std=0.05)
    ffm.net[0].weight.data.add_(1.0)
    go = rnd(rows, D, seed=3).to(torch.bfloat16).cuda()
    res = {}
    for fused in (True, False):
        rt = Runtime(torch.bfloat16, seed=1)
        rt.to(torch.device("cuda"))
        rt.fused_ffn = fused
        xx = x.clone().requires_grad_(True)
        ffm.zero_grad(set_to_none=True)
        y = ffm(xx, rt)
        y.backward(go)
        rt.flush_wgrads()
        rt.join_side()
        torch.cuda.synchronize()
        res[fused] = (y.detach().clone(), xx.grad.clone())
    same_bits(res[True][0], res[False][0], "y")
    same_bits(res[True][1], res[False][1], "dx")


@pytest.mark.parametrize("rows,p", [(16384, 0.1), (192, 0.0)])
def test_tail_backward_stage_1_writes_the_bits_of_the_two_data_gradient_products(rows, p):
    """ralf_tlayer_bwd stage 1: dz = (dy_m W2) masked by the forward hidden, dh = dz W1 (functional.FFNFn.backward's two products)"""
    from ralf_amd import ops

    w = make_weights(False, seed=21)
    dy_m = rnd(rows, D, seed=1).to(torch.bfloat16).cuda()
    hid = torch.relu(rnd(rows, FF, seed=2)).to(torch.bfloat16).cuda()
    dz = ops.gemm(dy_m, w["ffn2"][0], rows, FF, D, b_kcontig=False, aux=hid, aux_mode="relu_mask", aux_scale=1.0 / (1.0 - p))
    dh = ops.gemm(dz, w["ffn1"][0], rows, D, FF, b_kcontig=False)
    pk = ops.tlayer_pack([w["ffn2"][0], w["ffn1"][0]], transpose=(0, 1))
    t = ops.tlayer_bwd(dy_m, hid, {"w2t": pk[0], "w1t": pk[1]}, p=p)
    torch.cuda.synchronize()
    same_bits(t["dz"], dz, "dz")
    same_bits(t["g"], dh, "dh")


@pytest.mark.parametrize("rows,p", [(16384, 0.1), (192, 0.0), (64, 0.1), (3200, 0.1)])
def test_tail_backward_writes_the_bits_of_the_four_data_gradient_launches(rows, p):
    """ralf_tlayer_bwd stage 3 against ralf_gemm (dz) -> ralf_gemm (dh) -> ralf_layernorm_bwd (with the skip gradient and the masked second
    output) -> ralf_gemm (d o): dz, g, g_m, d_o bit for bit; dgamma / dbeta (fp32 atomics in both) to rounding"""
    from ralf_amd import ops

    w = make_weights(True, seed=23)
    dy = rnd(rows, D, seed=1).to(torch.bfloat16).cuda()
    seed = torch.tensor([31337], dtype=torch.int64, device="cuda")
    dy_m = ops.dropout(dy, p, seed, 9) if p > 0 else dy
    hid = torch.relu(rnd(rows, FF, seed=2)).to(torch.bfloat16).cuda()
    x2 = rnd(rows, D, seed=3).to(torch.bfloat16).cuda()
    _, mean, rstd = ops.layernorm_fwd(x2, *w["ln3"])
    dz = ops.gemm(dy_m, w["ffn2"][0], rows, FF, D, b_kcontig=False, aux=hid, aux_mode="relu_mask", aux_scale=1.0 / (1.0 - p))
    dh = ops.gemm(dz, w["ffn1"][0], rows, D, FF, b_kcontig=False)
    out = ops.layernorm_bwd(dh, x2, w["ln3"][0], mean, rstd, need_wgrad=True, skip=dy, drop=(p, seed, 4) if p > 0 else None)
    g, dgam, dbet = out[0], out[1], out[2]
    g_m = out[3] if p > 0 else g
    d_o = ops.gemm(g_m, w["ca_out"][0], rows, D, D, b_kcontig=False)
    pk = ops.tlayer_pack([w["ffn2"][0], w["ffn1"][0], w["ca_out"][0]], transpose=(0, 1, 2))
    dgam2, dbet2 = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    t = ops.tlayer_bwd(dy_m, hid, {"w2t": pk[0], "w1t": pk[1], "wot": pk[2]}, p=p, dy=dy, x2=x2, mean3=mean, rstd3=rstd, gamma=w["ln3"][0],
                       dgamma=dgam2, dbeta=dbet2, seed=seed, call_out=4)
    torch.cuda.synchronize()
    same_bits(t["dz"], dz, "dz")
    same_bits(t["g"], g, "g")
    same_bits(t["g_m"], g_m, "g_m")
    same_bits(t["d_o"], d_o, "d_o")
    torch.testing.assert_close(dgam2, dgam, rtol=1e-4, atol=1e-4 * dgam.abs().max().item())
    torch.testing.assert_close(dbet2, dbet, rtol=1e-4, atol=1e-4 * dbet.abs().max().item())


@pytest.mark.parametrize("rows,p,skip", [(16384, 0.1, True), (128, 0.0, False), (64, 0.1, True), (3200, 0.1, True)])
def test_layernorm_qkv_backward_writes_the_bits_of_the_two_launches(rows, p, skip):
    """ralf_tlayer_bwd stage 4 against ralf_gemm (dh = dqkv Win) -> ralf_layernorm_bwd (skip gradient, masked second output)"""
    from ralf_amd import ops

    w = make_weights(False, seed=29)
    dqkv = rnd(rows, 3 * D, seed=1).to(torch.bfloat16).cuda()
    x = rnd(rows, D, seed=2).to(torch.bfloat16).cuda()
    sk = rnd(rows, D, seed=3).to(torch.bfloat16).cuda() if skip else None
    seed = torch.tensor([2024], dtype=torch.int64, device="cuda")
    _, mean, rstd = ops.layernorm_fwd(x, *w["ln1"])
    dh = ops.gemm(dqkv, w["sa_in"][0], rows, D, 3 * D, b_kcontig=False)
    out = ops.layernorm_bwd(dh, x, w["ln1"][0], mean, rstd, need_wgrad=True, skip=sk, drop=(p, seed, 6) if p > 0 else None)
    (wt,) = ops.tlayer_pack([w["sa_in"][0]], transpose=(0,))
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dx, dxm = ops.tlayer_bwd_lnqkv(dqkv, wt, x, mean, rstd, w["ln1"][0], skip=sk, dgamma=dg, dbeta=db, p=p, seed=seed, call=6)
    torch.cuda.synchronize()
    same_bits(dx, out[0], "dx")
    if p > 0:
        same_bits(dxm, out[3], "dx masked")
    torch.testing.assert_close(dg, out[1], rtol=1e-4, atol=1e-4 * out[1].abs().max().item())
    torch.testing.assert_close(db, out[2], rtol=1e-4, atol=1e-4 * out[2].abs().max().item())


@pytest.mark.parametrize("B", [4, 8])
def test_long_sequence_decoder_layer_takes_the_strip_kernels(B):
    """S = 160 tokens per sample (N = 32 elements): [LayerNorm 1 + q|k|v] and [out-projection 2 + LayerNorm 3 + FFN] on strips of all samples' rows,
    the rest per operation (functional._tlayer_fwd_long); B = 8 (1 280 rows): the backward on strips too.  Same bits as the per-operation layer."""
    from ralf_amd.functional import Runtime

    S, M = 160, 300
    layer = _layer(True, 15)
    x0 = rnd(B, S, D, seed=1).to(torch.bfloat16).cuda()
    mem0 = rnd(B, M, D, seed=2).to(torch.bfloat16).cuda()
    kpm = torch.zeros(B, S, dtype=torch.uint8, device="cuda")
    kpm[1, 120:] = 1
    go = rnd(B, S, D, seed=3).to(torch.bfloat16).cuda()
    res = {}
    for fused in (True, False):
        rt = Runtime(torch.bfloat16, seed=11)
        rt.to(torch.device("cuda"))
        rt.training = True
        rt.fused_layers = fused
        rt.fused_ffn_bwd = fused
        rt.begin_step()
        x, mem = x0.clone().requires_grad_(True), mem0.clone().requires_grad_(True)
        layer.zero_grad(set_to_none=True)
        assert layer.fusable(x, rt) == fused
        y = layer(x, mem, rt, kpm)
        y.backward(go)
        rt.flush_wgrads()
        rt.join_side()
        torch.cuda.synchronize()
        res[fused] = {"y": y.detach().clone(), "dx": x.grad.clone(), "dmem": mem.grad.clone(), **{n: prm.grad.clone() for n, prm in layer.named_parameters()}}
    for k in res[True]:
        if k in ("y", "dx", "dmem"):
            same_bits(res[True][k], res[False][k], k)
        else:
            torch.testing.assert_close(res[True][k], res[False][k], rtol=1e-5, atol=1e-5 * res[False][k].abs().max().item(), msg=lambda m: f"{k}: {m}")
