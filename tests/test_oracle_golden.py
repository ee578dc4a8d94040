"""Pin the oracle (oracle/ralf_oracle.py) against outputs recorded from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import pytest
import torch

from oracle import ralf_oracle as O
from oracle.detweights import det_state_dict, resnet50_fpn_shapes

from conftest import GOLDEN

THIN = 37


def thin(v):
    return v.flatten()[::THIN] if v.numel() > 20000 else v


def shapes(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return {k: tuple(v) for k, v in json.load(f)["shapes"].items()}


def sd_for(prefixes, all_shapes):
    return det_state_dict({k: v for k, v in all_shapes.items() if any(k.startswith(p) for p in prefixes)})


def close(a, b, atol=2e-5, rtol=1e-4):
    torch.testing.assert_close(a.float(), b.float(), atol=atol, rtol=rtol)


@pytest.fixture(scope="module")
def ralf_sd():
    sd = det_state_dict(shapes("ralf_state_shapes.json"))
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    return sd


def test_positional_tables(golden):
    g = golden("modules.npz")
    close(O.pos2d_sine_table(3, 5, 256), g.sub("pos2d_3x5")["table"])
    close(O.pos2d_sine_table(16, 16, 256), g.sub("pos2d_16x16")["table"])
    p = g.sub("pe1d")
    close(O.pe1d_table(64, 256), p["pe_head"])
    close(O.pos_enc_1d(p["x"], O.pe1d_table(5000, 256)), p["y"])


def test_xattn_and_ff(golden, ralf_sd):
    g = golden("modules.npz")
    r = g.sub("xattn")
    x, ctx = r["x"].requires_grad_(True), r["ctx"].requires_grad_(True)
    y = O.xattn_fuse(x, ctx, ralf_sd, "attn")
    close(y, r["y"])
    gx, gc, gq, gkv, gn = torch.autograd.grad(y, [x, ctx, ralf_sd["attn.to_q.weight"], ralf_sd["attn.to_kv.weight"], ralf_sd["attn.norm.weight"]], r["go"])
    close(gx, r["gx"]); close(gc, r["gctx"]); close(thin(gq), r["g_to_q"]); close(thin(gkv), r["g_to_kv"]); close(gn, r["g_norm_w"], atol=1e-4)
    r = g.sub("ff")
    x = r["x"].requires_grad_(True)
    y = O.feed_forward(x, ralf_sd, "head")
    close(y, r["y"])
    gx, gw1, gb2 = torch.autograd.grad(y, [x, ralf_sd["head.net.1.weight"], ralf_sd["head.net.4.bias"]], r["go"])
    close(gx, r["gx"]); close(thin(gw1), r["g_w1"]); close(gb2, r["g_b2"], atol=1e-4)


def test_encoder_decoder_layers(golden, ralf_sd):
    g = golden("modules.npz")
    r = g.sub("enc_layer")
    x = r["x"].requires_grad_(True)
    p = "transformer_encoder.layers.0"
    y = O.encoder_layer_prenorm(x, ralf_sd, p, 8, O.key_padding_to_additive(r["kpm"]))
    close(y, r["y"])
    ws = [ralf_sd[p + ".self_attn.in_proj_weight"], ralf_sd[p + ".self_attn.in_proj_bias"], ralf_sd[p + ".linear2.weight"], ralf_sd[p + ".norm1.bias"]]
    gs = torch.autograd.grad(y, [x] + ws, r["go"])
    close(gs[0], r["gx"]); close(thin(gs[1]), r["g_in_proj_w"]); close(gs[2], r["g_in_proj_b"], atol=1e-4)
    close(thin(gs[3]), r["g_lin2_w"]); close(gs[4], r["g_norm1_b"], atol=1e-4)
    r = g.sub("dec_layer")
    x, mem = r["x"].requires_grad_(True), r["mem"].requires_grad_(True)
    p = "decoder.transformer.layers.0"
    mask = O.causal_additive(10) + O.key_padding_to_additive(r["kpm"])
    y = O.decoder_layer_prenorm(x, mem, ralf_sd, p, 8, mask)
    close(y, r["y"])
    gs = torch.autograd.grad(y, [x, mem, ralf_sd[p + ".multihead_attn.in_proj_weight"], ralf_sd[p + ".self_attn.out_proj.weight"]], r["go"])
    close(gs[0], r["gx"]); close(gs[1], r["gmem"]); close(thin(gs[2]), r["g_cross_in_w"]); close(thin(gs[3]), r["g_self_out_w"])


def test_fidnet(golden, ralf_sd):
    r = golden("modules.npz").sub("fidnet")
    with torch.no_grad():
        f = O.fidnet_extract(ralf_sd, "layout_encoer", r["in"])
    close(f, r["feat"])


@pytest.mark.parametrize("task", ["uncond", "refinement", "c"])
def test_ralf_e2e(golden, ralf_sd, task):
    r = golden("e2e.npz").sub("ralf_" + task)
    feat = r["feat"].requires_grad_(True)
    inputs = dict(r["inputs"])
    inputs["retrieved"] = r["retrieved"]
    logits = O.ralf_forward(ralf_sd, inputs, feat=feat)
    close(logits, r["logits"], atol=1e-4)
    loss = O.xent_label_smoothing(logits, r["targets"]["seq"], ignore_index=515)
    close(loss, r["loss"], atol=1e-5)
    keys = list(r["grads"].keys())
    params = [v for v in ralf_sd.values() if v.requires_grad]
    allg = torch.autograd.grad(loss, [feat] + params, allow_unused=True)
    close(allg[0], r["gfeat"], atol=1e-6)
    byname = {k: g for (k, v), g in zip([(k, v) for k, v in ralf_sd.items() if v.requires_grad], allg[1:])}
    for k in keys:
        close(thin(byname[k]), r["grads"][k], atol=2e-6, rtol=2e-3)
    trainable = [g for k, g in byname.items() if g is not None and not k.startswith("layout_encoer.")]
    gn = torch.sqrt(sum((g ** 2).sum() for g in trainable))
    close(gn, r["gradnorm"], atol=1e-6, rtol=1e-3)


def test_autoreg_e2e(golden):
    r = golden("e2e.npz").sub("autoreg_uncond")
    sd = det_state_dict(shapes("autoreg_state_shapes.json"))
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    feat = r["feat"].requires_grad_(True)
    logits = O.autoreg_forward(sd, r["inputs"], feat=feat)
    close(logits, r["logits"], atol=1e-4)
    loss = O.xent_label_smoothing(logits, r["targets"]["seq"], ignore_index=515)
    close(loss, r["loss"], atol=1e-5)
    ks = list(r["grads"].keys())
    gs = torch.autograd.grad(loss, [feat] + [sd[k] for k in ks])
    close(gs[0], r["gfeat"], atol=1e-6)
    for k, g in zip(ks, gs[1:]):
        close(thin(g), r["grads"][k], atol=2e-6, rtol=2e-3)


def _wrapper_inputs(seed, h3, w3, h4, w4):
    """the injected layer3 / layer4 maps and output gradient of tests/golden/make_golden.py: golden_backbone_wrapper"""
    gg = torch.Generator().manual_seed(seed)
    f3 = torch.randn(1, 1024, h3, w3, generator=gg)
    f4 = torch.randn(1, 2048, h4, w4, generator=gg)
    go = torch.randn(1, 256, h3, w3, generator=gg)
    return f3, f4, go


WRAPPER_CASES = {"c256": (16, 16, 8, 8), "c350x240": (22, 15, 11, 8)}


def test_backbone_wrapper_stem_channel(golden):
    """a1 wrapper, stem: the 4th input channel the reference's constructor builds (common/image.py:70-77)"""
    r = golden("backbone_wrapper.npz").sub("stem")
    w4 = O.stem_weight_4ch(r["w3"])
    assert torch.equal(w4, r["w4"])
    close(torch.nn.functional.conv2d(r["img"], w4, None, 2, 3), r["y"], atol=1e-6)


@pytest.mark.parametrize("case", list(WRAPPER_CASES))
def test_backbone_wrapper_fpn(golden, case):
    """a1 wrapper, FPN fuse + projection (common/image.py:99-111) incl. the non-integer nearest up-sampling at 350x240"""
    r = golden("backbone_wrapper.npz").sub(case)
    f3, f4, go = _wrapper_inputs(int(r["seed"]), *WRAPPER_CASES[case])
    f3.requires_grad_(True), f4.requires_grad_(True)
    sd = det_state_dict({k: v for k, v in resnet50_fpn_shapes().items() if ".body." not in k})
    for v in sd.values():
        v.requires_grad_(True)
    y = O.fpn_fuse(f3, f4, sd)
    close(y, r["y"], atol=1e-5)
    ks = list(r["grads"].keys())
    gs = torch.autograd.grad(y, [f3, f4] + [sd["encoder.extractor." + k] for k in ks], go)
    close(thin(gs[0]), r["g_layer3"], atol=1e-5); close(thin(gs[1]), r["g_layer4"], atol=1e-5)
    for k, g in zip(ks, gs[2:]):
        close(thin(g), r["grads"][k], atol=1e-4, rtol=1e-3)


def test_ralf_cgl_e2e(golden):
    """BASELINE config 3's label set (CGL: 4 labels, V = 519, Vc = 549) against the reference's recorded logits / gradients"""
    r = golden("e2e_cgl.npz").sub("ralf_c")
    assert int(r["meta"]["N_total"]) == 519 and int(r["meta"]["preproc_N_total"]) == 549
    sd = det_state_dict(shapes("ralf_cgl_state_shapes.json"))
    assert sd["decoder.emb.weight"].shape[0] == 519 and sd["user_const_encoder.emb.weight"].shape[0] == 549 and sd["layout_encoer.emb_label.weight"].shape[0] == 4
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    feat = r["feat"].requires_grad_(True)
    inputs = dict(r["inputs"])
    inputs["retrieved"] = r["retrieved"]
    logits = O.ralf_forward(sd, inputs, feat=feat)
    close(logits, r["logits"], atol=1e-4)
    loss = O.xent_label_smoothing(logits, r["targets"]["seq"], ignore_index=516)
    close(loss, r["loss"], atol=1e-5)
    ks = list(r["grads"].keys())
    gs = torch.autograd.grad(loss, [feat] + [sd[k] for k in ks])
    close(gs[0], r["gfeat"], atol=1e-6)
    for k, g in zip(ks, gs[1:]):
        close(thin(g), r["grads"][k], atol=2e-6, rtol=2e-3)


def hf_body_inputs(seed):
    """img / output gradients of tests/golden/make_golden.py: golden_resnet_body_hf, regenerated from the seed"""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(2, 4, 96, 128, generator=g)
    return img, torch.randn(2, 1024, 6, 8, generator=g) * 0.1, torch.randn(2, 2048, 3, 4, generator=g) * 0.1


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_resnet_body_against_an_independent_implementation(golden, mode):
    """a1 body: the oracle's restatement of ResNet-50 v1.5 against transformers.ResNetModel (bottleneck, downsample_in_bottleneck=False)
    carrying the same deterministic weights -- layer3 / layer4 maps and gradients, on running and on batch statistics.  Not a pin to timm
    (absent), but a second reading of the published architecture that shares nothing with oracle/ralf_oracle.py."""
    z = golden("resnet_body_hf.npz")
    r = z.sub(mode)
    img, go3, go4 = hf_body_inputs(int(z["seed"]))
    sd = det_state_dict({k: v for k, v in resnet50_fpn_shapes().items() if ".body." in k})
    for k, v in sd.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    x = img.clone().requires_grad_(True)
    f = O.resnet50_body(x, sd, training=(mode == "train"))
    close(f[3].flatten()[::5], r["layer3"], atol=2e-5, rtol=2e-4)
    close(f[4].flatten()[::3], r["layer4"], atol=2e-5, rtol=2e-4)
    ks = list(r["grads"].keys())
    gs = torch.autograd.grad((f[3] * go3).sum() + (f[4] * go4).sum(), [x] + [sd["encoder.extractor.body." + k] for k in ks])
    close(thin(gs[0]), r["g_img"], atol=1e-5, rtol=2e-3)
    for k, g in zip(ks, gs[1:]):
        close(thin(g), r["grads"][k], atol=5e-5, rtol=5e-3)
