"""ORACLE (test infrastructure, NOT product code) -- CPU fp32 restatement of the RALF hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
file.  It is the checker the HIP path is compared against, never the thing shipped or measured.

Everything here is a *functional* restatement over a flat ``state_dict`` (same key names as the
reference checkpoint, SURVEY.md section 5 "Checkpoint") in plain ``torch`` CPU ops.  Each function
cites the reference file:line it follows (paths relative to /root/reference).

Pinning status:
  * transformer / fusion / decoder / loss / layout-encoder math: PINNED against outputs of the
    stub-imported reference itself (tests/golden/make_golden.py -> tests/golden/*.npz,
    checked by tests/test_oracle_golden.py).
  * ResNet-50 body: the arithmetic lives in timm 0.9.x + torchvision 0.14 FX (absent from
    /root/reference and from this image) -> restated from the published ResNet-v1.5 bottleneck
    definition.  PARITY UNPINNED for the body (no reference run possible here).
  * backbone WRAPPER (4-channel stem construction, FPN fuse, projection; common/image.py:70-111): PINNED against the
    reference's own ResnetBackbone run around a stand-in body (tests/golden/backbone_wrapper.npz).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

NEG_INF = float("-inf")


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def layer_norm(x: Tensor, sd: SD, p: str, eps: float = 1e-5) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * sd[p + ".weight"] + sd[p + ".bias"]


def linear(x: Tensor, sd: SD, p: str) -> Tensor:
    y = x @ sd[p + ".weight"].t()
    b = sd.get(p + ".bias")
    return y if b is None else y + b


def drop(x: Tensor, p: float) -> Tensor:
    return F.dropout(x, p, True) if p > 0.0 else x


def torch_mha(
    q_in: Tensor,
    kv_in: Tensor,
    sd: SD,
    p: str,
    nhead: int,
    add_mask: Optional[Tensor] = None,  # [B,1|H,Sq,Sk] additive float (-inf = blocked)
    p_drop: float = 0.0,
) -> Tensor:
    """torch.nn.MultiheadAttention (batch-first view), packed in_proj [3d,d] in q,k,v order.
    Used by nn.TransformerEncoderLayer / DecoderLayer built at
    image2layout/train/models/retrieval_augmented_autoreg.py:116-126 and common/common.py:25-34."""
    B, Sq, d = q_in.shape
    Sk = kv_in.shape[1]
    dh = d // nhead
    W, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = q_in @ W[:d].t() + b[:d]
    k = kv_in @ W[d : 2 * d].t() + b[d : 2 * d]
    v = kv_in @ W[2 * d :].t() + b[2 * d :]
    q = q.view(B, Sq, nhead, dh).transpose(1, 2)
    k = k.view(B, Sk, nhead, dh).transpose(1, 2)
    v = v.view(B, Sk, nhead, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if add_mask is not None:
        s = s + add_mask
    a = drop(torch.softmax(s, dim=-1), p_drop)
    o = (a @ v).transpose(1, 2).reshape(B, Sq, d)
    return o @ sd[p + ".out_proj.weight"].t() + sd[p + ".out_proj.bias"]


def key_padding_to_additive(kpm: Optional[Tensor]) -> Optional[Tensor]:
    """bool [B,Sk] (True = pad) -> additive [B,1,1,Sk]."""
    if kpm is None:
        return None
    m = torch.zeros(kpm.shape, dtype=torch.float32)
    m.masked_fill_(kpm, NEG_INF)
    return m[:, None, None, :]


def causal_additive(S: int) -> Tensor:
    """nn.Transformer.generate_square_subsequent_mask (common/common.py:117)."""
    return torch.triu(torch.full((S, S), NEG_INF), diagonal=1)[None, None]


def encoder_layer_prenorm(x, sd, p, nhead, kpm_add=None, p_drop=0.0):
    """nn.TransformerEncoderLayer(norm_first=True, activation=relu)."""
    h = layer_norm(x, sd, p + ".norm1")
    x = x + drop(torch_mha(h, h, sd, p + ".self_attn", nhead, kpm_add, p_drop), p_drop)
    h = layer_norm(x, sd, p + ".norm2")
    h = linear(drop(torch.relu(linear(h, sd, p + ".linear1")), p_drop), sd, p + ".linear2")
    return x + drop(h, p_drop)


def encoder_layer_postnorm(x, sd, p, nhead, kpm_add=None, p_drop=0.0):
    """nn.TransformerEncoderLayer default (norm_first=False) as used by FIDNetV3
    (image2layout/train/fid/model.py:26-33)."""
    x = layer_norm(x + drop(torch_mha(x, x, sd, p + ".self_attn", nhead, kpm_add, p_drop), p_drop), sd, p + ".norm1")
    h = linear(drop(torch.relu(linear(x, sd, p + ".linear1")), p_drop), sd, p + ".linear2")
    return layer_norm(x + drop(h, p_drop), sd, p + ".norm2")


def decoder_layer_prenorm(x, mem, sd, p, nhead, self_mask, p_drop=0.0):
    """nn.TransformerDecoderLayer(norm_first=True); memory is NOT masked
    (common/common.py:116-123: no memory_key_padding_mask is passed)."""
    h = layer_norm(x, sd, p + ".norm1")
    x = x + drop(torch_mha(h, h, sd, p + ".self_attn", nhead, self_mask, p_drop), p_drop)
    h = layer_norm(x, sd, p + ".norm2")
    x = x + drop(torch_mha(h, mem, sd, p + ".multihead_attn", nhead, None, p_drop), p_drop)
    h = layer_norm(x, sd, p + ".norm3")
    h = linear(drop(torch.relu(linear(h, sd, p + ".linear1")), p_drop), sd, p + ".linear2")
    return x + drop(h, p_drop)


def pe1d_table(max_len: int, d_model: int) -> Tensor:
    """PositionalEncoding1d buffer `pe` (common/positional_encoding.py:77-90), shape [max_len,d]."""
    pos = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def pos_enc_1d(x: Tensor, pe: Tensor, p_drop: float = 0.0) -> Tensor:
    """x*sqrt(d) + pe[:S] (common/positional_encoding.py:92-107)."""
    d = x.shape[-1]
    return drop(x * math.sqrt(d) + pe[: x.shape[1]].to(x.dtype), p_drop)


def pos2d_sine_table(h: int, w: int, d_model: int, temperature: float = 10000.0) -> Tensor:
    """PositionEmbeddingSine(normalize=True, scale=2pi) table [h*w, d_model]
    (common/positional_encoding.py:182-209): first d/2 channels = y, last d/2 = x;
    coordinates divided by (h-1),(w-1)."""
    half = d_model // 2
    y, x = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    y = y / (h - 1) * (2 * math.pi)
    x = x / (w - 1) * (2 * math.pi)
    dim_t = torch.arange(half).float()
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / half)
    px = x.flatten()[:, None] / dim_t
    py = y.flatten()[:, None] / dim_t
    px = torch.stack((px[:, 0::2].sin(), px[:, 1::2].cos()), dim=2).flatten(1)
    py = torch.stack((py[:, 0::2].sin(), py[:, 1::2].cos()), dim=2).flatten(1)
    return torch.cat((py, px), dim=1)


def feed_forward(x: Tensor, sd: SD, p: str) -> Tensor:
    """FeedForward = LN -> Linear -> GELU(erf) -> Linear (common/attention.py:15-30);
    nn.Sequential indices 0,1,4 carry parameters."""
    h = layer_norm(x, sd, p + ".net.0")
    h = F.gelu(linear(h, sd, p + ".net.1"))
    return linear(h, sd, p + ".net.4")


def xattn_fuse(x: Tensor, ctx: Tensor, sd: SD, p: str, heads: int = 8, dim_head: int = 64) -> Tensor:
    """Attention.forward (common/attention.py:49-71): pre-LN on queries only, bias-free q/kv,
    softmax over the K retrieved keys, out proj with bias."""
    B, n, _ = x.shape
    xn = layer_norm(x, sd, p + ".norm")
    q = xn @ sd[p + ".to_q.weight"].t()
    kv = ctx @ sd[p + ".to_kv.weight"].t()
    inner = heads * dim_head
    k, v = kv[..., :inner], kv[..., inner:]
    q = q.view(B, n, heads, dim_head).transpose(1, 2)
    k = k.view(B, -1, heads, dim_head).transpose(1, 2)
    v = v.view(B, -1, heads, dim_head).transpose(1, 2)
    a = torch.softmax((q @ k.transpose(-1, -2)) * dim_head ** -0.5, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, n, inner)
    return o @ sd[p + ".to_out.0.weight"].t() + sd[p + ".to_out.0.bias"]


# --------------------------------------------------------------------------------------
# frozen layout encoder (a4)
# --------------------------------------------------------------------------------------
def fidnet_extract(sd: SD, p: str, layout: Dict[str, Tensor], nhead: int = 4, nlayers: int = 4, p_drop: float = 0.0) -> Tensor:
    """FIDNetV3.extract_features (fid/model.py:90-103) + TransformerWithToken (:15-50).
    layout: label int64 [B,N]; center_x/center_y/width/height float [B,N]; mask bool [B,N].
    Returns token-0 feature [B,256]."""
    bbox = torch.stack([layout[k] for k in ("center_x", "center_y", "width", "height")], dim=-1)
    hb = linear(bbox, sd, p + ".fc_bbox")
    hl = sd[p + ".emb_label.weight"][layout["label"]]
    x = torch.relu(linear(torch.cat([hb, hl], dim=-1), sd, p + ".enc_fc_in"))  # [B,N,d]
    B = x.shape[0]
    tok = sd[p + ".enc_transformer.token"].reshape(1, 1, -1).expand(B, -1, -1)
    x = torch.cat([tok, x], dim=1)
    pad = torch.cat([torch.zeros(B, 1, dtype=torch.bool), ~layout["mask"].bool()], dim=1)
    add = key_padding_to_additive(pad)
    for i in range(nlayers):
        x = encoder_layer_postnorm(x, sd, f"{p}.enc_transformer.core.layers.{i}", nhead, add, p_drop)
    return x[:, 0]


# --------------------------------------------------------------------------------------
# ResNet-50 / FPN backbone (a1)  -- body parity unpinned, see header
# --------------------------------------------------------------------------------------
def _bn(x, sd, p, training, stats_out=None):
    if training:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.1, 1e-5)


RESNET50_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # (planes, blocks, stride)


def resnet50_fpn(img: Tensor, sd: SD, p: str = "encoder.extractor", training: bool = False) -> Tensor:
    """ResnetBackbone.forward (common/image.py:90-120): 4-channel stem, ResNet-v1.5 bottlenecks
    (stride on the 3x3), taps at layer3/layer4, FPN fuse, 1x1 proj. img [B,4,H,W] -> [B,256,H/16,W/16]."""
    feats = resnet50_body(img, sd, p, training)
    return fpn_fuse(feats[3], feats[4], sd, p)


def resnet50_body(img: Tensor, sd: SD, p: str = "encoder.extractor", training: bool = False) -> Dict[int, Tensor]:
    """the body alone (what create_feature_extractor(timm resnet50, {layer3, layer4}) returns, common/image.py:39-67): stage outputs
    {1..4}.  timm / torchvision are absent here; CROSS-CHECKED against an independent implementation of the same published architecture,
    transformers.ResNetModel(layer_type="bottleneck", downsample_in_bottleneck=False) = ResNet-v1.5 (tests/golden/resnet_body_hf.npz,
    tests/golden/make_golden.py: golden_resnet_body_hf) -- not a pin to timm itself, but no reading of the architecture shared with this file."""
    b = p + ".body"
    x = F.conv2d(img, sd[b + ".conv1.weight"], None, 2, 3)
    x = torch.relu(_bn(x, sd, b + ".bn1", training))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = {}
    for li, (planes, blocks, stride) in enumerate(RESNET50_STAGES, start=1):
        for bi in range(blocks):
            q = f"{b}.layer{li}.{bi}"
            s = stride if bi == 0 else 1
            idn = x
            y = torch.relu(_bn(F.conv2d(x, sd[q + ".conv1.weight"]), sd, q + ".bn1", training))
            y = torch.relu(_bn(F.conv2d(y, sd[q + ".conv2.weight"], None, s, 1), sd, q + ".bn2", training))
            y = _bn(F.conv2d(y, sd[q + ".conv3.weight"]), sd, q + ".bn3", training)
            if (q + ".downsample.0.weight") in sd:
                idn = _bn(F.conv2d(x, sd[q + ".downsample.0.weight"], None, s), sd, q + ".downsample.1", training)
            x = torch.relu(y + idn)
        feats[li] = x
    return feats


def stem_weight_4ch(w3: Tensor) -> Tensor:
    """the 4-channel stem the wrapper builds from the pretrained 3-channel 7x7 filters: 4th input channel (saliency) =
    mean of the RGB filters (common/image.py:70-77).  PINNED: tests/golden/backbone_wrapper.npz."""
    return torch.cat([w3, w3.mean(dim=1, keepdim=True)], dim=1)


def fpn_fuse(layer3: Tensor, layer4: Tensor, sd: SD, p: str = "encoder.extractor") -> Tensor:
    """FPN fuse + projection of ResnetBackbone.forward (common/image.py:99-111): 1x1 laterals, nearest up-sampling of the
    layer-4 map to the layer-3 size (F.interpolate(size=...): src = floor(dst * in / out), non-integer ratio at 350x240),
    3x3 on the sum, channel concat [up, conv33], 1x1 proj.  PINNED: tests/golden/backbone_wrapper.npz."""
    f4 = F.conv2d(layer3, sd[p + ".fpn_conv11_4.weight"], sd[p + ".fpn_conv11_4.bias"])
    f5 = F.conv2d(layer4, sd[p + ".fpn_conv11_5.weight"], sd[p + ".fpn_conv11_5.bias"])
    f5u = F.interpolate(f5, size=f4.shape[2:], mode="nearest")
    fused = torch.cat([f5u, F.conv2d(f5u + f4, sd[p + ".fpn_conv33.weight"], sd[p + ".fpn_conv33.bias"], 1, 1)], dim=1)
    return F.conv2d(fused, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


# --------------------------------------------------------------------------------------
# RALF forward (a2-a11)
# --------------------------------------------------------------------------------------
def image_memory(feat: Tensor, sd: SD, nhead: int = 8, nlayers: int = 6, p_drop: float = 0.0) -> Tensor:
    """feat [B,256,h,w] -> PositionEmbeddingSine -> 6 pre-norm encoder layers, no final norm
    (retrieval_augmented_autoreg.py:967-970)."""
    B, C, h, w = feat.shape
    x = feat.flatten(2).transpose(1, 2) + pos2d_sine_table(h, w, C)[None]
    for i in range(nlayers):
        x = encoder_layer_prenorm(x, sd, f"transformer_encoder.layers.{i}", nhead, None, p_drop)
    return x


def retrieved_features(retrieved: Dict[str, Tensor], sd: SD, top_k: int, p_drop: float = 0.0) -> Tensor:
    """extract_retrieved_features (retrieval_augmented_autoreg.py:526-584) with
    use_reference_image=False: per k -> frozen layout encoder -> layout_adapter; stack;
    x*sqrt(d)+PE[0:K].  The K slices are batched into one B*K call (same math)."""
    B = retrieved["label"].shape[0]
    flat = {k: retrieved[k][:, :top_k].reshape(B * top_k, -1) for k in ("label", "mask", "center_x", "center_y", "width", "height")}
    flat["label"] = flat["label"].long()
    flat["mask"] = flat["mask"].bool()
    f = fidnet_extract(sd, "layout_encoer", flat, p_drop=p_drop)
    f = feed_forward(f, sd, "layout_adapter").view(B, top_k, -1)
    return pos_enc_1d(f, sd["pos_emb_1d.pe"][0], p_drop)


def constraint_features(seq: Tensor, pad_mask: Tensor, sd: SD, nhead: int = 8, nlayers: int = 6, p_drop: float = 0.0) -> Tensor:
    """UserConstraintTransformerEncoder.forward (common/common.py:238-252), task_token=None."""
    h = pos_enc_1d(sd["user_const_encoder.emb.weight"][seq], sd["user_const_encoder.pos_emb.pe"][0], p_drop)
    add = key_padding_to_additive(pad_mask)
    for i in range(nlayers):
        h = encoder_layer_prenorm(h, sd, f"user_const_encoder.encoder.layers.{i}", nhead, add, p_drop)
    return h


def ralf_memory(sd: SD, feat: Tensor, retrieved, seq_const, const_pad, top_k=16, p_drop=0.0) -> Tensor:
    """ConcateAuxilaryTaskConcateCrossAttnRetrievalAugmentedAutoreg._encode_into_memory
    (retrieval_augmented_autoreg.py:963-994, 1004-1033) given the backbone output `feat`."""
    mem = image_memory(feat, sd, p_drop=p_drop)
    ref = retrieved_features(retrieved, sd, top_k, p_drop)
    ca = xattn_fuse(mem, ref, sd, "attn")
    m = feed_forward(torch.cat([mem, ca, ref], dim=1), sd, "head")
    cf = constraint_features(seq_const, const_pad, sd, p_drop=p_drop)
    te = sd["task_emb.weight"]  # [2,1] learned scalars, broadcast over channels
    return torch.cat([m + te[0], cf + te[1]], dim=1)


def autoreg_memory(sd: SD, feat: Tensor, seq_const, const_pad, p_drop=0.0) -> Tensor:
    """ConcateAuxilaryTaskAutoreg._encode_into_memory (autoreg.py:594-622) -- baseline, no retrieval."""
    mem = image_memory(feat, sd, p_drop=p_drop)
    cf = constraint_features(seq_const, const_pad, sd, p_drop=p_drop)
    te = sd["task_emb.weight"]
    return torch.cat([mem + te[0], cf + te[1]], dim=1)


def decoder_logits(sd: SD, tgt: Tensor, memory: Tensor, tgt_pad: Optional[Tensor], nhead=8, nlayers=6, p_drop=0.0) -> Tensor:
    """BaseDecoder.forward (common/common.py:84-135) with is_causal=True."""
    h = pos_enc_1d(sd["decoder.emb.weight"][tgt], sd["decoder.pos_emb.pe"][0], p_drop)
    S = h.shape[1]
    mask = causal_additive(S)
    if tgt_pad is not None:
        mask = mask + key_padding_to_additive(tgt_pad)
    for i in range(nlayers):
        h = decoder_layer_prenorm(h, memory, sd, f"decoder.transformer.layers.{i}", nhead, mask, p_drop)
    h = layer_norm(h, sd, "decoder.head.0")
    return h @ sd["decoder.head.1.weight"].t()


def xent_label_smoothing(logits: Tensor, target: Tensor, ignore_index: int, eps: float = 0.1) -> Tensor:
    """nn.CrossEntropyLoss(label_smoothing=0.1, ignore_index=pad) on [B,S,V] logits
    (retrieval_augmented_autoreg.py:140-142, 209-216): mean over non-ignored targets of
    (1-eps)*nll + eps/V * sum_c(-logp_c)."""
    V = logits.shape[-1]
    logp = torch.log_softmax(logits.reshape(-1, V), dim=-1)
    t = target.reshape(-1)
    keep = t != ignore_index
    nll = -logp.gather(1, t.clamp(0, V - 1)[:, None])[:, 0]
    smooth = -logp.sum(-1) / V
    per = (1 - eps) * nll + eps * smooth
    return (per * keep).sum() / keep.sum()


def ralf_forward(sd: SD, inputs: dict, top_k: int = 16, training_bn: bool = False, feat: Optional[Tensor] = None, p_drop: float = 0.0) -> Tensor:
    """BaseRetrievalAugmentedAutoreg.forward (retrieval_augmented_autoreg.py:190-207).
    `feat` overrides the backbone output (backbone-independent fixtures)."""
    if feat is None:
        feat = resnet50_fpn(inputs["image"], sd, training=training_bn)
    mem = ralf_memory(sd, feat, inputs["retrieved"], inputs["seq_layout_const"], inputs["seq_layout_const_pad_mask"], top_k, p_drop)
    return decoder_logits(sd, inputs["seq"], mem, inputs["tgt_key_padding_mask"], p_drop=p_drop)


def autoreg_forward(sd: SD, inputs: dict, training_bn: bool = False, feat: Optional[Tensor] = None, p_drop: float = 0.0) -> Tensor:
    if feat is None:
        feat = resnet50_fpn(inputs["image"], sd, training=training_bn)
    mem = autoreg_memory(sd, feat, inputs["seq_layout_const"], inputs["seq_layout_const_pad_mask"], p_drop)
    return decoder_logits(sd, inputs["seq"], mem, inputs["tgt_key_padding_mask"], p_drop=p_drop)


# --------------------------------------------------------------------------------------
# greedy decode (a13, sampling=deterministic) -- full-prefix recompute like the reference
# --------------------------------------------------------------------------------------
def greedy_decode(sd: SD, memory: Tensor, token_mask: Tensor, bos: int, pad: int, max_tokens: int,
                  restrict=None, prefix: Optional[Tensor] = None) -> Tensor:
    """sample() loop (retrieval_augmented_autoreg.py:249-297) with argmax sampling
    (helpers/sampling.py:25-26).  token_mask bool [max_tokens, V] (True = allowed).
    restrict(step_index_plus_1, logits)->logits applies DECODE_SPACE_RESTRICTION."""
    B = memory.shape[0]
    seq = torch.full((B, 1), bos, dtype=torch.long)
    start = 0
    if prefix is not None:
        seq = torch.cat([seq, prefix], dim=1)
        start = prefix.shape[1]
    for i in range(start, max_tokens):
        logits = decoder_logits(sd, seq, memory, seq == pad)[:, i].clone()
        logits[:, ~token_mask[i]] = NEG_INF
        if restrict is not None:
            logits = restrict(i + 1, logits)
        seq = torch.cat([seq, logits.argmax(dim=1, keepdim=True)], dim=1)
    return seq[:, 1:]
