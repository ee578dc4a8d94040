"""Modules of the RALF hot path built on the HIP ops, with the SAME parameter / buffer names and
shapes as the torch modules the reference instantiates (checkpoints load with strict=True):

  nn.TransformerEncoderLayer / nn.TransformerDecoderLayer / nn.MultiheadAttention
      (image2layout/train/models/retrieval_augmented_autoreg.py:116-126, common/common.py:25-34,216-226)
  FeedForward, Attention                      (common/attention.py:15-71)
  BaseDecoder, UserConstraintTransformerEncoder (common/common.py:13-135,200-252)
  FIDNetV3 feature extractor                  (fid/model.py:15-103,150-175)
  ResnetFeatureExtractor / ResnetBackbone     (common/image.py:27-129; body = timm resnet50 keys)
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import functional as RF
from . import ops
from .functional import Runtime


def _param(*shape):
    return nn.Parameter(torch.empty(*shape))


class Affine(nn.Module):
    """weight (+ bias) container named like nn.Linear / nn.LayerNorm / nn.Conv2d."""

    def __init__(self, *wshape, bias_dim: Optional[int] = None):
        super().__init__()
        self.weight = _param(*wshape)
        if bias_dim is not None:
            self.bias = _param(bias_dim)
        else:
            self.register_parameter("bias", None)


def pe1d_buffer(max_len: int, d: int) -> torch.Tensor:
    """PositionalEncoding1d.pe (common/positional_encoding.py:77-90), [1, max_len, d]."""
    pos = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2) * (-math.log(10000.0) / d))
    pe = torch.zeros(1, max_len, d)
    pe[0, :, 0::2] = torch.sin(pos * div)
    pe[0, :, 1::2] = torch.cos(pos * div)
    return pe


def pos2d_sine(h: int, w: int, d_model: int, temperature: float = 10000.0) -> torch.Tensor:
    """PositionEmbeddingSine(normalize=True) table [h*w, d] (common/positional_encoding.py:182-209)."""
    half = d_model // 2
    y, x = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    y = y / (h - 1) * (2 * math.pi)
    x = x / (w - 1) * (2 * math.pi)
    dim_t = temperature ** (2 * torch.div(torch.arange(half).float(), 2, rounding_mode="floor") / half)
    px, py = x.flatten()[:, None] / dim_t, y.flatten()[:, None] / dim_t
    px = torch.stack((px[:, 0::2].sin(), px[:, 1::2].cos()), dim=2).flatten(1)
    py = torch.stack((py[:, 0::2].sin(), py[:, 1::2].cos()), dim=2).flatten(1)
    return torch.cat((py, px), dim=1)


class PosEnc1d(nn.Module):
    def __init__(self, d_model: int, max_len: int = 5000, dropout: float = 0.1):
        super().__init__()
        self.register_buffer("pe", pe1d_buffer(max_len, d_model))
        self.p = dropout


class MHAParams(nn.Module):
    """nn.MultiheadAttention parameter layout: packed in_proj [3d, d] (q, k, v), out_proj."""

    def __init__(self, d: int, nhead: int):
        super().__init__()
        self.in_proj_weight = _param(3 * d, d)
        self.in_proj_bias = _param(3 * d)
        self.out_proj = Affine(d, d, bias_dim=d)
        self.d, self.nhead = d, nhead

    def self_attn(self, x, res, rt: Runtime, causal=False, kpm=None, p_attn=0.0, p_out=0.0):
        """res + drop(out_proj(attn(in_proj(x))))"""
        qkv = RF.linear(x, self.in_proj_weight, self.in_proj_bias, rt=rt)
        o = RF.AttnFn.apply(qkv, None, self.nhead, self.d // self.nhead, causal, kpm, rt.drop_p(p_attn), rt)
        return self._out(o, res, rt, p_out)

    def cross_attn(self, x, mem, res, rt: Runtime, p_attn=0.0, p_out=0.0):
        d = self.d
        q = RF.linear(x, self.in_proj_weight, self.in_proj_bias, rt=rt, rows=(0, d))
        kv = RF.linear(mem, self.in_proj_weight, self.in_proj_bias, rt=rt, rows=(d, 3 * d))
        o = RF.AttnFn.apply(q, kv, self.nhead, d // self.nhead, False, None, rt.drop_p(p_attn), rt)
        return self._out(o, res, rt, p_out)

    def cross_attn_stacked(self, x, kv_all, li, plan, res, rt: Runtime, p_attn=0.0, p_out=0.0):
        """cross_attn with K/V taken from the projection of all layers at once (functional.CrossKVFn)"""
        d = self.d
        q = RF.linear(x, self.in_proj_weight, self.in_proj_bias, rt=rt, rows=(0, d))
        o = RF.AttnCrossSliceFn.apply(q, kv_all, li, plan, self.nhead, d // self.nhead, rt.drop_p(p_attn), rt)
        return self._out(o, res, rt, p_out)

    def _out(self, o, res, rt, p_out):
        return RF.linear(o, self.out_proj.weight, self.out_proj.bias, res=res, rt=rt, p=rt.drop_p(p_out))


class _FFNMixin:
    def _ffn(self, h, res, rt, p):
        return RF.FFNFn.apply(h, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, res, "relu", rt.drop_p(p), rt)


class TransformerEncoderLayer(nn.Module, _FFNMixin):
    def __init__(self, d: int, nhead: int, dim_ff: int, dropout: float = 0.1, norm_first: bool = True):
        super().__init__()
        self.self_attn = MHAParams(d, nhead)
        self.linear1, self.linear2 = Affine(dim_ff, d, bias_dim=dim_ff), Affine(d, dim_ff, bias_dim=d)
        self.norm1, self.norm2 = Affine(d, bias_dim=d), Affine(d, bias_dim=d)
        self.p, self.norm_first = dropout, norm_first

    def _params(self):
        a = self.self_attn
        return (self.norm1.weight, self.norm1.bias, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                self.norm2.weight, self.norm2.bias, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias)

    def fusable(self, x, rt: Runtime) -> bool:
        return self.norm_first and RF.tlayer_supported(x, rt, self.self_attn.d, self.self_attn.nhead, self.linear1.weight.shape[0])

    def forward(self, x, rt: Runtime, kpm=None, packed=None):
        if self.fusable(x, rt):
            return RF.TLayerFn.apply(x, None, kpm, False, rt.drop_p(self.p), rt, packed, *self._params())   # the whole layer in one launch
        if self.norm_first:
            a = self.self_attn
            if rt.fused_lnqkv and rt.fused_ffn_out and RF.tffn_supported(x, rt, a.d, self.linear1.weight.shape[0]):
                # long sequences, three launches per layer: LayerNorm + q | k | v projection; attention; out-projection + residual + LayerNorm +
                # feed-forward + residual
                qkv, x = RF.TLNQKVFn.apply(x, rt, packed[0] if packed else None, self.norm1.weight, self.norm1.bias, a.in_proj_weight, a.in_proj_bias,
                                           packed[7] if packed and len(packed) > 7 else None)
                o = RF.AttnFn.apply(qkv, None, a.nhead, a.d // a.nhead, False, kpm, rt.drop_p(self.p), rt)
                return RF.TFFNFn.apply(x, o, rt.drop_p(self.p), rt, packed[1:7] if packed else None, a.out_proj.weight, a.out_proj.bias, self.norm2.weight,
                                       self.norm2.bias, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias)
            h, x = RF.layer_norm_skip(x, self.norm1.weight, self.norm1.bias, rt)
            if RF.tffn_supported(x, rt, a.d, self.linear1.weight.shape[0]):
                # long sequences: attention per operation, then out-projection + residual + LayerNorm + feed-forward + residual in one launch
                if not rt.fused_ffn_out:
                    x = a.self_attn(h, x, rt, kpm=kpm, p_attn=self.p, p_out=self.p)
                    return RF.TFFNFn.apply(x, None, rt.drop_p(self.p), rt, packed[2:] if packed else None, self.norm2.weight, self.norm2.bias,
                                           self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias)
                qkv = RF.linear(h, a.in_proj_weight, a.in_proj_bias, rt=rt)
                o = RF.AttnFn.apply(qkv, None, a.nhead, a.d // a.nhead, False, kpm, rt.drop_p(self.p), rt)
                return RF.TFFNFn.apply(x, o, rt.drop_p(self.p), rt, packed[1:] if packed else None, a.out_proj.weight, a.out_proj.bias, self.norm2.weight,
                                       self.norm2.bias, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias)
            x = a.self_attn(h, x, rt, kpm=kpm, p_attn=self.p, p_out=self.p)
            h, x = RF.layer_norm_skip(x, self.norm2.weight, self.norm2.bias, rt)
            return self._ffn(h, x, rt, self.p)
        x = RF.layer_norm(self.self_attn.self_attn(x, x, rt, kpm=kpm, p_attn=self.p, p_out=self.p), self.norm1.weight, self.norm1.bias, rt)
        return RF.layer_norm(self._ffn(x, x, rt, self.p), self.norm2.weight, self.norm2.bias, rt)


class TransformerDecoderLayer(nn.Module, _FFNMixin):
    def __init__(self, d: int, nhead: int, dim_ff: int, dropout: float = 0.1):
        super().__init__()
        self.self_attn, self.multihead_attn = MHAParams(d, nhead), MHAParams(d, nhead)
        self.linear1, self.linear2 = Affine(dim_ff, d, bias_dim=dim_ff), Affine(d, dim_ff, bias_dim=d)
        self.norm1, self.norm2, self.norm3 = Affine(d, bias_dim=d), Affine(d, bias_dim=d), Affine(d, bias_dim=d)
        self.p = dropout

    def _params(self):
        sa, ca = self.self_attn, self.multihead_attn
        return (self.norm1.weight, self.norm1.bias, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
                self.norm2.weight, self.norm2.bias, ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias,
                self.norm3.weight, self.norm3.bias, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias)

    def fusable(self, x, rt: Runtime) -> bool:
        return RF.tlayer_supported(x, rt, self.self_attn.d, self.self_attn.nhead, self.linear1.weight.shape[0], allow_long=True)

    def project_memory(self, mem, rt: Runtime):
        """the memory's packed k | v projection [B, M, 2d] of this layer's cross-attention"""
        sa, ca = self.self_attn, self.multihead_attn
        return RF.linear(mem, ca.in_proj_weight, ca.in_proj_bias, rt=rt, rows=(sa.d, 3 * sa.d))

    def forward(self, x, mem, rt: Runtime, tgt_kpm=None, stacked=None, packed=None, kv=None):
        """stacked = (kv_all, layer index, plan): this layer's cross-attention K/V were projected with all other layers' (BaseDecoder);
        packed = this layer's weights in fragment order when the caller packed all layers in one launch; kv = project_memory(mem) when the
        caller issued it ahead of the layer (BaseDecoder: on the projection stream)"""
        sa, ca = self.self_attn, self.multihead_attn
        if stacked is None and self.fusable(x, rt):
            # short target sequences: the memory's K/V projection (the one big product of the layer), then the layer in three launches
            if kv is None:
                kv = self.project_memory(mem, rt)
            return RF.TLayerFn.apply(x, kv, tgt_kpm, True, rt.drop_p(self.p), rt, packed, *self._params())
        h, x = RF.layer_norm_skip(x, self.norm1.weight, self.norm1.bias, rt)
        x = self.self_attn.self_attn(h, x, rt, causal=True, kpm=tgt_kpm, p_attn=self.p, p_out=self.p)
        h, x = RF.layer_norm_skip(x, self.norm2.weight, self.norm2.bias, rt)
        if stacked is not None:
            x = self.multihead_attn.cross_attn_stacked(h, stacked[0], stacked[1], stacked[2], x, rt, p_attn=self.p, p_out=self.p)
        else:
            x = self.multihead_attn.cross_attn(h, mem, x, rt, p_attn=self.p, p_out=self.p)  # memory is NOT masked (common.py:116-123)
        h, x = RF.layer_norm_skip(x, self.norm3.weight, self.norm3.bias, rt)
        return self._ffn(h, x, rt, self.p)


class LayerStack(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)


def _kpm_u8(mask: Optional[torch.Tensor]):
    """key-padding mask as the uint8 the kernels read: a bool tensor is reinterpreted (same bytes), no conversion kernel"""
    if mask is None:
        return None
    m = mask.contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)


class FeedForward(nn.Module):
    """LN -> Linear -> GELU -> Linear (common/attention.py:15-30); params at net.0 / net.1 / net.4."""

    def __init__(self, dim: int, hidden: int, out_dim: Optional[int] = None):
        super().__init__()
        out_dim = out_dim or dim
        self.net = nn.ModuleList([Affine(dim, bias_dim=dim), Affine(hidden, dim, bias_dim=hidden), nn.Identity(), nn.Identity(),
                                  Affine(out_dim, hidden, bias_dim=out_dim), nn.Identity()])

    def forward(self, x, rt: Runtime):
        if RF.tffn_supported(x, rt, self.net[1].weight.shape[1], self.net[1].weight.shape[0]) and self.net[4].weight.shape[0] == 256:
            # LayerNorm + both products in one launch on 64-row strips (the 1024-wide hidden stays in LDS between them)
            return RF.TFeedForwardFn.apply(x, rt, None, self.net[0].weight, self.net[0].bias, self.net[1].weight, self.net[1].bias, self.net[4].weight,
                                           self.net[4].bias)
        h = RF.layer_norm(x, self.net[0].weight, self.net[0].bias, rt)
        return RF.FFNFn.apply(h, self.net[1].weight, self.net[1].bias, self.net[4].weight, self.net[4].bias, None, "gelu", 0.0, rt)


class FuseAttention(nn.Module):
    """Attention (common/attention.py:33-71): pre-LN on queries, bias-free to_q / to_kv, 8 x 64 heads."""

    def __init__(self, dim_q: int, dim_kv: int, heads: int = 8, dim_head: int = 64):
        super().__init__()
        inner = heads * dim_head
        self.norm = Affine(dim_q, bias_dim=dim_q)
        self.to_q, self.to_kv = Affine(inner, dim_q), Affine(2 * inner, dim_kv)
        self.to_out = nn.ModuleList([Affine(dim_q, inner, bias_dim=dim_q), nn.Identity()])
        self.heads, self.dim_head = heads, dim_head

    def forward(self, x, ctx, rt: Runtime):
        xn = RF.layer_norm(x, self.norm.weight, self.norm.bias, rt)
        q = RF.linear(xn, self.to_q.weight, rt=rt)
        kv = RF.linear(ctx, self.to_kv.weight, rt=rt)
        o = RF.AttnFn.apply(q, kv, self.heads, self.dim_head, False, None, 0.0, rt)
        return RF.linear(o, self.to_out[0].weight, self.to_out[0].bias, rt=rt)


def pack_ffn_layers(layers, x, rt: Runtime):
    """fragment-order in_proj / out_proj / linear1 / linear2 weights (ops.tlayer_pack) of all `layers` in ONE launch when they take the
    strip-wise path on x (functional.TLNQKVFn / TFFNFn) and not the per-sample one (long sequences); else None"""
    if not layers or len(layers) * 8 > ops.TLAYER_PACK_MAX or any(l.fusable(x, rt) for l in layers):
        return None
    if not all(l.norm_first and RF.tffn_supported(x, rt, l.self_attn.d, l.linear1.weight.shape[0]) for l in layers):
        return None
    bwd = rt.fused_ffn_bwd and torch.is_grad_enabled()   # + W2^T, W1^T, Wo^T, Win^T for the one-launch backward kernels
    mats, tr = [], []
    for l in layers:
        mats += [rt.lp(l.self_attn.in_proj_weight), rt.lp(l.self_attn.out_proj.weight), rt.lp(l.linear1.weight), rt.lp(l.linear2.weight)]
        if bwd:
            tr += [len(mats), len(mats) + 1, len(mats) + 2, len(mats) + 3]
            mats += [rt.lp(l.linear2.weight), rt.lp(l.linear1.weight), rt.lp(l.self_attn.out_proj.weight), rt.lp(l.self_attn.in_proj_weight)]
    flat = ops.tlayer_pack(mats, transpose=tuple(tr))
    per = 8 if bwd else 4
    return [flat[per * i:per * (i + 1)] for i in range(len(layers))]


def _pack_layers(layers, x, rt: Runtime):
    """fragment-order weights (ops.tlayer_pack) of all `layers` in ONE launch when they take the fused path on x; else None"""
    if not layers or not all(l.fusable(x, rt) for l in layers):
        return None
    bwd = rt.fused_ffn_bwd and torch.is_grad_enabled()   # + the transposes the strip-wise backward streams
    mats, tr = [], []
    for l in layers:
        prm = l._params()
        mats += RF.tlayer_matrices(prm, rt)
        if bwd:
            mt = RF.tlayer_matrices_t(prm, rt)
            tr += list(range(len(mats), len(mats) + len(mt)))
            mats += mt
    per = len(mats) // len(layers)
    if len(mats) > ops.TLAYER_PACK_MAX:
        return None
    flat = ops.tlayer_pack(mats, transpose=tuple(tr))
    return [flat[i * per:(i + 1) * per] for i in range(len(layers))]


class BaseDecoder(nn.Module):
    def __init__(self, d_label: int, d_model: int = 256, num_layers: int = 6, nhead: int = 8, dim_feedforward: int = 1024):
        super().__init__()
        self.transformer = LayerStack([TransformerDecoderLayer(d_model, nhead, dim_feedforward, 0.1) for _ in range(num_layers)])
        self.emb = Affine(d_label, d_model)
        self.pos_emb = PosEnc1d(d_model)
        self.head = nn.ModuleList([Affine(d_model, bias_dim=d_model), Affine(d_label, d_model)])
        self.d_model = d_model

    def forward(self, tgt, memory, rt: Runtime, tgt_key_padding_mask=None):
        h = RF.EmbedFn.apply(tgt, self.emb.weight, self.pos_emb.pe[0], rt)
        h = RF.drop_add(h, None, rt.drop_p(self.pos_emb.p), rt)
        kpm = _kpm_u8(tgt_key_padding_mask)
        layers = list(self.transformer.layers)
        attns = [l.multihead_attn for l in layers]
        # d(memory) = sum of the 6 cross-attention K/V projections' data gradients, summed in their epilogues: the projections read an
        # alias that only they hold, so any other consumer of `memory` is summed by autograd as usual
        memory = rt.fanout_alias(memory.contiguous())
        plan = RF.CrossKVPlan.make(attns, rt) if torch.is_grad_enabled() else None
        kv_all = None
        if plan is not None:   # all layers' cross-attention K/V projections of the memory in one launch
            wb = [t for a in attns for t in (a.in_proj_weight, a.in_proj_bias)]
            kv_all = RF.CrossKVFn.apply(memory, plan, rt, *wb)
        packs = _pack_layers(layers, h, rt) if plan is None else None   # short sequences: every layer's weights in fragment order, one launch
        # The six K/V projections of the memory depend on nothing a decoder layer computes: in engine mode they are issued AHEAD of the layers on a
        # stream of their own (a parallel branch of the captured graph), beside the layers' 64 .. 100-workgroup launches that leave most of the chip
        # idle; layer l waits for projection l just before its cross-attention (ops.tlayer_fwd).  Autograd replays their backward (the d(memory)
        # products) on that stream too, so those leave the layers' backward chain as well.
        kvs = rt.ahead([(lambda l=l: l.project_memory(memory, rt)) for l in layers], "kv") if packs is not None else None
        for li, layer in enumerate(layers):
            h = layer(h, memory, rt, kpm, stacked=(kv_all, li, plan) if plan is not None else None, packed=packs[li] if packs else None,
                      kv=kvs[li] if kvs else None)
        h = RF.layer_norm(h, self.head[0].weight, self.head[0].bias, rt)
        return RF.linear(h, self.head[1].weight, rt=rt, out_f32=True)


_DECODE_KV_PACKED = os.environ.get("RALF_DECODE_KV_PACKED", "1") != "0"   # A/B runs: 0 keeps the row-major cross-attention cache


class DecodeCache:
    """per-layer state of the KV-cached decoder: cross-attention K/V of the fixed memory (computed once) and the
    growing self-attention K/V of the generated prefix."""

    def __init__(self, cross_kv, self_kv, max_len, packed=None, cross_rows=None, cross_packed=False):
        self.cross_kv, self.self_kv, self.max_len = cross_kv, self_kv, max_len
        self.packed = packed   # per layer: (cross out_proj, linear1, linear2) in fragment order (ops.tlayer_pack), or None
        # cross_packed: cross_kv[l] is [B, 2 (k | v), H / 2 head pairs, rows, 64] instead of [B, rows, 2 d]: the fused decode block's workgroup
        # (element, head pair) then streams two CONTIGUOUS 68 KB blocks instead of 128-byte pieces 1 KB apart (26.6 against 29.6 us per call
        # at B = 256 over 532 rows: 5.24 against 4.70 TB/s, tools/decode_attn_bench.py)
        self.cross_rows = cross_rows if cross_rows is not None else cross_kv[0].shape[1]
        self.cross_packed = cross_packed


@torch.no_grad()
def decoder_init_cache(dec: "BaseDecoder", memory: torch.Tensor, rt: Runtime, max_len: int) -> DecodeCache:
    B, M, d = memory.shape
    cross, selfkv = [], []
    mem2 = memory.reshape(B * M, d).contiguous()
    H = dec.transformer.layers[0].self_attn.nhead
    # the fused decode block's conditions (decoder_step): then the projection writes the head-pair-major layout itself -- 8 column slices
    # (k | v x head pair) x B elements as ONE batched product, slice j = 4 kv + hp of element b at [b, j, :, :]
    pack = (_DECODE_KV_PACKED and rt.fused_decode and rt.dtype == torch.bfloat16 and memory.is_cuda and d == 256 and H == 8 and max(M, max_len) <= ops.decode_attn_max_keys())
    for layer in dec.transformer.layers:
        a = layer.multihead_attn
        if pack:
            kv = torch.empty(B, 8, M, 64, dtype=rt.dtype, device=memory.device)
            ops.gemm(mem2, rt.lp(a.in_proj_weight)[d:], M, 64, d, bias=a.in_proj_bias.detach()[d:], batch=(8, B), sA=(0, M * d), sB=(64 * d, 0),
                     sC=(M * 64, 8 * M * 64), sBias0=64, out=kv)
            cross.append(kv)
        else:
            kv = ops.gemm(mem2, rt.lp(a.in_proj_weight)[d:], B * M, 2 * d, d, bias=a.in_proj_bias.detach()[d:])
            cross.append(kv.view(B, M, 2 * d))
        selfkv.append(torch.zeros(B, max_len, 2 * d, dtype=rt.dtype, device=memory.device))
    packed = None
    layers = list(dec.transformer.layers)
    if (rt.fused_decode and rt.fused_decode_tail and rt.dtype == torch.bfloat16 and memory.is_cuda and d == 256 and len(layers) * 3 <= ops.TLAYER_PACK_MAX
            and all(l.linear1.weight.shape[0] == 1024 for l in layers)):
        # the step's tail per layer (out-projection + residual, LayerNorm, feed-forward, residual) runs as one launch on weights in fragment order
        flat = ops.tlayer_pack([m for l in layers for m in (rt.lp(l.multihead_attn.out_proj.weight), rt.lp(l.linear1.weight), rt.lp(l.linear2.weight))])
        packed = [flat[3 * i:3 * i + 3] for i in range(len(layers))]
    return DecodeCache(cross, selfkv, max_len, packed, cross_rows=M, cross_packed=pack)


def _decode_token_ok(dec, cache, rt, tok) -> bool:
    """ralf_decode_token's conditions: bf16, d = 256, 8 heads, feed-forward 1024, the head-pair-major cross-attention cache, sizes within its LDS"""
    if not (rt.fused_decode and rt.fused_decode_token and rt.dtype == torch.bfloat16 and tok.is_cuda and cache.cross_packed and cache.packed is None):
        return False
    layers = dec.transformer.layers
    maxl, maxm = ops.decode_token_limits()
    return (dec.d_model == 256 and layers[0].self_attn.nhead == 8 and len(layers) <= 8 and all(l.linear1.weight.shape[0] == 1024 for l in layers)
            and cache.max_len <= maxl and cache.cross_rows <= maxm and getattr(dec.head[1], "bias", None) is None and tok.dtype == torch.long
            and dec.pos_emb.pe.dtype == torch.float32 and dec.emb.weight.dtype == torch.float32)


def _decode_token_weights(dec, rt):
    """device tensors ralf_decode_token reads, per layer (the bf16 shadows are refreshed in place when the masters change: Runtime.lp)"""
    d = dec.d_model
    layers = []
    for l in dec.transformer.layers:
        sa, ca = l.self_attn, l.multihead_attn
        layers.append({"w_qkv": rt.lp(sa.in_proj_weight), "b_qkv": sa.in_proj_bias.detach(), "ln1_g": l.norm1.weight.detach(), "ln1_b": l.norm1.bias.detach(),
                       "w_o1": rt.lp(sa.out_proj.weight), "b_o1": sa.out_proj.bias.detach(), "ln2_g": l.norm2.weight.detach(), "ln2_b": l.norm2.bias.detach(),
                       "w_q2": rt.lp(ca.in_proj_weight)[:d], "b_q2": ca.in_proj_bias.detach()[:d], "w_o2": rt.lp(ca.out_proj.weight), "b_o2": ca.out_proj.bias.detach(),
                       "ln3_g": l.norm3.weight.detach(), "ln3_b": l.norm3.bias.detach(), "w_f1": rt.lp(l.linear1.weight), "b_f1": l.linear1.bias.detach(),
                       "w_f2": rt.lp(l.linear2.weight), "b_f2": l.linear2.bias.detach()})
    return {"layers": layers, "head": (dec.head[0].weight.detach(), dec.head[0].bias.detach(), rt.lp(dec.head[1].weight))}


@torch.no_grad()
def decoder_step(dec: "BaseDecoder", tok: torch.Tensor, pos: int, cache: DecodeCache, rt: Runtime, kpm_prefix: torch.Tensor,
                 kpm_stride: Optional[int] = None, pos_vec: Optional[torch.Tensor] = None, sample: Optional[dict] = None):
    """one KV-cached decode step: token ids [B] at position `pos` -> fp32 logits [B, V].  kpm_prefix uint8
    [B, pos+1] marks padded prefix tokens (tgt_key_padding_mask of the reference's full-prefix call,
    retrieval_augmented_autoreg.py:274-279); with `kpm_stride` it is the [B, kpm_stride] mask buffer of the whole loop, of
    which the first pos+1 columns are read.  Same arithmetic as BaseDecoder.forward restricted to the last row.
    pos_vec (int32 [B] on the device; fused bf16 path only): every element sits at ITS OWN position pos_vec[b] <= pos -- the samples of
    sample_relation rewind their prefixes independently (retrieval_augmented_autoreg.py:432-460) and still step together.
    sample (optional): the keyword arguments of ops.mask_sample -- the call then returns the chosen TOKENS int64 [B] instead of the logits (the same
    function on the same logits: inside the one-launch step where that path runs, as a launch of its own behind the others)."""
    B = tok.shape[0]
    d, H = dec.d_model, dec.transformer.layers[0].self_attn.nhead
    dh = d // H
    if _decode_token_ok(dec, cache, rt, tok):
        # the whole step as ONE launch, a workgroup per sample (bf16 throughput mode; same rounding points as the launches below)
        lw = _decode_token_weights(dec, rt)
        kst = kpm_stride if kpm_stride else kpm_prefix.shape[1]
        fuse_choice = sample is not None and rt.fused_decode_sample and dec.head[1].weight.shape[0] <= 1024
        r = ops.decode_token(lw["layers"], lw["head"], dec.emb.weight.detach(), dec.pos_emb.pe[0], math.sqrt(d), tok, pos, cache.self_kv, cache.cross_kv,
                             cache.max_len, cache.cross_rows, dec.head[1].weight.shape[0], kpm=kpm_prefix, kpm_stride=kst, pos_vec=pos_vec,
                             sample=sample if fuse_choice else None)
        if sample is None:
            return r
        return r[1] if fuse_choice else ops.mask_sample(r, **sample)
    if pos_vec is None:
        x = ops.embed_fwd(tok.view(B, 1).contiguous(), dec.emb.weight.detach(), dec.pos_emb.pe[0, pos:pos + 1].contiguous(), 1, math.sqrt(d), rt.dtype).view(B, d)
    else:   # the batch as ONE sequence whose positional rows are gathered per element: the same kernel, the same arithmetic
        pe = dec.pos_emb.pe[0].index_select(0, pos_vec.long())
        x = ops.embed_fwd(tok.view(1, B).contiguous(), dec.emb.weight.detach(), pe, B, math.sqrt(d), rt.dtype).view(B, d)
    L = cache.max_len
    # bf16, d = 256, 8 heads: LayerNorm + q / k / v projections + attention of each block in ONE launch (ralf_decode_attn)
    # (the fused kernel keeps the scores of at most ops.decode_attn_max_keys() keys in LDS; longer memories, e.g. 512x512 canvases, take
    #  the per-kernel path below)
    fused = (rt.fused_decode and rt.dtype == torch.bfloat16 and d == 256 and H == 8 and x.stride(1) == 1
             and cache.cross_rows <= ops.decode_attn_max_keys() and pos + 1 <= ops.decode_attn_max_keys())
    if cache.cross_packed and not fused:
        raise RuntimeError("decoder_step: this cache holds the fused block's head-pair-major K/V layout (decoder_init_cache under rt.fused_decode); "
                           "build the cache with the runtime flags the steps run under")
    # per-element positions on the per-kernel path (the fp32 parity mode): the new k / v rows are scattered to row pos_vec[b] of each element's
    # cache and the attention runs over ALL L cache rows with the keys beyond an element's prefix masked by `kpm_prefix` (the caller's
    # [B, kpm_stride] buffer: row b = the padding flags of its prefix, 1 from pos_vec[b] + 1 on) -- masked keys contribute exact zeros, so the
    # result equals the Sk = pos + 1 call bit for bit (tests/test_model_gpu.py)
    rowidx = None
    if pos_vec is not None and not fused:
        assert kpm_stride is not None and kpm_stride >= L, "per-element positions: the key-padding buffer must cover every cache row"
        rowidx = torch.arange(B, device=tok.device) * L + pos_vec.long()
    for li, layer in enumerate(dec.transformer.layers):
        sa, ca = layer.self_attn, layer.multihead_attn
        skv, ckv = cache.self_kv[li], cache.cross_kv[li]
        if fused:
            kst = kpm_stride if kpm_stride else kpm_prefix.shape[1]
            o = ops.decode_attn(x, layer.norm1.weight.detach(), layer.norm1.bias.detach(), rt.lp(sa.in_proj_weight), sa.in_proj_bias.detach(),
                                skv, pos, H, True, kpm=kpm_prefix, kpm_stride=kst, pos=pos_vec)
            x = ops.gemm(o, rt.lp(sa.out_proj.weight), B, d, d, bias=sa.out_proj.bias.detach(), res=x)
            o = ops.decode_attn(x, layer.norm2.weight.detach(), layer.norm2.bias.detach(), rt.lp(ca.in_proj_weight), ca.in_proj_bias.detach(),
                                ckv, cache.cross_rows, H, False, packed_rows=cache.cross_rows if cache.cross_packed else 0)
            if cache.packed is not None:   # out-projection + residual + LayerNorm + feed-forward + residual in one launch (ralf_tlayer_fwd part 2)
                pk = cache.packed[li]
                x = ops.tlayer_tail(o, x, {"out2": (pk[0], ca.out_proj.bias.detach()), "ln3": (layer.norm3.weight.detach(), layer.norm3.bias.detach()),
                                           "ffn1": (pk[1], layer.linear1.bias.detach()), "ffn2": (pk[2], layer.linear2.bias.detach())})
                continue
            x = ops.gemm(o, rt.lp(ca.out_proj.weight), B, d, d, bias=ca.out_proj.bias.detach(), res=x)
        else:
            h, _, _ = ops.layernorm_fwd(x, layer.norm1.weight.detach(), layer.norm1.bias.detach(), save_stats=False)
            W, bvec = rt.lp(sa.in_proj_weight), sa.in_proj_bias.detach()
            q = ops.gemm(h, W[:d], B, d, d, bias=bvec[:d])
            if rowidx is None:
                # k,v of the new token go straight into row `pos` of the cache (row stride = L*2d)
                ops.gemm(h, W[d:], B, 2 * d, d, bias=bvec[d:], out=skv.view(B, L * 2 * d)[:, pos * 2 * d:], ldc=L * 2 * d)
                o, _ = ops.attention_fwd(q.view(B, 1, d), skv, skv, B, H, 1, pos + 1, dh, 0, 0, d, causal=False, kpm=kpm_prefix, need_lse=False, kv_rows=L, kpm_stride=kpm_stride)
            else:
                skv.view(B * L, 2 * d).index_copy_(0, rowidx, ops.gemm(h, W[d:], B, 2 * d, d, bias=bvec[d:]))
                o, _ = ops.attention_fwd(q.view(B, 1, d), skv, skv, B, H, 1, L, dh, 0, 0, d, causal=False, kpm=kpm_prefix, need_lse=False, kv_rows=L, kpm_stride=kpm_stride)
            x = ops.gemm(o.view(B, d), rt.lp(sa.out_proj.weight), B, d, d, bias=sa.out_proj.bias.detach(), res=x)
            h, _, _ = ops.layernorm_fwd(x, layer.norm2.weight.detach(), layer.norm2.bias.detach(), save_stats=False)
            q = ops.gemm(h, rt.lp(ca.in_proj_weight)[:d], B, d, d, bias=ca.in_proj_bias.detach()[:d])
            o, _ = ops.attention_fwd(q.view(B, 1, d), ckv, ckv, B, H, 1, ckv.shape[1], dh, 0, 0, d, need_lse=False)
            x = ops.gemm(o.view(B, d), rt.lp(ca.out_proj.weight), B, d, d, bias=ca.out_proj.bias.detach(), res=x)
        if rt.decode_ln_gemm and ops.gemm_ln_ok(x, B, d):   # LayerNorm inside the few-row product (every tile's wave normalises the 32 rows it has just loaded)
            f = ops.gemm(x, rt.lp(layer.linear1.weight), B, layer.linear1.weight.shape[0], d, bias=layer.linear1.bias.detach(), act="relu",
                         ln=(layer.norm3.weight.detach(), layer.norm3.bias.detach(), 1e-5))
        else:
            h, _, _ = ops.layernorm_fwd(x, layer.norm3.weight.detach(), layer.norm3.bias.detach(), save_stats=False)
            f = ops.gemm(h, rt.lp(layer.linear1.weight), B, layer.linear1.weight.shape[0], d, bias=layer.linear1.bias.detach(), act="relu")
        x = ops.gemm(f, rt.lp(layer.linear2.weight), B, d, f.shape[1], bias=layer.linear2.bias.detach(), res=x, few_row_split=rt.decode_few_row_split)
    if rt.decode_ln_gemm and ops.gemm_ln_ok(x, B, d):
        logits = ops.gemm(x, rt.lp(dec.head[1].weight), B, dec.head[1].weight.shape[0], d, out_dtype=torch.float32,
                          ln=(dec.head[0].weight.detach(), dec.head[0].bias.detach(), 1e-5))
    else:
        h, _, _ = ops.layernorm_fwd(x, dec.head[0].weight.detach(), dec.head[0].bias.detach(), save_stats=False)
        logits = ops.gemm(h, rt.lp(dec.head[1].weight), B, dec.head[1].weight.shape[0], d, out_dtype=torch.float32)
    return logits if sample is None else ops.mask_sample(logits, **sample)


class UserConstraintTransformerEncoder(nn.Module):
    def __init__(self, d_model: int, nhead: int, num_layers: int, d_label: int, dim_feedforward: int):
        super().__init__()
        self.encoder = LayerStack([TransformerEncoderLayer(d_model, nhead, dim_feedforward, 0.1, True) for _ in range(num_layers)])
        self.emb = Affine(d_label, d_model)
        self.pos_emb = PosEnc1d(d_model)

    def forward(self, src, src_key_padding_mask, rt: Runtime):
        h = RF.EmbedFn.apply(src, self.emb.weight, self.pos_emb.pe[0], rt)
        h = RF.drop_add(h, None, rt.drop_p(self.pos_emb.p), rt)
        kpm = _kpm_u8(src_key_padding_mask)
        layers = list(self.encoder.layers)
        packs = _pack_layers(layers, h, rt)
        for li, layer in enumerate(layers):
            h = layer(h, rt, kpm, packed=packs[li] if packs else None)
        return h


class _TokenCore(nn.Module):
    def __init__(self, d, nhead, dim_ff, num_layers):
        super().__init__()
        self.token = nn.Parameter(torch.randn(1, 1, d))
        self.register_buffer("token_mask", torch.zeros(1, 1, dtype=torch.bool))
        self.core = LayerStack([TransformerEncoderLayer(d, nhead, dim_ff, 0.1, False) for _ in range(num_layers)])


class LayoutEncoder(nn.Module):
    """Frozen FIDNetV3 feature extractor (fid/model.py:90-103,150-175): decoder-side modules are deleted
    by load_fidnet_feature_extractor, so only the encoder keys exist."""

    def __init__(self, num_label: int, d_model: int = 256, nhead: int = 4, num_layers: int = 4):
        super().__init__()
        self.emb_label = Affine(num_label, d_model)
        self.fc_bbox = Affine(d_model, 4, bias_dim=d_model)
        self.enc_fc_in = Affine(d_model, 2 * d_model, bias_dim=d_model)
        self.enc_transformer = _TokenCore(d_model, nhead, d_model // 2, num_layers)
        # load_fidnet_feature_extractor deletes the decoder-side modules but leaves dec_fc_in (fid/model.py:168-174):
        # it is never used, yet it is part of the reference's checkpoint layout
        self.dec_fc_in = Affine(d_model, 2 * d_model, bias_dim=d_model)
        self.d = d_model

    @torch.no_grad()
    def load_fidnet(self, dataset_name: str, ckpt_dir: str = "tmp/fidnet") -> "LayoutEncoder":
        """load_fidnet_feature_extractor (fid/model.py:131-175): the trained FIDNetV3 of `dataset_name` (pku -> pku10) from
        `<ckpt_dir>/<dataset>/model_best.pth.tar`, else from ./cache/PRECOMPUTED_WEIGHT_DIR/fidnet/...; the decoder-side keys the reference
        deletes after its strict load are dropped, anything else that does not match fails"""
        from .pretrained import fidnet_encoder_state, load_fidnet_state
        self.load_state_dict(fidnet_encoder_state(load_fidnet_state(dataset_name, ckpt_dir), self.state_dict().keys()), strict=True)
        return self.eval()

    def _frozen_operands(self, rt: Runtime, dev):
        """the zero-padded fc_bbox weight [d, 8] and the learned token in the compute dtype: the encoder is frozen
        (retrieval_augmented_autoreg.py:150-155), so they are rebuilt only when the masters change"""
        key = (self.fc_bbox.weight.data_ptr(), self.fc_bbox.weight._version, self.enc_transformer.token._version, rt.dtype, str(dev))
        hit = getattr(self, "_frozen_cache", None)
        if hit is None or hit[0] != key:
            wpad = torch.zeros(self.d, 8, dtype=torch.float32, device=dev)
            wpad[:, :4] = self.fc_bbox.weight.detach()
            tok = ops.cast(self.enc_transformer.token.detach().reshape(1, self.d).contiguous(), rt.dtype)
            self._frozen_cache = hit = (key, ops.cast(wpad, rt.dtype), tok)
        return hit[1], hit[2]

    @torch.no_grad()
    def extract_features(self, layout: dict, rt: Runtime) -> torch.Tensor:
        """layout fields [R, N] (R = B*K rows batched in ONE call instead of the reference's K-loop,
        retrieval_augmented_autoreg.py:539-568) -> [R, d]."""
        R, N = layout["label"].shape
        dev = layout["label"].device
        d = self.d
        # geometry rows (4 inputs padded to 8 columns: the bf16 operand rows are 16-byte vectors) and the sequence's padding mask: one launch
        bbox_c, kpm = ops.layout_pack(layout["center_x"], layout["center_y"], layout["width"], layout["height"], layout["mask"].bool(), rt.dtype)
        wpad, tok = self._frozen_operands(rt, dev)
        cat = torch.empty(R * N, 2 * d, dtype=rt.dtype, device=dev)
        ops.gemm(bbox_c, wpad, R * N, d, 8, bias=self.fc_bbox.bias, out=cat, ldc=2 * d)
        lab = ops.embed_fwd(layout["label"].long().contiguous(), self.emb_label.weight, None, N, 1.0, rt.dtype)
        ops.copy2d(lab, cat.view(-1)[d:], R * N, d, d, 2 * d)                 # right half of the concatenation
        xe = ops.gemm(cat, rt.lp(self.enc_fc_in.weight), R * N, d, 2 * d, bias=self.enc_fc_in.bias, act="relu")
        x = torch.empty(R, N + 1, d, dtype=rt.dtype, device=dev)              # [learned token; elements]
        ops.copy2d(tok, x, R, d, 0, (N + 1) * d)
        ops.copy2d(xe, x.view(-1)[d:], R, N * d, N * d, (N + 1) * d)
        for layer in self.enc_transformer.core.layers:
            x = layer(x, rt, kpm)
        out = torch.empty(R, d, dtype=x.dtype, device=dev)
        ops.copy2d(x, out, R, d, (N + 1) * d, d)       # the token row of every sequence
        return out


# ----------------------------------------------------------------------------------------------
# ResNet-50 / FPN backbone, NHWC
# ----------------------------------------------------------------------------------------------
class BN(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.fold = None   # (scale, shift) views of the backbone's eval-mode fold buffer (ResnetBackbone._refresh_fold)

    def forward(self, x, rt: Runtime, relu: bool, res=None, stats=None):
        """stats: column-statistics partials from the producing convolution (Conv.forward(..., stats=True))."""
        # num_batches_tracked is bumped by the statistics kernel (no separate launch per layer)
        return RF.BatchNormFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, res, relu, rt.training, rt,
                                    self.num_batches_tracked, stats if rt.training else None)


class Conv(nn.Module):
    def __init__(self, cin, cout, k, stride=1, pad=0, bias=False):
        super().__init__()
        self.weight = _param(cout, cin, k, k)
        if bias:
            self.bias = _param(cout)
        else:
            self.register_parameter("bias", None)
        self.stride, self.pad = stride, pad

    def forward(self, x, rt: Runtime, pos=None, fork=False, stats=False):
        return RF.conv2d(x, self.weight, self.bias, self.stride, self.pad, rt, pos, fork, stats)


class Bottleneck(nn.Module):
    """ResNet-v1.5 bottleneck (stride on the 3x3), keys conv1..3 / bn1..3 / downsample.{0,1}."""

    def __init__(self, inpl, planes, stride, downsample):
        super().__init__()
        self.conv1, self.bn1 = Conv(inpl, planes, 1), BN(planes)
        self.conv2, self.bn2 = Conv(planes, planes, 3, stride, 1), BN(planes)
        self.conv3, self.bn3 = Conv(planes, planes * 4, 1), BN(planes * 4)
        if downsample:
            self.downsample = nn.ModuleList([Conv(inpl, planes * 4, 1, stride, 0), BN(planes * 4)])
        else:
            self.downsample = None

    def forward_infer(self, x, rt: Runtime):
        """inference (no autograd, running statistics): each BatchNorm (+ ReLU, + the residual) is the epilogue of its convolution"""
        cb = lambda conv, bn, inp, relu, res=None: RF.conv_bn_infer(inp, conv.weight, bn.fold[0], bn.fold[1], conv.stride, conv.pad, relu, res, rt)  # noqa: E731
        y = cb(self.conv1, self.bn1, x, True)
        y = cb(self.conv2, self.bn2, y, True)
        idn = cb(self.downsample[0], self.downsample[1], x, False) if self.downsample is not None else x
        return cb(self.conv3, self.bn3, y, True, idn)

    def forward(self, x, rt: Runtime, want_alias: bool = False):
        """want_alias (blocks with a downsample branch): also return an alias of x for ONE more consumer outside the block (the FPN lateral
        of layer3's output): its gradient is summed in the downsample convolution's data-gradient epilogue, that sum in conv1's -- the block
        input's gradient leaves conv1 complete (no autograd add kernel, and its BatchNorm backward takes the fused reductions)"""
        # training: every BatchNorm's batch statistics are produced by the preceding convolution's epilogue
        y, idn, st = self.conv1(x, rt, fork=True, stats=True)   # idn aliases x: its gradient is summed inside conv1's data-gradient GEMM
        y = self.bn1(y, rt, True, stats=st)
        y, st = self.conv2(y, rt, stats=True)
        y = self.bn2(y, rt, True, stats=st)
        y, st3 = self.conv3(y, rt, stats=True)
        alias = None
        if self.downsample is not None:
            if want_alias:
                idn, alias, st = self.downsample[0](idn, rt, fork=True, stats=True)
            else:
                idn, st = self.downsample[0](idn, rt, stats=True)
            idn = self.downsample[1](idn, rt, False, stats=st)
        out = self.bn3(y, rt, True, res=idn, stats=st3)
        return (out, alias) if want_alias else out


RESNET50_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


class ResNetBody(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.bn1 = Conv(4, 64, 7, 2, 3), BN(64)
        inpl = 64
        for li, (planes, blocks, stride) in enumerate(RESNET50_STAGES, start=1):
            layer = nn.ModuleList([Bottleneck(inpl if bi == 0 else planes * 4, planes, stride if bi == 0 else 1, bi == 0) for bi in range(blocks)])
            setattr(self, f"layer{li}", layer)
            inpl = planes * 4


class ResnetBackbone(nn.Module):
    """common/image.py:27-120 with head='transformer'; returns the projected map ALREADY as the
    [B, h*w, d] sequence with the 2-D sine table added (PositionEmbeddingSine fused into proj)."""

    def __init__(self, d_model: int = 256, pretrained: bool = False):
        """pretrained=True is the reference constructor's behaviour (common/image.py:38-48,70-77): the timm ResNet-50 checkpoint is looked
        up in the working directory, then under ./cache/PRECOMPUTED_WEIGHT_DIR (AssertionError when absent) and loaded into the body; the
        generators pass it (models/ralf.py), the building-block default is off"""
        super().__init__()
        self.body = ResNetBody()
        self.fpn_conv11_4, self.fpn_conv11_5 = Conv(1024, 256, 1, bias=True), Conv(2048, 256, 1, bias=True)
        self.fpn_conv33 = Conv(256, 256, 3, 1, 1, bias=True)
        self.proj = Conv(512, d_model, 1, bias=True)
        self._pos_cache: dict = {}
        self._conv_weights = None
        self._fold_state = None
        self.pretrained = bool(pretrained)
        if pretrained:
            from .pretrained import load_resnet50_state
            self.load_pretrained_body(load_resnet50_state())

    def parameters_before_cut(self):
        """parameters whose gradients are produced AFTER the rt.grad_cut() point in the backward (stem, layer1, layer2)"""
        b = self.body
        return [p for m in (b.conv1, b.bn1, b.layer1, b.layer2) for p in m.parameters() if p.requires_grad]

    def pos_table(self, h, w, d, rt, device):
        key = (h, w, d, rt.dtype, str(device))
        if key not in self._pos_cache:
            self._pos_cache[key] = ops.cast(pos2d_sine(h, w, d).to(device), rt.dtype)
        return self._pos_cache[key]

    def _refresh_fold(self, device):
        """eval-mode scale / shift of all BatchNorm layers, recomputed by ONE launch per forward (so a captured graph always folds the
        current statistics); the job table is rebuilt only when a parameter or buffer moved (optimizer flat buffers, .to())"""
        bns = [m for m in self.body.modules() if isinstance(m, BN)]
        key = tuple(t.data_ptr() for m in bns for t in (m.weight, m.bias, m.running_mean, m.running_var))
        if self._fold_state is None or self._fold_state[0] != key:
            buf = torch.empty(2 * sum(m.weight.numel() for m in bns), dtype=torch.float32, device=device)
            table, views = ops.bn_fold_table([(m.weight.detach(), m.bias.detach(), m.running_mean, m.running_var) for m in bns], buf, device)
            for m, v in zip(bns, views):
                m.fold = v
            self._fold_state = (key, table, buf, len(bns))
        ops.bn_fold_batched(self._fold_state[1], self._fold_state[3])

    def forward(self, img: torch.Tensor, rt: Runtime) -> torch.Tensor:
        lat3, layer4 = self.body_features(img, rt)
        return self.fpn(lat3, layer4, rt)

    def body_features(self, img: torch.Tensor, rt: Runtime):
        """the body's two taps (common/image.py:66-67,96-97: layer3, layer4) as NHWC maps; the layer3 map is the alias the FPN lateral reads"""
        B, C, H, W = img.shape
        assert C == 4
        # NCHW fp32 [B,4,H,W] -> NHWC compute dtype, channels zero-padded to 8 (16-byte pixel vectors)
        b = self.body
        if self._conv_weights is None:
            self._conv_weights = [m.weight for m in self.modules() if isinstance(m, Conv)]
        rt.refresh_conv_shadows(self._conv_weights)   # all 3x3 / 7x7 weight re-layouts of this step in one launch
        infer = rt.fold_bn and not rt.training and not torch.is_grad_enabled()

        def pack(im):
            # NCHW fp32 [b,4,H,W] -> NHWC compute dtype, channels zero-padded to 8 (16-byte pixel vectors).  (An image batch that arrives in the compute dtype
            # already -- models/ralf.py: sample() with RALF_UPLOAD_LP=1, the same rounding one step earlier -- is packed as it is)
            src = im.contiguous()
            if not (src.dtype == rt.dtype == torch.bfloat16):
                src = src.float()
            return ops.permute4(src, (im.shape[0], H, W, 8), (4 * H * W, W, 1, H * W), 4, rt.dtype)
        if infer:
            self._refresh_fold(img.device)
            gates = rt.input_gates if (rt.input_gates and torch.cuda.is_current_stream_capturing() and B >= 2 * len(rt.input_gates)) else None
            step = rt.infer_chunk if (rt.infer_chunk and B > rt.infer_chunk) else 0
            bounds = rt.input_bounds(B, len(gates)) if gates else (list(range(0, B, step)) + [B] if step else None)
            if bounds:
                # The batch in slices.  Gated (engine.GraphedDecode): slice i's kernels sit behind an event-wait node for the copy of ITS images, so the host
                # link carries slice i + 1 while the backbone works on slice i (a B = 256 batch is 4.9 ms on the link, a fifth of the whole captured loop).
                # Ungated (RALF_INFER_CHUNK, off by default): samples do not interact in eval mode, and at 64 images layer1's maps (134 MB) would stay in
                # the 256 MB infinity cache between the kernel that writes them and the one that reads them (at B = 256: 537 MB each).  Measured at
                # B = 256, same box, same tokens: decode loop 26.1 ms whole / 26.3 in slices of 128 / 26.6 of 64 -- the shorter launches lose more than
                # the cache gives (profiles/r06_infer_chunk.txt)
                outs = []
                for gi in range(len(bounds) - 1):
                    if gates:
                        gates[gi].wait(torch.cuda.current_stream())
                    outs.append(self._body_infer(pack(img[bounds[gi]:bounds[gi + 1]]), rt))
                return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
            return self._body_infer(pack(img), rt)
        x = pack(img)
        if rt.training and rt.fused_stem:
            # training: BatchNorm (batch statistics) + ReLU + max-pool in one pass over the convolution output (RF.StemBNReluPoolFn)
            x, st = b.conv1(x, rt, stats=True)
            x = RF.StemBNReluPoolFn.apply(x, b.bn1.weight, b.bn1.bias, b.bn1.running_mean, b.bn1.running_var, b.bn1.num_batches_tracked, st, rt)
        else:
            x, st = b.conv1(x, rt, stats=True)
            x = b.bn1(x, rt, True, stats=st)
            x = RF.MaxPoolFn.apply(x)
        feats, lat3 = {}, None
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(b, f"layer{li}")):
                if li == 4 and bi == 0 and torch.is_grad_enabled():
                    # layer3's output has two consumers (layer4 and the FPN lateral): the lateral reads an alias handed out by layer4's first
                    # block, so both gradients meet inside that block's data-gradient epilogues instead of in an autograd add kernel
                    x, lat3 = blk(x, rt, want_alias=True)
                else:
                    x = blk(x, rt)
            if li == 2:   # data parallel: stem + layer1-2 hold 6 % of the parameters and most of the backbone's backward time
                x = rt.grad_cut(x)
            feats[li] = x
        return (lat3 if lat3 is not None else feats[3]), feats[4]

    def _body_infer(self, x: torch.Tensor, rt: Runtime):
        """the folded-BatchNorm forward of the body on NHWC pixels [b, H, W, 8] -> (layer3, layer4) maps"""
        b = self.body
        # the stem in direct form (ops.stem7x7_fwd) and ONE pass for the folded BatchNorm + ReLU + max-pool (the training forward's two kernels with the
        # eval-mode scale / shift): at B = 256 the general 8-channel gather + a separate max-pool took 633 + 174 us
        if rt.stem_direct and rt.fused_stem and x.dtype == torch.bfloat16 and tuple(b.conv1.weight.shape) == (64, 4, 7, 7):
            y, _ = ops.stem7x7_fwd(x, rt.lp(b.conv1.weight, "ohwi"), want_stats=False)
            x, _ = ops.bn_relu_maxpool_fwd(y, b.bn1.fold[0], b.bn1.fold[1])
        else:
            x = RF.conv_bn_infer(x, b.conv1.weight, b.bn1.fold[0], b.bn1.fold[1], b.conv1.stride, b.conv1.pad, True, None, rt)
            x = RF.MaxPoolFn.apply(x)
        feats = {}
        for li in (1, 2, 3, 4):
            for blk in getattr(b, f"layer{li}"):
                x = blk.forward_infer(x, rt)
            feats[li] = x
        return feats[3], feats[4]

    def fpn(self, layer3: torch.Tensor, layer4: torch.Tensor, rt: Runtime) -> torch.Tensor:
        """FPN fuse + projection (common/image.py:99-111) on NHWC maps -> [B, h*w, d] with the 2-D sine table added."""
        B = layer3.shape[0]
        f4 = self.fpn_conv11_4(layer3, rt)
        f5 = self.fpn_conv11_5(layer4, rt)
        f5u, s = RF.UpsampleAddFn.apply(f5, f4)
        c33 = self.fpn_conv33(s, rt)
        fused = RF.ConcatColsFn.apply(f5u, c33)   # channel concat
        h, w = fused.shape[1:3]
        out = self.proj(fused, rt, pos=self.pos_table(h, w, self.proj.weight.shape[0], rt, layer3.device))
        return out.view(B, h * w, -1)

    @torch.no_grad()
    def load_pretrained_body(self, state_dict: dict) -> None:
        """timm ResNet-50 checkpoint (3-channel stem, e.g. resnet50_a1_0-14fe96d1.pth) -> this body, the way the reference's
        constructor does it (common/image.py:39-48, 70-77): `fc.*` is not part of the extracted body and the 4th stem channel
        (saliency) is the mean of the RGB filters."""
        fc = {k for k in state_dict if k.startswith("fc.")}
        # the reference loads the checkpoint STRICTLY into timm's resnet50 (fc.weight / fc.bias included) before it extracts the body
        assert fc == {"fc.weight", "fc.bias"}, f"not a timm resnet50 checkpoint: classifier keys {sorted(fc)}"
        sd = {k: v for k, v in state_dict.items() if k not in fc}
        w3 = sd["conv1.weight"]
        assert w3.shape[1] == 3, "expected the pretrained 3-channel stem"
        sd["conv1.weight"] = torch.cat([w3, w3.mean(dim=1, keepdim=True)], dim=1)
        self.body.load_state_dict(sd, strict=True)


class ResnetFeatureExtractor(nn.Module):
    def __init__(self, d_model: int = 256, pretrained: bool = False):
        super().__init__()
        self.extractor = ResnetBackbone(d_model, pretrained)

    def forward(self, img, rt):
        return self.extractor(img, rt)
