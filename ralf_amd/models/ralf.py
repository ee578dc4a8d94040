"""RALF generator and the Autoreg baseline on the MI355X HIP path.

Drop-in for the classes exported by image2layout/train/models/generator.py:
  ConcateAuxilaryTaskConcateCrossAttnRetrievalAugmentedAutoreg
      image2layout/train/models/retrieval_augmented_autoreg.py:60-216,635-785,944-1033
  ConcateAuxilaryTaskAutoreg
      image2layout/train/models/autoreg.py:29-118,590-622
Same constructor keywords, method set (preprocess / train_loss / sample / optim_groups /
update_per_epoch / compute_stats / aggregate_sampling_config), attributes and state_dict layout
(SURVEY.md section 8b).  Device arithmetic runs only through libralf_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import logging
import os
import random
from typing import Iterable, Optional

import torch
import torch.nn as nn

from .. import functional as RF
from .. import nn as RN
from .. import ops
from ..functional import Runtime
from ..helpers.sampling import DECODE_SPACE_RESTRICTION, _get, forced_tokens_all, sample as sample_tokens
from ..helpers.task import COND_TYPES, cat_image, get_condition, pinned_copy_issued
from ..helpers.task_preprocessor import PREPROCESSOR

logger = logging.getLogger(__name__)
_DTYPES = {"float32": torch.float32, "fp32": torch.float32, "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}
NEG_INF = -float("inf")
_CONSTRAINT_FIRST = os.environ.get("RALF_CONSTRAINT_FIRST", "1") != "0"


def _num_classes(features) -> int:
    f = features["label"] if not hasattr(features, "num_classes") else features
    f = getattr(f, "feature", f)
    return int(f.num_classes)


class _GeneratorBase(nn.Module):
    """Host-side API shared by both generators (BaseModel, common/base_model.py:118-389)."""

    def _init_runtime(self, compute_dtype):
        dt = _DTYPES[compute_dtype] if isinstance(compute_dtype, str) else compute_dtype
        self.rt = Runtime(dt)

    # ---- torch.nn.Module plumbing -------------------------------------------------------------
    def train(self, mode: bool = True):
        super().train(mode)
        self.rt.training = mode
        return self

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def compute_dtype(self):
        return self.rt.dtype

    @property
    def special_token_ids(self):
        ids = {k: self.tokenizer.name_to_id(k) for k in self.tokenizer.special_tokens}
        ids.setdefault("mask", -1)
        return ids

    def compute_stats(self) -> None:
        logger.info("number of parameters: %.2fM", sum(p.numel() for p in self.parameters()) / 1e6)

    def update_per_epoch(self, epoch: int, freeze_dis_epoch: int, max_epoch: int) -> None:
        pass

    def aggregate_sampling_config(self, sampling_cfg, test_cfg=None):
        if test_cfg is not None and getattr(test_cfg, "cond_type", None) == "refinement":
            for name in ("mode", "offset_ratio", "lambda"):
                key = f"refine_{name}"
                if hasattr(test_cfg, key) or (isinstance(test_cfg, dict) and key in test_cfg):
                    sampling_cfg[key] = test_cfg[key]
        return sampling_cfg

    def optim_groups(self, base_lr: Optional[float] = None, weight_decay: float = 0.0, forced_no_weight_decay=None,
                     custom_lr: Optional[dict] = None) -> Iterable[dict]:
        """AdamW groups with the reference's rule (common/base_model.py:207-347): matrix weights of
        Linear / attention / Conv decay; biases, LayerNorm / BatchNorm / Embedding weights do not;
        frozen parameters are skipped; `custom_lr` maps a name prefix to its own learning rate."""
        named = {n: p for n, p in self.named_parameters() if p.requires_grad}
        emb_like = ("emb.weight", "task_emb.weight", "emb_label.weight", "emb_hybrid_ret.weight")
        decay, no_decay = set(), set()
        for n, p in named.items():
            if n.endswith("bias") or p.ndim <= 1 or n.endswith(emb_like) or (forced_no_weight_decay and n in forced_no_weight_decay):
                no_decay.add(n)
            else:
                decay.add(n)
        from ..engine import register_model
        token = register_model(self)   # extra group option (torch optimizers keep unknown keys): lets engine.GraphedAdamW find this model
        groups, taken = [], set()
        for prefix, lr in (custom_lr or {}).items():
            for names, wd in ((decay, weight_decay), (no_decay, 0.0)):
                sel = sorted(n for n in names if n.startswith(prefix))
                if sel:
                    groups.append({"params": [named[n] for n in sel], "weight_decay": wd, "lr": lr, "ralf_model": token})
                    taken.update(sel)
        for names, wd in ((decay, weight_decay), (no_decay, 0.0)):
            sel = sorted(n for n in names if n not in taken)
            if sel:
                groups.append({"params": [named[n] for n in sel], "weight_decay": wd, "lr": base_lr, "ralf_model": token})
        return groups

    def _upload_batch(self, inputs: dict, targets: dict):
        """with an engine.GraphedAdamW attached: the batch starts its way to the device HERE -- page-locked staging (the 4-channel image is
        assembled in a page-locked buffer already, helpers/task.py: cat_image), then asynchronous copies on a library-owned copy stream -- and preprocess returns DEVICE tensors, so the loop's `.to(rank)` calls (train/train.py:434-439) are no-ops.  Why: a pageable
        source makes `.to(rank)` a host-blocking staged copy on the loop's stream, which waited for the previous replay (17-35 ms per
        iteration measured).  Three sets of device buffers rotate; a set is rewritten only after the step that read it
        (_uploaded_batch_consumed)."""
        eng = getattr(self, "_engine", None)
        if eng is None or self.device.type != "cuda":
            return inputs, targets
        st = self.__dict__.setdefault("_h2d", {"sets": [dict(), dict(), dict()], "done": [None] * 3, "i": 0})
        j = st["i"] % 3
        st["i"] += 1
        bufs = st["sets"][j]
        # a plain library-owned copy stream beside the step's stream: the 67 MB (1.3 ms at the 54 GB/s of this host link) overlap the replay in
        # flight.  Measured (tools/loop_phases.py, loss_lag 1): this 16.2 ms per iteration against 15.9 for the resident batch; the copies on
        # the step's own stream 17.4 (serial); on a PRIORITISED stream 20.6 (a prioritised stream re-maps the hardware queues of the graph's
        # branches: the replay itself slows to 20 ms).  RALF_UPLOAD_STREAM = run / prio select those for A/B runs.
        which = os.environ.get("RALF_UPLOAD_STREAM", "own")
        if which == "run":
            cs = getattr(eng.engine, "_run", None) or torch.cuda.current_stream()
        else:
            cs = ops.own_stream("h2d", self.device, priority="high" if which == "prio" else None)
        if st["done"][j] is not None:
            st["done"][j].synchronize()        # (the step that read this set: three iterations back, long finished)
        # readers of this set that ran on the loop's own stream (evaluate(): the ordinary forward) are ordered before the rewrite as well; in
        # training the loop's stream is all but empty (the step runs on the engine's stream), so this costs nothing
        cs.wait_stream(torch.cuda.current_stream())
        dev = self.device

        def up(key, t):
            if not torch.is_tensor(t) or t.is_cuda:
                return t
            slot = bufs.get(key)
            if slot is None or slot[1].shape != t.shape or slot[1].dtype != t.dtype:
                slot = bufs[key] = (None if t.is_pinned() else torch.empty(t.shape, dtype=t.dtype, pin_memory=True), torch.empty(t.shape, dtype=t.dtype, device=dev))
            src = t
            if not t.is_pinned():
                if slot[0] is None:   # (the key's first tensor was page-locked already, this one is not: cat_image ran out of staging buffers)
                    slot = bufs[key] = (torch.empty(t.shape, dtype=t.dtype, pin_memory=True), slot[1])
                slot[0].copy_(t)
                src = slot[0]
            with torch.cuda.stream(cs):
                slot[1].copy_(src, non_blocking=True)
            if src is t and t.is_pinned():
                pinned.append(t)           # (possibly a cat_image staging buffer: it must not be rewritten before this copy has read it)
            return slot[1]

        def walk(prefix, tree):
            return {k: (walk(prefix + k + "/", v) if isinstance(v, dict) else up(prefix + k, v)) for k, v in tree.items()}
        pinned: list = []
        out_i, out_t = walk("i/", inputs), walk("t/", targets)
        ev = torch.cuda.Event()
        ev.record(cs)
        for t in pinned:
            pinned_copy_issued(t, ev)
        pinned.clear()   # (`walk` refers to itself: the closures form a reference cycle that only the cyclic collector frees -- it must not hold the staging buffers)
        torch.cuda.current_stream().wait_event(ev)   # (asynchronous: readers on the loop's stream are ordered after the copies)
        st["last"] = j
        return out_i, out_t

    def _start_image_upload(self, img):
        """sample(): the host image batch (268 MB of fp32 at B = 256: 4.9 ms at the 55 GB/s of the host link, a fifth of the captured decode loop) is
        copied to the device on the library's copy stream BEFORE the host-side task preprocessing (1.9 ms at B = 256) instead of after it: sample()
        32.9 -> 31.0 ms.  Opt-in (RALF_UPLOAD_LP=1, bf16 mode): as bf16 -- the backbone's first kernel rounds the pixels to bf16 anyway
        (nn.ResnetBackbone.body_features), the host copy into the page-locked staging buffer applies the same round-to-nearest-even, so the tokens do
        not change; measured no faster (Runtime.upload_lp).
        -> (device tensor, event of the copy) or None (nothing to upload)."""
        dev = self.device
        if dev.type != "cuda" or not torch.is_tensor(img) or img.is_cuda or img.dim() != 4:
            return None
        rt = self.rt
        lp = rt.upload_lp and rt.dtype == torch.bfloat16 and img.dtype == torch.float32
        cs = ops.own_stream("h2d", dev)
        src = img
        slot = None
        if lp or not img.is_pinned():
            ring = self.__dict__.setdefault("_h2d_sample", {})
            key = (tuple(img.shape), torch.bfloat16 if lp else img.dtype)
            slot = ring.get(key)
            if slot is None:
                slot = ring[key] = [torch.empty(key[0], dtype=key[1], pin_memory=True), None]
            if slot[1] is not None:
                slot[1].synchronize()      # (the previous call's copy out of this buffer)
            slot[0].copy_(img)             # (multi-threaded; with lp the fp32 -> bf16 rounding)
            src = slot[0]
        dst = torch.empty(img.shape, dtype=src.dtype, device=dev)
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            dst.copy_(src, non_blocking=True)
        dst.record_stream(cs)
        ev = torch.cuda.Event()
        ev.record(cs)
        if slot is not None:
            slot[1] = ev
        else:
            pinned_copy_issued(img, ev)    # (possibly a cat_image staging buffer: not rewritten before this copy has read it)
        return dst, ev

    def _uploaded_batch_consumed(self, event) -> None:
        st = self.__dict__.get("_h2d")
        if st is not None and "last" in st:
            st["done"][st["last"]] = event

    def zero_grad(self, set_to_none: bool = True) -> None:
        """train/train.py:440.  With an engine.GraphedAdamW attached the captured step zeroes its own flat gradient buffer and the
        parameters' .grad views must survive until the capture: nothing to do here."""
        if getattr(self, "_engine", None) is None:
            super().zero_grad(set_to_none)

    def postprocess(self, outputs: dict) -> dict:
        if "seq" in outputs:
            seq = outputs["seq"]
        else:
            logits = outputs["logits"].clone()
            logits[:, ~self.tokenizer.token_mask.to(logits.device)] = NEG_INF
            seq = torch.argmax(logits, dim=-1)
        return self.tokenizer.decode(seq)

    # ---- shared model pieces --------------------------------------------------------------------
    def _build_common(self, tokenizer, d_model, auxilary_task, use_flag_embedding, use_multitask, global_task_embedding, decoder_d_model=256, pretrained=True):
        assert auxilary_task in COND_TYPES, f"{auxilary_task=} must be one of {COND_TYPES}"
        assert d_model == 256 and decoder_d_model == 256, "the reference configs use d_model = 256"
        self.tokenizer = tokenizer
        self.d_model = d_model
        self.num_layers, self.nhead, self.dropout = 6, 8, 0.1
        self.dim_feedforward = 4 * d_model
        self.auxilary_task, self.use_multitask, self.global_task_embedding = auxilary_task, use_multitask, global_task_embedding
        assert not global_task_embedding, "global_task_embedding=True is not used by any shipped config"
        self.preprocessor = self._make_preprocessor(auxilary_task)
        self.pretrained = bool(pretrained)
        self.encoder = RN.ResnetFeatureExtractor(d_model, pretrained)   # pretrained: timm checkpoint loaded here, like common/image.py:38-48
        self.transformer_encoder = RN.LayerStack([RN.TransformerEncoderLayer(d_model, self.nhead, self.dim_feedforward, self.dropout, True)
                                                  for _ in range(self.num_layers)])
        self.decoder = RN.BaseDecoder(tokenizer.N_total, decoder_d_model, self.num_layers, self.nhead, self.dim_feedforward)
        self.user_const_encoder = RN.UserConstraintTransformerEncoder(d_model, self.nhead, self.num_layers, self.preprocessor.N_total, self.dim_feedforward)
        self.use_flag_embedding = use_flag_embedding
        if use_flag_embedding:
            self.task_emb = RN.Affine(2, 1)
            self.register_buffer("flag_img", torch.zeros(1).long())
            self.register_buffer("flag_user_const", torch.ones(1).long())

    def _make_preprocessor(self, task):
        kw = dict(tokenizer=self.tokenizer, global_task_embedding=self.global_task_embedding)
        if task == "relation":   # `relation_table` (ctor keyword, extension): in-memory table instead of the cache file
            kw["table"] = getattr(self, "relation_table", None)
        return PREPROCESSOR[task](**kw)

    def set_task_preprocessor(self, task: str) -> None:
        assert task in COND_TYPES
        if not self.use_multitask:
            return
        self.auxilary_task = task
        self.preprocessor = self._make_preprocessor(task)

    def get_random_task(self) -> str:
        tasks = ["uncond", "c", "cwh", "partial", "refinement", "relation"]
        return random.choices(tasks, weights=[1 / 12, 1 / 3, 1 / 3, 1 / 12, 1 / 3, 1 / 12])[0]

    def init_weights(self) -> None:
        """reference initialisation (retrieval_augmented_autoreg.py:171-175, common/common.py:69-82,228-236)
        on top of torch-default-like fills for the rest."""
        g = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        # what the constructor loaded from the authors' weight files stays as loaded (the reference's init_weights only touches the decoder
        # and the image transformer encoder); the frozen layout encoder is loaded AFTER this (see the RALF constructor)
        loaded = ("encoder.extractor.body.",) if getattr(self, "pretrained", False) else ()
        for name, p in self.named_parameters():
            if name.startswith(loaded):
                continue
            with torch.no_grad():
                if name.endswith("bias"):
                    p.zero_()
                elif p.ndim == 1:
                    p.fill_(1.0)
                elif name.endswith(("emb.weight", "task_emb.weight", "decoder.head.1.weight")):
                    p.normal_(0.0, 0.02, generator=g)
                elif name.endswith("emb_label.weight") or name.endswith("token"):
                    p.normal_(0.0, 1.0, generator=g)
                elif p.ndim == 4:  # conv: He (fan_out) like timm's resnet init
                    fan_out = p.shape[0] * p.shape[2] * p.shape[3]
                    p.normal_(0.0, (2.0 / fan_out) ** 0.5, generator=g)
                else:              # xavier-uniform matrices
                    a = (6.0 / (p.shape[0] + p.shape[1])) ** 0.5
                    p.uniform_(-a, a, generator=g)

    def _constraint_features(self, inputs):
        """constraint encoder (a8) on its own graph branch: 6 layers of launch-latency-bound kernels (Lc = 4 ... 63 tokens per
        sample) that depend on nothing but the token ids"""
        rt = self.rt
        with rt.branch("constraint"):
            cf = self.user_const_encoder(inputs["seq_layout_const"], inputs["seq_layout_const_pad_mask"], rt)
        return cf

    def _constraint_memory(self, img_mem, inputs, cf=None):
        rt = self.rt
        if cf is None:
            cf = self._constraint_features(inputs)
        rt.join_branch("constraint", cf)
        cf = rt.branch_cut(cf)   # (side-graph capture: the constraint encoder's backward becomes side work)
        if self.use_flag_embedding:   # cat([img_mem + task_emb[0], cf + task_emb[1]]) in one launch (retrieval_augmented_autoreg.py:1022-1028)
            return RF.concat_rows([img_mem, cf], rt, self.task_emb.weight, [0, 1])
        return RF.concat_rows([img_mem, cf], rt)

    def _image_memory(self, image):
        rt = self.rt.to(image.device)
        x = self.encoder(image, rt)  # [B, hw, d] with the 2-D sine table already added
        layers = list(self.transformer_encoder.layers)
        packs = RN.pack_ffn_layers(layers, x, rt)   # (bf16: the feed-forward halves run as one launch each on weights in fragment order)
        for li, layer in enumerate(layers):
            x = layer(x, rt, packed=packs[li] if packs else None)
        return x

    def forward(self, inputs: dict) -> dict:
        self.rt.to(inputs["seq"].device).begin_step()
        memory = self._encode_into_memory(inputs)["memory"]
        logits = self.decoder(inputs["seq"], memory, self.rt, inputs["tgt_key_padding_mask"])
        return {"logits": logits}

    def train_loss(self, inputs: dict, targets: dict, test: bool = False):
        eng = getattr(self, "_engine", None)
        if eng is not None and self.training and torch.is_grad_enabled() and not test:
            return eng.train_step(inputs, targets)   # whole optimisation step = one hipGraph replay (engine.GraphedAdamW)
        return self._train_loss(inputs, targets)

    def _train_loss(self, inputs: dict, targets: dict):
        outputs = self(inputs)
        loss = RF.XentFn.apply(outputs["logits"], targets["seq"], self.tokenizer.name_to_id("pad"), 0.1, self.rt)
        return outputs, {"nll_loss": loss}

    # ---- autoregressive sampling: KV-cached decode (default) or the reference's full-prefix recompute ----
    @torch.no_grad()
    def decode_tokens(self, enc_in: dict, cond_seq, cond_type: str, sampling_cfg, use_kv_cache: bool = True) -> torch.Tensor:
        """device part of sample(): encoder inputs (device tensors) -> generated token ids [B, 5N] (device).
        No host synchronisation inside, so the whole loop can be captured into a hipGraph (engine.GraphedDecode)."""
        dev = enc_in["image"].device
        B = enc_in["image"].size(0)
        ids = self.special_token_ids
        token_mask = self._token_mask_dev(dev)
        token_mask_u8 = self._token_mask_u8
        self.rt.to(dev).begin_step()
        memory = self._encode_into_memory(enc_in)["memory"]
        seq = torch.full((B, 1), ids["bos"], dtype=torch.long, device=dev)
        start = 0
        if cond_type == "partial":
            seq = torch.cat([seq, cond_seq[:, 1:6]], dim=1)
            start = 5
        restrict = DECODE_SPACE_RESTRICTION[cond_type]
        T = self.tokenizer.max_token_length
        name = _get(sampling_cfg, "name")
        cache = None
        fused = name in RN.ops.SAMPLING_MODES         # vocabulary mask + restriction + choice in one kernel (every sampling.py mode)
        forced_all = forced_tokens_all(cond_seq, cond_type, ids["pad"], ids["eos"], T) if fused else None
        if use_kv_cache and fused:
            # static buffers for the whole loop: the sampling kernel writes each token into its column of `seqbuf`, its
            # padding flag into `padbuf` (the self-attention's key-padding mask, read with row stride T + 1) and hands the
            # contiguous token vector to the next step -- no slice / compare / concatenate kernels between steps
            cache = RN.decoder_init_cache(self.decoder, memory, self.rt, T)
            seqbuf = torch.full((B, T + 1), ids["pad"], dtype=torch.long, device=dev)
            seqbuf[:, :start + 1] = seq
            padbuf = (seqbuf == ids["pad"]).to(torch.uint8)
            for j in range(start):   # prefix given by the condition (partial): fill the cache
                RN.decoder_step(self.decoder, seqbuf[:, j].contiguous(), j, cache, self.rt, padbuf, kpm_stride=T + 1)
            mode, k = RN.ops.SAMPLING_MODES[name], int(_get(sampling_cfg, "top_k", 1) or 1)
            temp, top_p = float(_get(sampling_cfg, "temperature", 1.0) or 1.0), float(_get(sampling_cfg, "top_p", 1.0) or 1.0)

            def loop(b0, b1, c):   # rows b0 .. b1 - 1 of the batch through all steps (views of the loop's static buffers)
                tok = seqbuf[b0:b1, start].contiguous()
                pb = padbuf[b0:b1]
                for i in range(start, T):
                    # (decoder step + decode-space mask + token choice: ONE launch on the bf16 path, ralf_decode_token with its s_* arguments)
                    tok = RN.decoder_step(self.decoder, tok, i, c, self.rt, pb, kpm_stride=T + 1,
                                          sample=dict(allowed=token_mask_u8[i], forced=forced_all[i][b0:b1] if forced_all is not None else None, mode=mode, top_k=k,
                                                      temperature=temp, seed=self.rt.seed, call_id=1000 + i, seq_col=seqbuf[b0:b1, i + 1], pad_flag_col=pb[:, i + 1],
                                                      pad_id=ids["pad"], top_p=top_p, row0=b0))
            # A decode step is a chain of ~45 launches that each occupy a fraction of the chip for 5-25 us (few-row products, the per-element
            # attention blocks); elements never interact.  `Runtime.decode_slices` > 1 decodes a large batch as that many independent chains on
            # streams of their own (parallel branches of the captured loop; same kernels, same per-row arithmetic, the same draws through
            # mask_sample's row0).  Measured SLOWER than one chain (Runtime.decode_slices): off by default.
            nsl = self.rt.decode_slices if (dev.type == "cuda" and B >= 64 * self.rt.decode_slices) else 1
            if nsl <= 1:
                loop(0, B, cache)
            else:
                cur = torch.cuda.current_stream()
                bounds = [B * s // nsl for s in range(nsl + 1)]
                streams = [ops.own_stream(("decode", s)) for s in range(nsl)]
                for s, st in enumerate(streams):
                    b0, b1 = bounds[s], bounds[s + 1]
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        loop(b0, b1, RN.DecodeCache([t[b0:b1] for t in cache.cross_kv], [t[b0:b1] for t in cache.self_kv], cache.max_len, cache.packed,
                                                    cross_rows=cache.cross_rows, cross_packed=cache.cross_packed))
                for st in streams:
                    cur.wait_stream(st)
            self.rt.advance_seed()   # on-device: the next call (or graph replay) draws different samples
            return seqbuf[:, 1:]
        if use_kv_cache:  # O(S) decoder work per sample instead of the reference's O(S^2) prefix recompute
            cache = RN.decoder_init_cache(self.decoder, memory, self.rt, T)
            for j in range(start):  # prefix given by the condition (partial): fill the cache
                RN.decoder_step(self.decoder, seq[:, j].contiguous(), j, cache, self.rt, (seq[:, :j + 1] == ids["pad"]).to(torch.uint8).contiguous())
        for i in range(start, T):
            if cache is not None:
                logits = RN.decoder_step(self.decoder, seq[:, i].contiguous(), i, cache, self.rt, (seq == ids["pad"]).to(torch.uint8).contiguous())
            else:
                logits = self.decoder(seq, memory, self.rt, seq == ids["pad"])[:, i].clone()
            if fused:
                forced = forced_all[i] if forced_all is not None else None
                nxt = RN.ops.mask_sample(logits.float(), token_mask_u8[i], forced, RN.ops.SAMPLING_MODES[name],
                                         int(_get(sampling_cfg, "top_k", 1) or 1), float(_get(sampling_cfg, "temperature", 1.0) or 1.0),
                                         self.rt.seed, 1000 + i, top_p=float(_get(sampling_cfg, "top_p", 1.0) or 1.0))
                seq = torch.cat([seq, nxt.view(B, 1)], dim=1)
                continue
            logits[:, ~token_mask[i]] = NEG_INF
            logits = restrict(i + 1, cond_seq, logits, pad_id=ids["pad"], eos_id=ids["eos"], max_length=T)
            seq = torch.cat([seq, sample_tokens(logits, sampling_cfg)], dim=1)
        self.rt.advance_seed()   # on-device: the next call (or graph replay) draws different samples
        return seq[:, 1:]

    def _token_mask_dev(self, dev):
        if getattr(self, "_token_mask_cache", None) is None or self._token_mask_cache.device != dev:
            self._token_mask_cache = self.tokenizer.token_mask.to(dev)
            self._token_mask_u8 = self._token_mask_cache.to(torch.uint8).contiguous()
        return self._token_mask_cache

    @torch.no_grad()
    def sample(self, cond, batch_size: Optional[int] = None, sampling_cfg=None, cond_type: Optional[str] = "uncond",
               return_violation: bool = False, use_backtrack: bool = True, return_decoded_cond: bool = False,
               use_kv_cache: bool = True, decoder=None, **kwargs):
        """`decoder`: optional engine.GraphedDecode that replays the captured device loop."""
        if self.use_multitask:
            self.set_task_preprocessor(cond.task)
        if cond_type == "relation" and use_backtrack:
            return self.sample_relation(cond, batch_size=batch_size, sampling_cfg=sampling_cfg, return_violation=return_violation, **kwargs)
        dev = self.device
        B = cond.image.size(0)
        if B == 1 and batch_size and batch_size > 1:
            B = batch_size
            cond.image = cond.image.expand(B, -1, -1, -1).contiguous()
        # the batch's one big tensor is on its way while the host builds the constraint sequences: slice by slice into a captured loop's own buffer
        # (engine.GraphedDecode.upload_image), else as one copy on the library's copy stream
        piped = decoder is not None and hasattr(decoder, "upload_image") and decoder.upload_image(cond.image)
        up = None if piped else self._start_image_upload(cond.image)
        enc_in, seqc = self._create_encoder_inputs(cond)
        if piped:
            enc_in = dict(enc_in, image=decoder.static_image())
        elif up is not None:
            torch.cuda.current_stream().wait_event(up[1])
            enc_in = dict(enc_in, image=up[0])
        enc_in = {k: ({kk: vv.to(dev) for kk, vv in v.items() if torch.is_tensor(vv)} if isinstance(v, dict) else (v.to(dev) if torch.is_tensor(v) else v))
                  for k, v in enc_in.items()}
        cond_seq = cond.seq.to(dev) if cond.seq is not None else None
        if decoder is not None:
            tokens = decoder(enc_in, cond_seq)
        else:
            tokens = self.decode_tokens(enc_in, cond_seq, cond_type, sampling_cfg, use_kv_cache)
        out_tokens = tokens.cpu()
        result = self.postprocess({"seq": out_tokens})
        if not return_violation:
            return result
        if cond_type == "relation":   # restriction without back-tracking: relations are only scored afterwards
            from ..helpers.relation_restriction import RelationConstraint
            rc = RelationConstraint(self.preprocessor)
            return result, self._violation_relation(result, [rc.prepare(seqc["seq"][b].cpu()) for b in range(seqc["seq"].size(0))])
        return result, self._violation(cond_type, cond, out_tokens)

    class _StepGraphs:
        """one captured hipGraph per decode position for the batch-1 KV-cached decoder step (static token / padding-mask /
        cache buffers): the relation loop replays them instead of issuing ~60 eager launches per step."""

        def __init__(self, model, T, dev):
            self.model, self.T = model, T
            self.tok = torch.zeros(1, dtype=torch.long, device=dev)
            self.kpm = torch.zeros(1, T, dtype=torch.uint8, device=dev)
            self.cache, self.graphs, self.out = None, {}, {}

        def bind(self, cache):
            if self.cache is None:
                self.cache = cache
            else:
                for dst, src in zip(self.cache.cross_kv, cache.cross_kv):
                    dst.copy_(src)
                # the fragment-order weight copies of the fused tail (Runtime.fused_decode_tail) are baked into the captured graphs as well:
                # a load_state_dict / training step between two calls must reach them (the row-major shadows are refreshed in place)
                if self.cache.packed is not None and cache.packed is not None:
                    for dl, sl in zip(self.cache.packed, cache.packed):
                        for dst, src in zip(dl, sl):
                            dst.copy_(src)

        def __call__(self, token, pos, kpm_prefix):
            m = self.model
            self.tok.copy_(token)
            self.kpm[:, : pos + 1].copy_(kpm_prefix)
            if pos not in self.graphs:
                run = lambda: RN.decoder_step(m.decoder, self.tok, pos, self.cache, m.rt, self.kpm[:, : pos + 1])  # noqa: E731
                side = ops.own_stream("capture")
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    run()
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=ops.own_stream("capture"), capture_error_mode="thread_local"):
                    self.out[pos] = run()
                self.graphs[pos] = g
            self.graphs[pos].replay()
            return self.out[pos]

    class _LockstepStep:
        """ONE captured hipGraph for the decoder step of a whole batch whose elements sit at positions of their own (decoder_step's pos_vec):
        token / position / key-padding buffers are static and filled from pinned host mirrors, the logits come back in one copy."""

        def __init__(self, model, T, dev, B):
            self.model, self.T, self.B = model, T, B
            self.tok = torch.zeros(B, dtype=torch.long, device=dev)
            self.pos = torch.zeros(B, dtype=torch.int32, device=dev)
            self.kpm = torch.ones(B, T, dtype=torch.uint8, device=dev)
            self.tok_h, self.pos_h, self.kpm_h = (torch.empty_like(t, device="cpu").pin_memory() for t in (self.tok, self.pos, self.kpm))
            self.tok_h.zero_(); self.pos_h.zero_(); self.kpm_h.fill_(1)
            self.cache, self.graph, self.out, self.out_h = None, None, None, None

        def bind(self, cache):
            if self.cache is None:
                self.cache = cache
                return
            for dst, src in zip(self.cache.cross_kv, cache.cross_kv):
                dst.copy_(src)
            for t in self.cache.self_kv:
                t.zero_()
            if self.cache.packed is not None and cache.packed is not None:   # (see _StepGraphs.bind)
                for dl, sl in zip(self.cache.packed, cache.packed):
                    for dst, src in zip(dl, sl):
                        dst.copy_(src)

        def __call__(self):
            m = self.model
            self.tok.copy_(self.tok_h, non_blocking=True)
            self.pos.copy_(self.pos_h, non_blocking=True)
            self.kpm.copy_(self.kpm_h, non_blocking=True)
            if self.graph is None:
                run = lambda: RN.decoder_step(m.decoder, self.tok, self.T - 1, self.cache, m.rt, self.kpm, kpm_stride=self.T, pos_vec=self.pos)  # noqa: E731
                side = ops.own_stream("capture")
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    run()
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=ops.own_stream("capture"), capture_error_mode="thread_local"):
                    self.out = run()
                self.graph = g
                self.out_h = torch.empty_like(self.out, device="cpu").pin_memory()
            self.graph.replay()
            self.out_h.copy_(self.out, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            return self.out_h

    class _RelationState:
        """one sample of sample_relation between two decoder steps (retrieval_augmented_autoreg.py:372-383: input_, REL_COUNT, reset_num, idx)"""
        __slots__ = ("seq", "flagged", "back_flag", "n_back", "resets", "idx", "rel", "con", "cond_seq", "done", "rng", "forced")

        def __init__(self, bos, rel, con, cond_seq, rng=random, forced=None):
            self.seq = torch.full((1, 1), bos, dtype=torch.long)
            self.flagged, self.back_flag, self.n_back, self.resets, self.idx = [], False, 0, 0, 0
            self.rel, self.con, self.cond_seq, self.done = rel, con, cond_seq, False
            self.rng = rng   # where the back-track positions come from: Python's global stream (the reference), or a generator of the sample's own
            self.forced = forced   # per step: the token DECODE_SPACE_RESTRICTION["relation"] forces (-1: free), tabulated once (helpers/sampling.forced_tokens_all)

    def _relation_advance(self, st, logits, may_draw, env) -> bool:
        """the body of the reference's `while` loop after the decoder call (retrieval_augmented_autoreg.py:394-460) for ONE sample: `logits`
        fp32 [1, V] of the sample's current prefix (modified in place).  Returns False -- with the state untouched -- when the step needs a
        draw from Python's `random` and may_draw is False (lock-step decoding: an earlier sample has not finished drawing yet); True otherwise."""
        ids, T, token_mask_h, restrict, prob_gate, sampling_cfg = env
        seq = st.seq
        n_dec = seq.size(1) - 1
        logits[:, ~token_mask_h[n_dec]] = NEG_INF
        if st.forced is not None:   # the same restriction from the per-sample table: one kept logit instead of the generic [1, V] mask arithmetic
            f = st.forced[n_dec]
            if f >= 0:
                keep = float(logits[0, f])
                logits.fill_(NEG_INF)
                logits[0, f] = keep
        else:
            logits = restrict(n_dec + 1, st.cond_seq, logits, pad_id=ids["pad"], eos_id=ids["eos"], max_length=T)
        raw = logits.clone()
        mask, back_idx = st.con(seq, st.rel)
        logits[:, mask] = NEG_INF
        pruned_max = torch.where(logits < prob_gate, torch.full_like(logits, NEG_INF), logits).max()
        if st.resets > 3:
            logits, st.back_flag = raw, False
        elif (not st.back_flag and bool(pruned_max == NEG_INF)) or bool(logits.max() == NEG_INF):
            # (the reference appends idx to its flag list first and then asks whether it is there fewer than 3 times)
            draw = not (back_idx is not None and st.flagged.count(st.idx) + 1 < 3)
            if draw and not may_draw:
                return False
            st.flagged.append(st.idx)
            st.back_flag = True
            st.idx = st.rng.randint(2, max(2, st.idx - 1)) if draw else back_idx
            st.seq = seq[:, :st.idx]
            st.n_back += 1
            if st.n_back > 30:
                st.flagged, st.back_flag, st.n_back = [], False, 0
                st.resets += 1
                st.seq = torch.full((1, 1), ids["bos"], dtype=torch.long)
                st.idx = 0
            return True
        temperature = None
        if st.back_flag:
            st.back_flag, temperature = False, 1.5
        nxt = sample_tokens(logits, sampling_cfg, temperature=temperature)
        st.seq = torch.cat([seq, nxt], dim=1)
        if int(nxt) == ids["eos"] or st.seq.size(1) == T + 1:
            st.done = True
        else:
            st.idx += 1
        return True

    def _relation_lockstep(self, states, memory, env, dev, min_active, independent=False):
        """All samples step TOGETHER while enough of them can: one decoder launch chain per step for the whole batch (every element at its own
        position), the masks and the control flow per sample on the host.  The reference's only cross-sample coupling is the global `random`
        stream, consumed in sample order: a sample whose step needs a draw waits ("parks", its state untouched) until every earlier sample has
        finished -- so the stream is consumed exactly as in the sequential loop and the tokens are the same.  Returns the batch cache (the
        unfinished samples' prefixes live in it).
        independent (rng="per_sample"): every sample draws from a generator of its own, so nobody waits and the loop runs until all are done."""
        ids, T = env[0], env[1]
        B = len(states)
        pool = self.__dict__.setdefault("_relation_lockstep_steps", {})
        key = (B, int(memory.shape[1]), str(dev))
        if key not in pool:
            while len(pool) >= 2:
                pool.pop(next(iter(pool)))
            pool[key] = self._LockstepStep(self, T, dev, B)
        step = pool.pop(key)
        pool[key] = step
        step.bind(RN.decoder_init_cache(self.decoder, memory, self.rt, T))
        step.kpm_h.fill_(1)
        active = list(range(B))
        first_open = 0                                   # lowest sample that has not finished: the only one allowed to draw
        while len(active) >= min_active:
            for b in active:
                seq = states[b].seq
                L = seq.size(1)
                step.tok_h[b] = seq[0, L - 1]
                step.pos_h[b] = L - 1
                row = step.kpm_h[b]
                row[:L] = (seq[0] == ids["pad"]).to(torch.uint8)
                row[L:] = 1
            logits = step()
            still = []
            for b in active:
                while first_open < B and states[first_open].done:
                    first_open += 1
                st = states[b]
                if not self._relation_advance(st, logits[b:b + 1].clone(), independent or b == first_open, env):
                    continue                             # parked: resumes in the sequential phase, after every earlier sample
                if not st.done:
                    still.append(b)
            while first_open < B and states[first_open].done:
                first_open += 1
            if not independent and first_open < B and (not still or still[0] != first_open):
                still.insert(0, first_open)              # a parked sample whose turn to draw has come steps with the others again
            active = still
        return step.cache

    def _relation_lockstep_batched(self, states, memory, env, dev, shared: bool = False, serial: bool = False):
        """The relation decode of a whole batch in LOCK-STEP for argmax decoding: one batched decoder step per iteration (every sample at a
        position of its own), the masks (tokenizer slot, forced label, relation restriction), the gate and the arg-max of ALL samples taken on
        one [B, V] numpy array; per sample only the restriction's interval logic (RelationConstraint.step, no tensors) and the back-track
        bookkeeping on plain ints stay in Python.  Same masks, same comparisons, same draws per sample as _relation_advance.

        MEMO (round 6).  Argmax decoding is deterministic: the logits of a sample depend on its prefix alone, and back-tracking re-decodes the
        same few prefixes again and again (a sample of the benchmark's workload takes 150-500 steps over a few dozen distinct prefixes).  Every
        logits row that came back from the device is kept per (sample, prefix); a step whose prefix is known costs no device work, and an
        iteration in which every stepping sample hits its memo replays nothing at all.  `devtok[b]` records which tokens' keys / values the
        device cache of sample b holds; when a NEW prefix must be decoded and the cache was last written by another branch, the positions
        from the first difference on are re-fed first (their rows go to the memo as well).

        shared = True: the EXACT mode, the reference's draw order.  All samples draw from Python's global `random`, consumed sample after
        sample as in the sequential loop (retrieval_augmented_autoreg.py:437-443), and still nobody waits for a sample that never draws a VALUE:
          * `random.randint(2, max(2, idx - 1))` with idx <= 3 has ONE possible result, so the sample takes 2 at once and only the CONSUMPTION
            of the stream (randint(2, 2) loops over 1-bit draws: a stream-dependent number of words) is deferred: `pending[b]` counts those
            draws and catch_up() replays them on the global stream in sample order, as soon as every lower-indexed sample has finished;
          * a draw with more than one possible result parks its sample (state untouched: the step is repeated later from the memo) until it is
            the lowest unfinished sample -- then every draw before it in the reference's order has been consumed and it draws directly.
        Tokens and the final state of `random` are those of the sequential loop by construction (tests/test_relation_cpu.py on a fake decoder,
        tests/test_relation_gpu.py).  serial = True (test aid): only the lowest unfinished sample steps -- the sequential ORDER on the batched
        step's arithmetic.
        shared = False (rng="per_sample"): every sample draws from its own generator; nobody parks."""
        import numpy as np

        ids, T, token_mask_h, _restrict, prob_gate, _cfg = env
        B = len(states)
        pool = self.__dict__.setdefault("_relation_lockstep_steps", {})
        key = (B, int(memory.shape[1]), str(dev))
        if key not in pool:
            while len(pool) >= 2:
                pool.pop(next(iter(pool)))
            pool[key] = self._LockstepStep(self, T, dev, B)
        step = pool.pop(key)
        pool[key] = step
        step.bind(RN.decoder_init_cache(self.decoder, memory, self.rt, T))
        tok_h, pos_h, kpm_h = step.tok_h.numpy(), step.pos_h.numpy(), step.kpm_h.numpy()   # (views of the pinned mirrors)
        pad, eos, bos = int(ids["pad"]), int(ids["eos"]), int(ids["bos"])
        tok_h[:] = bos
        pos_h[:] = 0
        kpm_h[:] = 1
        kpm_h[:, 0] = int(bos == pad)
        tmask = token_mask_h.numpy()                                     # [positions, V]: what the tokenizer admits at a position
        forced = np.asarray([st.forced for st in states], dtype=np.int64)   # [B, T]: the forced label of a step, -1 = free
        start = states[0].con.start
        seqs = [[bos] for _ in range(B)]
        # per sample: prefix (tuple of tokens) -> the DECISION of the step at that prefix, everything the control flow needs of the logits:
        # (max admissible logit, its token, the arg-max without the relation mask, back-track target, the constraint's state after the step)
        memo = [dict() for _ in range(B)]
        devtok = [[] for _ in range(B)]              # per sample: the tokens whose keys / values sit in the device cache, by position
        active = [0] if (shared and serial) else list(range(B))
        pending, first_open = [0] * B, 0
        parked = set(range(1, B)) if (shared and serial) else set()
        memo_drop = int(os.environ.get("RALF_RELATION_MEMO_DROP", "0"))
        self.relation_stats = stats = {"iterations": 0, "device_steps": 0, "rows_from_device": 0, "rows_from_memo": 0, "refed_positions": 0}

        def catch_up():
            """consume, in sample order, the range-one draws of every sample below (and of) the lowest unfinished one"""
            nonlocal first_open
            while first_open < B:
                for _ in range(pending[first_open]):
                    random.randint(2, 2)
                pending[first_open] = 0
                if not states[first_open].done:
                    break
                first_open += 1
        while active or parked:
            if shared:
                catch_up()
                if first_open in parked:             # its turn: every earlier draw of the reference's order has been consumed
                    parked.discard(first_open)
                    active = sorted(active + [first_open])
                assert active, "relation lock-step: parked samples but nobody to step"
            stats["iterations"] += 1
            if memo_drop and stats["iterations"] % memo_drop == 0:       # (test aid: forget every decision -- the caches now hold other branches' tokens)
                for m_ in memo:
                    m_.clear()
            # ---- device work: only for prefixes this sample has not decoded before
            need = []
            for b in active:
                sq = seqs[b]
                if tuple(sq) in memo[b]:
                    continue
                dt, L = devtok[b], len(sq)
                q = 0
                while q < L - 1 and q < len(dt) and dt[q] == sq[q]:
                    q += 1
                # q == L - 1: the cache holds the keys / values of sq[:-1] -> feed the last token; else re-feed position q first
                stats["refed_positions"] += int(q < L - 1)
                need.append((b, q))
                tok_h[b], pos_h[b] = sq[q], q
                kpm_h[b, :q + 1] = [int(t == pad) for t in sq[:q + 1]]
                kpm_h[b, q + 1:] = 1
            if need:
                stats["device_steps"] += 1
                stats["rows_from_device"] += len(need)
                out = step().numpy()
                for b, q in need:
                    devtok[b] = seqs[b][:q + 1]
                fresh = [b for b, q in need if q == len(seqs[b]) - 1]    # rows of the samples' CURRENT prefixes (the others re-fed an older position)
                if fresh:
                    n_dec = np.fromiter((len(seqs[b]) - 1 for b in fresh), np.int64, len(fresh))
                    lg = out[fresh]                                          # (a copy: fancy indexing)
                    lg[~tmask[n_dec]] = NEG_INF
                    f = forced[fresh, n_dec]
                    r = np.nonzero(f >= 0)[0]
                    if r.size:                                               # one kept logit, as _relation_advance
                        keep = lg[r, f[r]]
                        lg[r] = NEG_INF
                        lg[r, f[r]] = keep
                    rawarg = lg.argmax(axis=1)                               # (what `resets > 3` decodes: the relation mask dropped)
                    allow = np.zeros(lg.shape, dtype=bool)
                    aux = []
                    for i, b in enumerate(fresh):
                        st, sq = states[b], seqs[b]
                        n = len(sq) - 1
                        what, back = st.con.step(n, sq[-1], st.rel)
                        h = st.con.history
                        aux.append((back, h[n + 1] if len(h) > n + 1 else None))
                        if what[0] == "only":
                            allow[i, what[1]] = True
                        elif what[0] == "bins":
                            lo, hi = what[2]
                            if hi > lo:
                                allow[i, start[what[1]] + int(lo):start[what[1]] + int(hi)] = True
                        else:
                            allow[i] = tmask[what[1]]
                    lg[~allow] = NEG_INF
                    rowmax, arg = lg.max(axis=1), lg.argmax(axis=1)
                    for i, b in enumerate(fresh):
                        memo[b][tuple(seqs[b])] = (float(rowmax[i]), int(arg[i]), int(rawarg[i]), aux[i][0], aux[i][1])
            # ---- control flow on plain numbers
            still = []
            for b in active:
                st, sq = states[b], seqs[b]
                d = memo[b].get(tuple(sq))
                if d is None:                        # still re-feeding an older branch's positions
                    still.append(b)
                    continue
                rowmax_b, arg_b, rawarg_b, back_idx, cstate = d
                n = len(sq) - 1
                h = st.con.history                   # the constraint's history follows the prefix (RelationConstraint.step appends one state per token)
                del h[n + 1:]
                if cstate is not None:
                    h.append(cstate)
                stats["rows_from_memo"] += 1
                if st.resets > 3:
                    st.back_flag = False
                    nxt = rawarg_b
                elif (not st.back_flag and not rowmax_b >= prob_gate) or rowmax_b == NEG_INF:
                    draw = not (back_idx is not None and st.flagged.count(st.idx) + 1 < 3)
                    pos = back_idx
                    if draw and not shared:
                        pos = st.rng.randint(2, max(2, st.idx - 1))
                    elif draw:
                        hi = max(2, st.idx - 1)
                        catch_up()
                        if b == first_open:          # (pending[b] is 0 now: its deferred draws went first)
                            pos = random.randint(2, hi)
                        elif hi == 2 and not serial:
                            pending[b] += 1
                            pos = 2
                        else:
                            parked.add(b)            # state untouched: this step is taken again when b is the lowest unfinished sample
                            continue
                    st.flagged.append(st.idx)
                    st.back_flag = True
                    st.idx = pos
                    del sq[st.idx:]
                    st.n_back += 1
                    if st.n_back > 30:
                        st.flagged, st.back_flag, st.n_back = [], False, 0
                        st.resets += 1
                        sq[:] = [bos]
                        st.idx = 0
                    still.append(b)
                    continue
                else:
                    st.back_flag = False                                 # (its temperature does not move an argmax)
                    nxt = arg_b
                sq.append(nxt)
                if nxt == eos or len(sq) == T + 1:
                    st.done = True
                else:
                    st.idx += 1
                    still.append(b)
            active = still
            if shared and serial:                    # (only the lowest unfinished sample ever steps)
                parked.update(b for b in active if b != first_open and not states[b].done)
                active = [b for b in active if b not in parked]
        if shared:
            catch_up()
            assert first_open == B and not any(pending)
        for st, sq in zip(states, seqs):
            st.seq = torch.tensor([sq], dtype=torch.long)
        return step.cache

    @torch.no_grad()
    def sample_relation(self, cond, batch_size: Optional[int] = None, sampling_cfg=None, return_violation: bool = False,
                        prob_gate: float = 0.3, RELATION_SIZE: int = 10, use_graph: bool = True, lockstep: Optional[bool] = None, rng: str = "shared",
                        **kwargs):
        """Relation-constrained decoding with back-tracking (retrieval_augmented_autoreg.py:336-507): per sample, every step
        masks the vocabulary by (token mask, forced label, relation constraints); when nothing admissible is left (or only
        logits below `prob_gate`), the prefix is cut back to the element the violated constraint refers to (random position
        after three failures at one step), the next draw uses temperature 1.5; > 30 back-tracks restart the sample and after
        the 4th restart the relation mask is dropped.  Same control flow and `random` consumption as the reference; the
        decoder runs KV-cached on the device (a cut prefix just rewinds the cache position).
        lockstep (default for argmax decoding of >= 2 samples on the GPU; RALF_RELATION_LOCKSTEP=0 / lockstep=False keep the sample-after-sample
        loop): the samples step together, one batched decoder step per iteration, and `random` is still consumed in the reference's order
        (_relation_lockstep_batched: range-one draws deferred, other draws wait for the lower-indexed samples, every (sample, prefix) decoded
        once).  Tokens and the final state of `random` are identical either way (tests/test_relation_cpu.py, test_relation_gpu.py,
        test_configs_gpu.py); B = 256 on the benchmark's workload: 27.9 s -> 1.6 s per batch.
        rng = "per_sample" (opt-in THROUGHPUT mode, NOT the reference's draw order): every sample draws its back-track positions from a
        generator of its own, so no sample waits for another and the whole batch decodes in lock-step, one batched decoder step per token
        (B = 256: 28 s -> 0.97 s per batch; argmax decoding takes the masks of all samples on one array, _relation_lockstep_batched).  Sample 0 continues Python's global stream -- a batch of one decodes exactly as the sequential
        loop does -- and leaves it where it stopped; sample b > 0 is seeded from that stream's state and b.  Same masks, same control flow per
        sample, same distribution of the draws; only WHICH random number a sample sees differs from the reference."""
        assert rng in ("shared", "per_sample")
        from ..helpers.relation_restriction import RelationConstraint

        self.preprocessor.set_relation_size(RELATION_SIZE)
        dev = self.device
        B = cond.image.size(0)
        if B == 1 and batch_size and batch_size > 1:
            B = batch_size
            cond.image = cond.image.expand(B, -1, -1, -1).contiguous()
        ids = self.special_token_ids
        T = self.tokenizer.max_token_length
        enc_in, seqc = self._create_encoder_inputs(cond)
        enc_dev = {k: ({kk: vv.to(dev) for kk, vv in v.items() if torch.is_tensor(vv)} if isinstance(v, dict) else (v.to(dev) if torch.is_tensor(v) else v))
                   for k, v in enc_in.items()}
        self.rt.to(dev).begin_step()
        memory = self._encode_into_memory(enc_dev)["memory"]
        cond_seq = cond.seq.cpu()
        env = (ids, T, self.tokenizer.token_mask.cpu(), DECODE_SPACE_RESTRICTION["relation"], prob_gate, sampling_cfg)
        if lockstep is None:
            # a stochastic draw consumes torch's generator at every step, in sample order: only argmax decoding leaves `random` as the one
            # shared stream, which the lock-step loop consumes in the reference's order (round 6: on by default -- range-one draws no longer
            # serialise the batch, _relation_lockstep_batched; RALF_RELATION_LOCKSTEP=0 keeps the sample-after-sample loop)
            lockstep = (os.environ.get("RALF_RELATION_LOCKSTEP", "1") != "0" and B >= 2 and dev.type == "cuda" and _get(sampling_cfg, "name") == "deterministic")
        independent = rng == "per_sample"
        gens = [random] * B
        if independent:
            g0 = random.Random()
            g0.setstate(random.getstate())                      # sample 0 continues the global stream ...
            probe = random.Random()
            probe.setstate(random.getstate())
            base = probe.getrandbits(64)                        # ... the others are seeded from its state (a copy draws: the stream itself is not consumed)
            gens = [g0] + [random.Random((base + 0x9E3779B97F4A7C15 * b) & (2 ** 64 - 1)) for b in range(1, B)]
            lockstep = dev.type == "cuda" and B > 1
        states = []
        forced_tab = forced_tokens_all(cond_seq, "relation", ids["pad"], ids["eos"], T)   # [T, B]: step i + 1 of sample b
        forced_tab = forced_tab.t().tolist() if forced_tab is not None else [None] * B
        for b in range(B):   # (a constraint object keeps the decode history of ITS sample)
            con = RelationConstraint(self.preprocessor)
            states.append(self._RelationState(ids["bos"], con.prepare(seqc["seq"][b].cpu()), con, cond_seq[b:b + 1], gens[b], forced_tab[b]))
        batch_cache = None
        if lockstep and _get(sampling_cfg, "name") == "deterministic" and forced_tab[0] is not None and os.environ.get("RALF_RELATION_BATCHED", "1") != "0":
            batch_cache = self._relation_lockstep_batched(states, memory, env, dev, shared=not independent,
                                                          serial=os.environ.get("RALF_RELATION_SERIAL", "0") == "1")
        elif lockstep:
            batch_cache = self._relation_lockstep(states, memory, env, dev, 1 if independent else max(2, B // 32), independent)
        stepper = None
        if use_graph and not all(st.done for st in states):
            # the captured steps hold the cross-attention cache of ONE memory length (2 h w + K + Lc, and Lc is padded per batch):
            # one set of graphs per length, the few most recent kept
            pool = self.__dict__.setdefault("_relation_steppers", {})
            key = (int(memory.shape[1]), str(dev))
            if key not in pool:
                while len(pool) >= 4:
                    pool.pop(next(iter(pool)))    # least recently USED (a hit re-inserts its key below)
                pool[key] = self._StepGraphs(self, T, dev)
            stepper = pool.pop(key)
            pool[key] = stepper
        # the sequence, the masks and the draw live on the HOST (518 logits per step come back in one copy): the reference's
        # control flow is host logic anyway, and a dozen tiny device ops + three syncs per step cost more than the decoder step
        rows = []
        for b, st in enumerate(states):
            if not st.done:
                cache = RN.decoder_init_cache(self.decoder, memory[b:b + 1].contiguous(), self.rt, T)
                if stepper is not None:
                    stepper.bind(cache)
                    cache = stepper.cache
                if batch_cache is not None:       # the prefix decoded in lock-step: its keys / values are rows of the batch cache
                    for dst, src in zip(cache.self_kv, batch_cache.self_kv):
                        dst.copy_(src[b:b + 1])
            while not st.done:
                seq = st.seq
                L = seq.size(1)
                kpm = (seq == ids["pad"]).to(torch.uint8)
                if stepper is not None:
                    logits = stepper(seq[:, L - 1].to(dev), L - 1, kpm.to(dev)).float().cpu()
                else:
                    logits = RN.decoder_step(self.decoder, seq[:, L - 1].to(dev).contiguous(), L - 1, cache, self.rt, kpm.to(dev).contiguous()).float().cpu()
                self._relation_advance(st, logits, True, env)
            # (sic) finished sequences are padded with the literal True (= token 1), retrieval_augmented_autoreg.py:475-483
            rows.append(torch.cat([st.seq, torch.full((1, T + 2 - st.seq.size(1)), 1, dtype=torch.long)], dim=1))
        if independent:
            random.setstate(gens[0].getstate())                 # (what the sequential loop would have left behind after ITS first sample)
        prepared = [st.rel for st in states]
        tokens = torch.cat(rows, dim=0)[:, 1:-1]
        result = self.postprocess({"seq": tokens})
        if not return_violation:
            return result
        return result, self._violation_relation(result, prepared)

    @staticmethod
    def _violation_relation(result: dict, prepared: list) -> dict:
        """layoutformerpp/violate.py:143-236: re-detect every prepared relation on the decoded boxes."""
        from ..helpers.relationships import (RelLoc, RelSize, detect_loc_relation_between_element_and_canvas, detect_loc_relation_between_elements,
                                             detect_size_relation)

        total = bad = 0
        for b, cons in enumerate(prepared):
            box = lambda i: [result[k][b, i].item() for k in ("center_x", "center_y", "width", "height")]  # noqa: E731
            bad_b = 0
            for i, lst in enumerate(cons):
                for kind, arg in lst:
                    total += 1
                    if kind == "canvas":
                        if detect_loc_relation_between_element_and_canvas(box(i)) != arg:
                            bad += 1
                    else:
                        found = detect_size_relation(box(i), box(arg)) if isinstance(kind, RelSize) else detect_loc_relation_between_elements(box(i), box(arg))
                        if found != kind:
                            bad_b += 1
            bad += bad_b
        return {"total": total, "viorated": bad}

    def _violation(self, cond_type, cond, out_tokens) -> dict:
        """layoutformerpp/violate.py:24-141 for the table-free tasks."""
        if cond_type in ("uncond", "none", None, "partial"):
            return {"total": 1, "viorated": 0}
        pad, eos = self.tokenizer.name_to_id("pad"), self.tokenizer.name_to_id("eos")
        total = bad = 0
        cs, cm = cond.seq[:, 1:].cpu(), cond.mask[:, 1:].cpu()
        for b in range(cs.size(0)):
            given = cs[b][cm[b]]
            given = given[(given != pad) & (given != eos)]
            if cond_type == "refinement":
                got, given = out_tokens[b][: given.size(0)][::5], given[::5]
            else:
                got = out_tokens[b][cm[b]]
                got = got[(got != pad) & (got != eos)]
            assert given.size(0) == got.size(0)
            bad += int((given != got).sum())
            total += given.size(0)
        return {"total": total, "viorated": bad}

    def _create_encoder_inputs(self, cond):
        seqc = self.preprocessor(cond)
        enc = {"image": cond.image, "seq_layout_const": seqc["seq"], "seq_layout_const_pad_mask": seqc["pad_mask"]}
        if getattr(cond, "retrieved", None):
            enc["retrieved"] = cond.retrieved
        return enc, seqc


class ConcateAuxilaryTaskConcateCrossAttnRetrievalAugmentedAutoreg(_GeneratorBase):
    """Final RALF architecture ("# Final architecture", retrieval_augmented_autoreg.py:997-1033)."""

    def __init__(self, features, tokenizer, dataset_name: str, max_seq_length: int, db_dataset=None, d_model: int = 256,
                 encoder_pos_emb: str = "sine", decoder_pos_emb: str = "layout", weight_init: bool = True, top_k: int = 16,
                 layout_backbone: str = "feature_extractor", use_reference_image: bool = False, freeze_layout_encoder: bool = True,
                 retrieval_backbone: str = "saliency", random_retrieval: bool = False, saliency_k=8, decoder_d_model: int = 256,
                 auxilary_task: Optional[str] = None, use_flag_embedding: bool = True, use_multitask: bool = False, RELATION_SIZE: int = 10,
                 shared_embedding: bool = False, global_task_embedding: bool = False, compute_dtype="float32", relation_table=None,
                 pretrained: bool = True, **_ignored):
        """pretrained (extension, default = the reference's behaviour): load `resnet50_a1_0-14fe96d1.pth` into the ResNet-50 body
        (common/image.py:38-48,70-77) and `fidnet/<dataset>/model_best.pth.tar` into the frozen layout encoder (fid/model.py:131-175,
        retrieval_augmented_autoreg.py:144-155) from the reference's locations; missing files fail like the reference's.  pretrained=False
        (tests, smoke, bench: no weight files on the box) keeps the random initialisation."""
        super().__init__()
        self.relation_table = relation_table   # extension: in-memory relationship table (default: the reference's cache file)
        assert encoder_pos_emb == "sine" and decoder_pos_emb == "layout" and not use_reference_image and not shared_embedding
        assert freeze_layout_encoder is True and saliency_k != "dynamic"
        self._init_runtime(compute_dtype)
        self.features = features
        self.dataset_name, self.max_seq_length, self.top_k = dataset_name, max_seq_length, top_k
        self.retrieval_backbone, self.random_retrieval, self.saliency_k = retrieval_backbone, random_retrieval, saliency_k
        self.use_reference_image, self.layout_backbone, self.weight_init = use_reference_image, layout_backbone, weight_init
        self._build_common(tokenizer, d_model, auxilary_task, use_flag_embedding, use_multitask, global_task_embedding, decoder_d_model, pretrained)
        self.layout_encoer = RN.LayoutEncoder(_num_classes(features))  # [sic] checkpoint key of the reference
        self.pos_emb_1d = RN.PosEnc1d(d_model, 5000)
        self.layout_adapter = RN.FeedForward(256, 4 * d_model, d_model)
        self.head = RN.FeedForward(d_model, 4 * d_model)
        self.attn = RN.FuseAttention(d_model, d_model, heads=8, dim_head=64)
        self.init_weights()
        if pretrained:   # after init_weights(): the trained FIDNetV3 encoder is what gets frozen, not a re-drawn one
            self.layout_encoer.load_fidnet(dataset_name)
        for p in self.layout_encoer.parameters():
            p.requires_grad = False

    def preprocess(self, inputs: dict):
        """host path a12 (retrieval_augmented_autoreg.py:764-785,509-523)."""
        if self.use_multitask:
            self.set_task_preprocessor(self.get_random_task())
        cond, inputs = get_condition(inputs, self.auxilary_task, self.tokenizer)
        seqc = self.preprocessor(cond)
        data = self.tokenizer.encode(inputs)
        image = cond.image if torch.is_tensor(getattr(cond, "image", None)) and cond.image.size(1) == 4 else cat_image(inputs["image"], inputs["saliency"])   # (get_condition already built the 4-channel image: 67 MB per batch)
        assert inputs["retrieved"]["image"].size(2) == 4
        _inputs = {"seq": data["seq"][:, :-1], "tgt_key_padding_mask": ~data["mask"][:, :-1], "image": image,
                   "retrieved": inputs["retrieved"], "seq_layout_const": seqc["seq"], "seq_layout_const_pad_mask": seqc["pad_mask"]}
        return self._upload_batch(_inputs, {"seq": data["seq"][:, 1:]})

    def _retrieved_features(self, retrieved: dict, device) -> torch.Tensor:
        """extract_retrieved_features (retrieval_augmented_autoreg.py:526-584): frozen layout encoder over
        all B*K exemplars in one batch -> layout_adapter -> x*sqrt(d) + PE[0:K] (+dropout)."""
        rt = self.rt
        K = self.top_k
        B = retrieved["label"].shape[0]
        flat = {k: retrieved[k][:, :K].reshape(B * K, -1).to(device) for k in ("label", "mask", "center_x", "center_y", "width", "height")}
        f = self.layout_encoer.extract_features(flat, rt)                    # [B*K, 256], no grad
        f = self.layout_adapter(f, rt).view(B, K, -1)
        return RF.ScalePEDropFn.apply(f, self.pos_emb_1d.pe[0, :K].contiguous(), self.d_model ** 0.5, rt.drop_p(self.pos_emb_1d.p), rt)

    def _encode_into_memory(self, inputs: dict) -> dict:
        rt = self.rt.to(inputs["image"].device)
        assert inputs["image"].size(1) == 4
        # three independent sub-networks; the two small ones run on their own graph branches (Runtime.branch: issued late,
        # started with the step)
        # The constraint branch is recorded FIRST: its six launch-latency-bound layers then sit at the head of the captured graph and start with the
        # step (recorded after the image encoder, the graph executor started them ~0.8 ms into the replay: they, not the fusion head, gated the
        # decoder), and autograd reaches the branch's backward LAST, after it has issued the image encoder's -- recorded last, its ~50 tiny kernels
        # per layer were enqueued in front of the fusion head's and the image encoder's backward and held those back by ~0.3 ms
        # (profiles/r05_graph_timeline_*: -0.11 ms per step).  RALF_CONSTRAINT_FIRST=0: the round-4 order.
        cf = self._constraint_features(inputs) if _CONSTRAINT_FIRST else None
        mem = self._image_memory(inputs["image"])
        with rt.branch("retrieved"):
            ref = self._retrieved_features(inputs["retrieved"], inputs["image"].device)
        rt.join_branch("retrieved", ref)
        mem_a, mem_b = RF.fork2(mem)     # two consumers each: the fusion attention and the concatenation
        ref_a, ref_b = RF.fork2(ref)
        ca = self.attn(mem_a, ref_a, rt)
        fused = self.head(RF.concat_rows([mem_b, ca, ref_b], rt), rt)
        return {"memory": self._constraint_memory(fused, inputs, cf)}


class ConcateAuxilaryTaskAutoreg(_GeneratorBase):
    """Autoreg baseline without retrieval (autoreg.py:29-118, 466-622): BASELINE config 1."""

    def __init__(self, features, tokenizer, d_model: int = 256, encoder_pos_emb: str = "sine", decoder_pos_emb: str = "layout",
                 weight_init: bool = False, shared_embedding: bool = False, decoder_num_layers: int = 6, decoder_d_model: int = 256,
                 auxilary_task: Optional[str] = None, use_flag_embedding: bool = True, use_multitask: bool = False, RELATION_SIZE: int = 10,
                 global_task_embedding: bool = False, compute_dtype="float32", relation_table=None, pretrained: bool = True, **_ignored):
        """pretrained (extension, default = the reference's behaviour, autoreg.py:29-118 -> common/image.py:38-48): the ResNet-50 body is
        loaded from `resnet50_a1_0-14fe96d1.pth`; pretrained=False keeps the random initialisation (tests, bench)."""
        super().__init__()
        self.relation_table = relation_table
        assert encoder_pos_emb == "sine" and decoder_pos_emb == "layout" and not shared_embedding and decoder_num_layers == 6
        self._init_runtime(compute_dtype)
        self.features = features
        self._build_common(tokenizer, d_model, auxilary_task, use_flag_embedding, use_multitask, global_task_embedding, decoder_d_model, pretrained)
        self.init_weights()

    def preprocess(self, inputs: dict):
        if self.use_multitask:
            self.set_task_preprocessor(self.get_random_task())
        cond, inputs = get_condition(inputs, self.auxilary_task, self.tokenizer)
        seqc = self.preprocessor(cond)
        data = self.tokenizer.encode(inputs)
        image = cond.image if torch.is_tensor(getattr(cond, "image", None)) and cond.image.size(1) == 4 else cat_image(inputs["image"], inputs["saliency"])   # (get_condition already built the 4-channel image: 67 MB per batch)
        _inputs = {"seq": data["seq"][:, :-1], "tgt_key_padding_mask": ~data["mask"][:, :-1], "image": image,
                   "seq_layout_const": seqc["seq"], "seq_layout_const_pad_mask": seqc["pad_mask"]}
        return self._upload_batch(_inputs, {"seq": data["seq"][:, 1:]})

    def _encode_into_memory(self, inputs: dict) -> dict:
        self.rt.to(inputs["image"].device)
        return {"memory": self._constraint_memory(self._image_memory(inputs["image"]), inputs)}
