"""Layout <-> token-id conversion (host integer path, must be bit-exact with the reference).

Mirrors the interface of image2layout/train/helpers/layout_tokenizer.py:288-446
(LayoutSequenceTokenizer.encode / decode / token_mask, name_to_id, N_total, ...) and the linear
bucketizer of image2layout/train/helpers/bucketizer.py:44-81.  Vocabulary layout:
  [0, N_label) labels | N_label + i*num_bin + b for geometry variable i in GEO_KEYS order
  (center_x, center_y, width, height; one shared block when is_loc_vocab_shared) | special tokens.
Only geo_quantization="linear" is implemented (kmeans needs the authors' pickled cluster centres).
"""
from __future__ import annotations

from typing import Sequence

import torch

GEO_KEYS = ["center_x", "center_y", "width", "height"]
SPECIAL = ["pad", "bos", "eos", "mask"]


class LabelFeature:
    """Stand-in for datasets.ClassLabel (only .names / .num_classes are used)."""

    def __init__(self, names: Sequence[str]):
        self.names = list(names)
        self.num_classes = len(self.names)


class LinearBucketizer:
    def __init__(self, n: int):
        edges = torch.arange(n + 1) / n
        self.boundaries = edges[1:]
        self.centers = ((edges[:-1] + edges[1:]) / 2.0).unsqueeze(1)

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        # == torch.bucketize(x, boundaries) (same kernel, same right=False rule); bucketize itself spends milliseconds per call in
        # this torch build's dispatch on tiny tensors, searchsorted microseconds: 8 calls per batch were 10-40 ms of host time
        return torch.searchsorted(self.boundaries, torch.clamp(x, min=0.0, max=1.0).contiguous())

    def decode(self, idx: torch.Tensor) -> torch.Tensor:
        idx = torch.clamp(idx, min=0, max=len(self.centers) - 1)
        return self.centers[idx][..., 0]


class LayoutSequenceTokenizer:
    def __init__(self, label_feature, max_seq_length: int, num_bin: int = 128,
                 var_order=("label", "width", "height", "center_x", "center_y"), pad_until_max: bool = False,
                 special_tokens=("pad", "bos", "eos"), is_loc_vocab_shared: bool = False, geo_quantization: str = "linear", **_):
        if geo_quantization != "linear":
            raise NotImplementedError("only geo_quantization='linear' is available (kmeans needs the authors' cluster pickle)")
        if not hasattr(label_feature, "num_classes"):
            label_feature = LabelFeature(label_feature)
        self._label_feature = label_feature
        self._max_seq_length = int(max_seq_length)
        self._num_bin = int(num_bin)
        self._var_order = list(var_order)
        self._pad_until_max = bool(pad_until_max)
        self._special_tokens = list(special_tokens)
        self._is_loc_vocab_shared = bool(is_loc_vocab_shared)
        self._geo_quantization = geo_quantization
        assert "pad" in self._special_tokens and all(t in SPECIAL for t in self._special_tokens)
        assert sorted(self._var_order) == sorted(["label"] + GEO_KEYS)
        self._bucketizers = {k: LinearBucketizer(self._num_bin) for k in GEO_KEYS}
        base = self.N_label + self.N_bbox
        self._special_token_name_to_id = {t: base + i for i, t in enumerate(self._special_tokens)}
        self._special_token_id_to_name = {v: k for k, v in self._special_token_name_to_id.items()}

    # ---- vocabulary geometry -----------------------------------------------------------------
    N_label = property(lambda s: int(s._label_feature.num_classes))
    N_bbox_per_var = property(lambda s: s._num_bin)
    N_bbox = property(lambda s: s._num_bin if s._is_loc_vocab_shared else 4 * s._num_bin)
    N_sp_token = property(lambda s: len(s._special_tokens))
    N_total = property(lambda s: s.N_label + s.N_bbox + s.N_sp_token)
    N_var_per_element = property(lambda s: len(s._var_order))
    max_seq_length = property(lambda s: s._max_seq_length)
    max_token_length = property(lambda s: s._max_seq_length * len(s._var_order))
    var_order = property(lambda s: s._var_order)
    special_tokens = property(lambda s: s._special_tokens)
    is_loc_vocab_shared = property(lambda s: s._is_loc_vocab_shared)
    geo_quantization = property(lambda s: s._geo_quantization)
    pad_until_max = property(lambda s: s._pad_until_max)
    bucketizers = property(lambda s: s._bucketizers)

    def name_to_id(self, name: str) -> int:
        return self._special_token_name_to_id[name]

    def id_to_name(self, i: int) -> str:
        return self._special_token_id_to_name[i]

    def _geo_offset(self, key: str) -> int:
        return self.N_label + (0 if self._is_loc_vocab_shared else GEO_KEYS.index(key) * self._num_bin)

    # ---- encode ------------------------------------------------------------------------------
    def encode(self, inputs: dict) -> dict:
        """label int64 [B,S], geometry float [B,S], mask bool [B,S] -> seq int64 / mask bool [B, 5S+1]."""
        mask = inputs["mask"].clone()
        cols = {"label": inputs["label"].clone()}
        for key in GEO_KEYS:
            cols[key] = self._bucketizers[key].encode(inputs[key]) + self._geo_offset(key)
        if self._pad_until_max and mask.shape[-1] < self._max_seq_length:
            extra = self._max_seq_length - mask.shape[-1]
            mask = torch.cat([mask, torch.zeros(*mask.shape[:-1], extra, dtype=torch.bool)], dim=-1)
            for k in cols:
                cols[k] = torch.cat([cols[k], torch.zeros(*cols[k].shape[:-1], extra, dtype=cols[k].dtype)], dim=-1)
        pad_id = self.name_to_id("pad")
        for k in cols:
            cols[k][~mask] = pad_id
        B, S = cols["label"].shape
        C = self.N_var_per_element
        n_elem = mask.int().sum(dim=1, keepdim=True)
        assert torch.equal(~mask, n_elem <= torch.arange(S).unsqueeze(0)), "mask must be a prefix mask"
        seq = torch.stack([cols[k] for k in self._var_order], dim=-1).reshape(B, S * C)
        tmask = mask.unsqueeze(-1).expand(B, S, C).reshape(B, S * C).clone()
        if "bos" in self._special_tokens and "eos" in self._special_tokens:
            at_end = (n_elem * C) == torch.arange(S * C).unsqueeze(0)
            seq[at_end] = self.name_to_id("eos")
            tmask[at_end] = True
            seq = torch.cat([torch.full((B, 1), self.name_to_id("bos")), seq], dim=-1)
            tmask = torch.cat([torch.ones(B, 1, dtype=torch.bool), tmask], dim=-1)
        return {"seq": seq, "mask": tmask}

    # ---- decode ------------------------------------------------------------------------------
    def decode(self, seq: torch.Tensor) -> dict:
        """seq int64 [B, 5S] (no BOS) -> label / geometry (bin centres) / mask [B, S]."""
        C = self.N_var_per_element
        t = seq.clone().reshape(seq.shape[0], -1, C)
        out = {}
        for i, key in enumerate(self._var_order):
            out[key] = t[..., i] - (self._geo_offset(key) if key in GEO_KEYS else 0)
        if "bos" in self._special_tokens and "eos" in self._special_tokens:
            invalid = torch.cumsum(out["label"] == self.name_to_id("eos"), dim=1) > 0
        else:
            invalid = torch.zeros_like(out["label"], dtype=torch.bool)
        ok = (out["label"] >= 0) & (out["label"] < self.N_label)
        for key in GEO_KEYS:
            ok &= (out[key] >= 0) & (out[key] < self.N_bbox)
        invalid = invalid | ~ok
        for key in GEO_KEYS:
            idx = out[key].clone()
            idx[invalid] = 0
            val = self._bucketizers[key].decode(idx)
            val[invalid] = 0.0
            out[key] = val
        out["label"] = out["label"].clone()
        out["label"][invalid] = 0
        out["mask"] = ~invalid
        return out

    # ---- per-position vocabulary mask -----------------------------------------------------------
    @property
    def token_mask(self) -> torch.Tensor:
        """bool [5S, V]: True where a token may be predicted at that position (no bos/mask)."""
        if self._is_loc_vocab_shared:
            # the reference's own property fails for a shared vocabulary (mismatched stack sizes)
            raise NotImplementedError("token_mask is undefined for is_loc_vocab_shared=True in the reference")
        V, nb, nl = self.N_total, self._num_bin, self.N_label
        tail = torch.tensor([t not in ("bos", "mask") for t in self._special_tokens])
        rows = []
        for key in self._var_order:
            m = torch.zeros(V, dtype=torch.bool)
            if key == "label":
                m[:nl] = True
            else:
                o = self._geo_offset(key)
                m[o:o + nb] = True
            m[nl + self.N_bbox:] = tail
            rows.append(m)
        return torch.stack(rows, dim=0).repeat(self._max_seq_length, 1)
