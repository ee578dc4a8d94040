"""Task conditions for constrained generation (host side).

get_condition mirrors image2layout/train/helpers/task.py:45-183; the containers mirror
image2layout/train/models/common/base_model.py:17-115 (same attribute names, `.to`, `.duplicate`).
The `relation` task adds the randomly thinned pairwise relations of helpers/relationships.compute_relation
(consumes Python's global `random` like the reference).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional

import torch

from .layout_tokenizer import GEO_KEYS

REFINEMENT_NOISE_STD = 0.01
EDGE_RATIO = 0.1          # fraction of relations kept by compute_relation (helpers/task.py:18)
COND_TYPES = ["c", "cwh", "partial", "gt", "refinement", "relation", None, "none", "uncond"]
VARS = {
    "c": ["label"],
    "cwh": ["label", "width", "height"],
    "relation": ["label"],
    "refinement": ["label", "width", "height", "center_x", "center_y"],
    "partial": ["label", "width", "height", "center_x", "center_y"],
}


@dataclass
class ConditionalInputsForDiscreteLayout:
    image: torch.Tensor
    id: Any = None
    task: Optional[str] = None
    seq: Optional[torch.Tensor] = None
    mask: Optional[torch.Tensor] = None
    seq_observed: Any = None
    weak_mask: Optional[torch.Tensor] = None
    weak_logits: Optional[torch.Tensor] = None
    edge_indexes: Optional[torch.Tensor] = None
    edge_attributes: Optional[torch.Tensor] = None

    def _tensor_fields(self):
        for name, value in vars(self).items():
            if torch.is_tensor(value):
                yield name, value

    def to(self, x):
        for name, value in list(self._tensor_fields()):
            setattr(self, name, value.to(x))
        return self

    def duplicate(self, n: int):
        for name, value in list(self._tensor_fields()):
            setattr(self, name, value.repeat_interleave(n, dim=0))
        return self


@dataclass
class RetrievalAugmentedConditionalInputsForDiscreteLayout(ConditionalInputsForDiscreteLayout):
    retrieved: dict = field(default_factory=dict)

    def __post_init__(self):
        r = self.retrieved
        if "image" in r and "saliency" in r and r["image"].size(2) < 4:
            r["image"] = torch.cat([r["image"], r["saliency"]], dim=2)

    def to(self, x):
        super().to(x)
        for k, v in self.retrieved.items():
            if torch.is_tensor(v):
                self.retrieved[k] = v.to(x)
        return self


# ---- conditions -------------------------------------------------------------------------------------------------
# A condition is the tokenised ground truth with the tokens the model must generate blanked out.  Every task below is a
# COLUMN rule on the [B, S] token grid (column 0 = BOS when the tokenizer has one, then C tokens per element in
# tokenizer.var_order), so the whole batch is handled with one boolean column vector -- no per-sample loop.

def _column_roles(tokenizer, S: int) -> torch.Tensor:
    """role of every sequence column: index into tokenizer.var_order, -1 for the BOS column"""
    lead = 1 if "bos" in tokenizer.special_tokens else 0
    roles = (torch.arange(S) - lead) % tokenizer.N_var_per_element
    roles[:lead] = -1
    return roles


def _given_columns(tokenizer, S: int, names) -> torch.Tensor:
    """bool [S]: BOS plus the columns carrying one of the attributes `names`"""
    roles = _column_roles(tokenizer, S)
    wanted = torch.tensor([tokenizer.var_order.index(n) for n in names])
    return (roles == -1) | torch.isin(roles, wanted)


def _blank_id(tokenizer) -> int:
    return tokenizer.name_to_id("mask") if "mask" in tokenizer.special_tokens else -1


def _cond_unconditional(enc, batch, tokenizer):
    return {"seq": None, "mask": None}


def _cond_ground_truth(enc, batch, tokenizer):
    return enc


def _cond_first_element(enc, batch, tokenizer):
    """`partial`: only the first element is known.  Its tokens already sit at the head of the sequence (right behind BOS), so
    "move the known tokens to the front" is the identity on the first `lead + C` columns, for every row alike."""
    S = enc["seq"].shape[1]
    lead = 1 if "bos" in tokenizer.special_tokens else 0
    known = torch.arange(S) < lead + tokenizer.N_var_per_element
    seq = torch.where(known, enc["seq"], torch.full_like(enc["seq"], _blank_id(tokenizer)))
    return {"seq": seq, "mask": known.expand_as(enc["mask"]).clone()}


def _cond_attributes(names, with_relations=False):
    def build(enc, batch, tokenizer):
        out = dict(enc)
        if with_relations:   # consumes Python's `random` exactly once per call, before anything else (like the reference)
            from .relationships import compute_relation
            out.update(compute_relation(batch, edge_ratio=EDGE_RATIO))
        given = _given_columns(tokenizer, enc["seq"].shape[1], names)
        real = enc["mask"]                                   # tokens of existing elements (and BOS / EOS)
        seq = torch.where(given, enc["seq"], torch.full_like(enc["seq"], _blank_id(tokenizer)))
        out["seq"] = torch.where(real, seq, torch.full_like(seq, tokenizer.name_to_id("pad")))   # the element count is known
        out["mask"] = (real & given) | ~real
        return out
    return build


def _cond_noisy_geometry(enc, batch, tokenizer):
    """`refinement`: the geometry is observed through N(0, 0.01) noise (one torch.normal per geometry key, in GEO_KEYS order:
    the reference's random stream); the perturbed layout also REPLACES the batch's geometry, as in the reference"""
    observed = {"label": batch["label"], "mask": batch["mask"]}
    present = batch["mask"]
    for key in GEO_KEYS:
        jitter = torch.normal(0, REFINEMENT_NOISE_STD, size=batch[key].size())
        observed[key] = torch.clamp(batch[key] + jitter, min=0.0, max=1.0) * present
        batch[key] = observed[key].clone()
    return {"seq": tokenizer.encode(observed)["seq"], "mask": enc["mask"], "seq_observed": observed}


_CONDITION_BUILDERS = {
    None: _cond_unconditional, "none": _cond_unconditional, "uncond": _cond_unconditional,
    "gt": _cond_ground_truth,
    "partial": _cond_first_element,
    "c": _cond_attributes(VARS["c"]),
    "cwh": _cond_attributes(VARS["cwh"]),
    "relation": _cond_attributes(VARS["relation"], with_relations=True),
    "refinement": _cond_noisy_geometry,
}


def _sample_ids(batch):
    ids = batch.get("id")
    try:
        return torch.tensor([int(i) for i in ids], dtype=torch.long)
    except (TypeError, ValueError):
        return ids


_PINNED: dict = {}
_PINNED_MAX = 6


def _storage_unshared(t: torch.Tensor) -> bool:
    """no tensor but `t` refers to t's storage (torch's own reference count of the storage: `t` and the temporary handle made here)"""
    try:
        return torch._C._storage_Use_Count(t.untyped_storage()._cdata) <= 2
    except AttributeError:   # a torch without that hook: never reuse (every call gets a buffer of its own, up to the cap, then torch.cat)
        return False


def pinned_copy_issued(src: torch.Tensor, event) -> None:
    """an asynchronous host-to-device copy that READS a cat_image staging buffer has been issued (models/ralf.py: _upload_batch): the buffer
    is not rewritten before `event`"""
    for ring in _PINNED.values():
        for slot in ring["bufs"]:
            if slot[0].data_ptr() == src.data_ptr():
                slot[1] = event


def cat_image(image: torch.Tensor, saliency: torch.Tensor) -> torch.Tensor:
    """torch.cat([image, saliency], dim=1) (helpers/task.py:78-80 of the reference).  On a GPU box the 4-channel batch (67 MB at B = 64,
    256 x 256) is assembled in a page-locked staging buffer by two multi-threaded slice copies: the loop's `.to(rank)` (train/train.py:434)
    is then one DMA at PCIe rate instead of a pageable copy, and the single-threaded cat (10-13 ms of a 15 ms step) is gone.
    The buffers rotate, but a buffer is only ever REUSED when nothing refers to it any more: a caller that keeps the returned tensor (or a
    view of it: a collected condition, a prefetched batch) keeps its buffer, and an asynchronous copy that reads it is waited for
    (pinned_copy_issued).  With all buffers (at most six per shape) in use the plain torch.cat semantics apply."""
    if not (image.device.type == "cpu" and saliency.device.type == "cpu" and image.numel() >= (1 << 20) and torch.cuda.is_available()):
        return torch.cat([image, saliency], dim=1)
    B, C, H, W = image.shape
    key = (B, C + saliency.size(1), H, W, image.dtype)
    ring = _PINNED.setdefault(key, {"bufs": []})
    slot = None
    for cand in ring["bufs"]:   # [pinned storage owner, event of the last asynchronous reader]
        if _storage_unshared(cand[0]):   # only the ring itself refers to the storage: the tensor handed out last time (and every view of it) is gone
            slot = cand
            break
    if slot is None:
        if len(ring["bufs"]) >= _PINNED_MAX:
            return torch.cat([image, saliency], dim=1)
        slot = [torch.empty(key[:4], dtype=image.dtype, pin_memory=True), None]
        ring["bufs"].append(slot)
    if slot[1] is not None:
        slot[1].synchronize()
        slot[1] = None
    buf = slot[0].view(key[:4])   # (a VIEW is handed out: its lifetime is what the use count above observes)
    buf[:, :C].copy_(image)
    buf[:, C:].copy_(saliency)
    return buf


def get_condition(batch: dict, cond_type: Optional[str] = None, tokenizer=None):
    """(condition container, batch) for `cond_type` -- same contract as image2layout/train/helpers/task.py:45-183: `seq` holds
    the known tokens with the blank id (-1 / [MASK]) where the model must generate, `mask` is True on known and special
    tokens.  Without a tokenizer the batch passes through untouched (GAN-style generators)."""
    assert cond_type in COND_TYPES
    if tokenizer is None:
        return batch, batch
    image = batch["image"] if batch["image"].size(1) == 4 else cat_image(batch["image"], batch["saliency"])
    fields = _CONDITION_BUILDERS[cond_type](tokenizer.encode(batch), batch, tokenizer)
    fields = dict(fields, id=_sample_ids(batch))
    container = ConditionalInputsForDiscreteLayout
    if "retrieved" in batch:
        if isinstance(batch["retrieved"], list):   # collate_fn wraps the exemplars in a one-element list
            (batch["retrieved"],) = batch["retrieved"]
        fields["retrieved"] = batch["retrieved"]
        container = RetrievalAugmentedConditionalInputsForDiscreteLayout
    return container(image=image, task=cond_type, **fields), batch
