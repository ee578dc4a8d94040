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


def get_condition(batch: dict, cond_type: Optional[str] = None, tokenizer=None):
    assert cond_type in COND_TYPES
    if tokenizer is None:
        return batch, batch
    image = batch["image"] if batch["image"].size(1) == 4 else torch.cat([batch["image"], batch["saliency"]], dim=1)
    specials = tokenizer.special_tokens
    pad_id = tokenizer.name_to_id("pad")
    mask_id = tokenizer.name_to_id("mask") if "mask" in specials else -1
    cond = tokenizer.encode(batch)
    B, S = cond["seq"].shape
    C = tokenizer.N_var_per_element
    has_bos = "bos" in specials

    if cond_type in (None, "none", "uncond"):
        cond = {"seq": None, "mask": None}
    elif cond_type == "partial":
        keep = torch.zeros_like(batch["mask"])
        keep[:, 0] = True  # only the first element is given
        keep = keep.unsqueeze(-1).expand(-1, -1, C).reshape(B, -1)
        if has_bos:
            keep = torch.cat([torch.ones(B, 1, dtype=torch.bool), keep], dim=-1)
            new_seq = torch.full_like(cond["seq"], mask_id)
            new_mask = torch.zeros_like(cond["mask"])
            for i in range(B):
                n = int(keep[i].sum())
                new_seq[i, :n] = cond["seq"][i][keep[i]]
                new_mask[i, :n] = True
            cond["seq"], cond["mask"] = new_seq, new_mask
        else:
            cond["seq"][~keep] = mask_id
            cond["mask"] = keep
    elif cond_type in ("c", "cwh", "relation"):
        if cond_type == "relation":
            from .relationships import compute_relation
            cond.update(compute_relation(batch, edge_ratio=EDGE_RATIO))
        if has_bos:
            attr = (torch.arange(S).view(1, S) - 1) % C
            attr[:, 0] = -1
        else:
            attr = torch.arange(S).view(1, S) % C
        keep = torch.zeros(B, S, dtype=torch.bool)
        if has_bos:
            keep[:, 0] = True
        for name in VARS[cond_type]:
            keep |= attr == tokenizer.var_order.index(name)
        cond["seq"][~keep] = mask_id
        cond["seq"][~cond["mask"]] = pad_id
        cond["mask"] = (cond["mask"] & keep) | ~cond["mask"]
    elif cond_type == "gt":
        pass
    elif cond_type == "refinement":
        noisy = {"label": batch["label"], "mask": batch["mask"]}
        for key in GEO_KEYS:
            noise = torch.normal(0, REFINEMENT_NOISE_STD, size=batch[key].size())
            noisy[key] = torch.clamp(batch[key] + noise, min=0.0, max=1.0)
            noisy[key][~batch["mask"]] = 0.0
            batch[key] = noisy[key].clone()
        cond = {"seq": tokenizer.encode(noisy)["seq"], "mask": cond["mask"], "seq_observed": noisy}
    else:
        raise NotImplementedError(cond_type)

    try:
        cond["id"] = torch.tensor([int(i) for i in batch["id"]], dtype=torch.long)
    except Exception:
        cond["id"] = batch.get("id")

    if "retrieved" in batch:
        if isinstance(batch["retrieved"], list):
            assert len(batch["retrieved"]) == 1
            batch["retrieved"] = batch["retrieved"][0]
        cond["retrieved"] = batch["retrieved"]
        cls = RetrievalAugmentedConditionalInputsForDiscreteLayout
    else:
        cls = ConditionalInputsForDiscreteLayout
    return cls(image=image, task=cond_type, **cond), batch
