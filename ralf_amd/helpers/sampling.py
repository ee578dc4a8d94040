"""Token sampling and decode-space restriction (host side of the autoregressive loop).

sample(): image2layout/train/helpers/sampling.py:18-71 (deterministic / top_k / top_p / random / gumbel).
restrict_*: image2layout/train/models/layoutformerpp/decoding_space_restriction.py:5-106, vectorised
over the batch (the reference loops per sample with .item() syncs; results are identical).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

NEG_INF = -float("inf")


def _get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def top_k_logits(logits: torch.Tensor, k: int) -> torch.Tensor:
    v, _ = torch.topk(logits, k, dim=1)
    out = logits.clone()
    out[out < v[:, [-1]]] = NEG_INF
    return out


def sample(logits: torch.Tensor, sampling_cfg, temperature=None) -> torch.Tensor:
    """logits [B, C] -> token ids [B, 1]."""
    assert logits.ndim == 2
    name = _get(sampling_cfg, "name")
    if name == "deterministic":
        return torch.argmax(logits, dim=1, keepdim=True)
    if temperature is None:
        temperature = _get(sampling_cfg, "temperature", 1.0)
    scaled = logits / temperature
    if name == "top_k":
        scaled = top_k_logits(scaled, _get(sampling_cfg, "top_k"))
    elif name == "top_p":
        top_p = _get(sampling_cfg, "top_p")
        assert 0.0 < top_p <= 1.0
        srt, order = torch.sort(scaled, descending=True, dim=1)
        cum = torch.cumsum(F.softmax(srt, dim=1), dim=1)
        pos = torch.arange(scaled.size(1), device=scaled.device).view(1, -1)
        srt[(cum > top_p) & (pos > 0)] = NEG_INF
        scaled = srt.gather(dim=1, index=order.argsort(dim=1))
    elif name == "gumbel":
        u = torch.rand_like(scaled)
        scaled = scaled - torch.log(-torch.log(u + 1e-30) + 1e-30)
    elif name != "random":
        raise NotImplementedError(name)
    return torch.multinomial(F.softmax(scaled, dim=1), num_samples=1)


def _first_pad(cond: torch.Tensor, pad_id: int) -> torch.Tensor:
    """index of the first pad token per row (= position of the end), +inf-like when absent."""
    is_pad = cond == pad_id
    first = torch.argmax(is_pad.float(), dim=1)
    return torch.where(is_pad.any(dim=1), first, torch.full_like(first, cond.shape[1] + 1))


def restrict_reliable_label_or_size(step: int, cond, logits, pad_id, eos_id, max_length):
    """c / cwh: force the given token before the end, force <eos> at/after it; free where cond is pad or -1."""
    assert cond.size(1) == max_length + 1
    given = cond[:, step]
    first_pad = _first_pad(cond, pad_id)
    before = step < first_pad
    free = before & ((given == pad_id) | (given == -1))
    forced = torch.where(before, given, torch.full_like(given, eos_id))
    keep = torch.zeros_like(logits, dtype=torch.bool)
    rows = ~free
    keep[rows, forced[rows].clamp(min=0)] = True
    keep[free] = True
    logits[~keep] = NEG_INF
    return logits


def restrict_only_category(step: int, cond, logits, pad_id, eos_id, max_length):
    """refinement / relation: only the label slots ((step-1) % 5 == 0) are constrained."""
    if (step - 1) % 5 != 0:
        return logits
    return restrict_reliable_label_or_size(step, cond, logits, pad_id, eos_id, max_length)


def identity(step, cond, logits, pad_id, eos_id, max_length):
    return logits


DECODE_SPACE_RESTRICTION = {
    "none": identity, "uncond": identity, "partial": identity,
    "c": restrict_reliable_label_or_size, "cwh": restrict_reliable_label_or_size,
    "refinement": restrict_only_category, "relation": restrict_only_category,
}


def forced_tokens_all(cond: torch.Tensor, cond_type: str, pad_id: int, eos_id: int, max_length: int):
    """forced_tokens for every decode step at once: int64 [max_length, B] (row i = step i + 1), or None when the task does
    not constrain tokens.  The decode loop indexes one row per step instead of re-deriving it with ~10 tiny launches."""
    if cond_type in ("none", "uncond", "partial", None) or cond is None:
        return None
    assert cond.size(1) == max_length + 1
    steps = torch.arange(1, max_length + 1, device=cond.device).view(1, -1)        # [1, T]
    given = cond[:, 1:]                                                              # [B, T]
    before = steps < _first_pad(cond, pad_id).view(-1, 1)
    free = before & ((given == pad_id) | (given == -1))
    forced = torch.where(before, given, torch.full_like(given, eos_id))
    forced = torch.where(free, torch.full_like(given, -1), forced)
    if cond_type in ("refinement", "relation"):                                      # only the label slots are constrained
        forced = torch.where((steps - 1) % 5 == 0, forced, torch.full_like(forced, -1))
    return forced.t().contiguous()


def forced_tokens(step: int, cond: torch.Tensor, cond_type: str, pad_id: int, eos_id: int, max_length: int):
    """DECODE_SPACE_RESTRICTION expressed as one forced token per sample (-1 = unconstrained) for the fused
    on-device mask+sample kernel; equivalent to restrict_* above (which mask every other logit)."""
    if cond_type in ("none", "uncond", "partial", None) or cond is None:
        return None
    if cond_type in ("refinement", "relation") and (step - 1) % 5 != 0:
        return None
    assert cond.size(1) == max_length + 1
    given = cond[:, step]
    before = step < _first_pad(cond, pad_id)
    free = before & ((given == pad_id) | (given == -1))
    forced = torch.where(before, given, torch.full_like(given, eos_id))
    return torch.where(free, torch.full_like(given, -1), forced).contiguous()
