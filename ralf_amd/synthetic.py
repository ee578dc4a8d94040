"""Synthetic PKU/CGL-shaped batches: U[0,1) canvases, n ~ U{1..N} elements with
uniform labels / geometry, K retrieved exemplar layouts per sample.  Same dict layout as the
reference's collate_fn (image2layout/train/data.py:74-117) minus the unused retrieved pixels
(a 1x1 placeholder keeps the 4-channel assertion of preprocess)."""
from __future__ import annotations

import torch


def _layouts(g, lead, N, C):
    shape = tuple(lead) + (N,)
    n = torch.randint(1, N + 1, tuple(lead), generator=g)
    mask = torch.arange(N).expand(shape) < n.unsqueeze(-1)
    lab = torch.randint(0, C, shape, generator=g)
    lab = torch.sort(lab.masked_fill(~mask, C), dim=-1).values.masked_fill(~mask, 0)  # label-sorted, as ralf.yaml sorts
    out = {"mask": mask, "label": lab}
    for k in ("center_x", "center_y", "width", "height"):
        out[k] = torch.rand(shape, generator=g) * mask
    return out


def make_batch(B: int, N: int = 10, K: int = 16, num_labels: int = 3, H: int = 256, W: int = 256, seed: int = 1, with_retrieval: bool = True) -> dict:
    g = torch.Generator().manual_seed(seed)
    b = _layouts(g, (B,), N, num_labels)
    b["image"] = torch.rand(B, 3, H, W, generator=g)
    b["saliency"] = torch.rand(B, 1, H, W, generator=g)
    b["id"] = [str(i) for i in range(B)]
    if with_retrieval:
        r = _layouts(g, (B, K), N, num_labels)
        r["image"] = torch.zeros(B, K, 4, 1, 1)  # retrieved pixels are unused when use_reference_image=False
        b["retrieved"] = [r]
    return b


def to_device(x, device):
    if torch.is_tensor(x):
        return x.to(device, non_blocking=True)
    if isinstance(x, dict):
        return {k: to_device(v, device) for k, v in x.items()}
    return x
