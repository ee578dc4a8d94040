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


def make_learnable_set(S: int, B: int, N: int = 10, K: int = 16, num_labels: int = 3, H: int = 128, W: int = 128, seed: int = 7) -> list:
    """a FIXED set of S samples in batches of B whose layouts can be learned from the inputs (convergence runs): the saliency channel
    shows every element's box, the red channel its label, and the K exemplars are jittered copies of the target layout -- so the loss
    falls through the image branch AND the retrieval branch, not only by memorising token statistics.  Same dict layout as make_batch."""
    g = torch.Generator().manual_seed(seed)
    out = []
    ys = (torch.arange(H).float() + 0.5) / H
    xs = (torch.arange(W).float() + 0.5) / W
    for _ in range(S // B):
        n = torch.randint(1, N + 1, (B,), generator=g)
        mask = torch.arange(N).expand(B, N) < n.unsqueeze(-1)
        lab = torch.randint(0, num_labels, (B, N), generator=g)
        lab = torch.sort(lab.masked_fill(~mask, num_labels), dim=-1).values.masked_fill(~mask, 0)
        w = (0.15 + 0.35 * torch.rand(B, N, generator=g)) * mask
        h = (0.10 + 0.30 * torch.rand(B, N, generator=g)) * mask
        cx = (w / 2 + (1 - w) * torch.rand(B, N, generator=g)) * mask
        cy = (h / 2 + (1 - h) * torch.rand(B, N, generator=g)) * mask
        inside = ((xs.view(1, 1, 1, W) - cx.view(B, N, 1, 1)).abs() <= w.view(B, N, 1, 1) / 2) & \
                 ((ys.view(1, 1, H, 1) - cy.view(B, N, 1, 1)).abs() <= h.view(B, N, 1, 1) / 2) & mask.view(B, N, 1, 1)
        sal = inside.any(dim=1, keepdim=True).float()
        img = 0.25 * torch.rand(B, 3, H, W, generator=g)
        img[:, 0] += 0.75 * (inside.float() * ((lab.float() + 1) / num_labels).view(B, N, 1, 1)).amax(dim=1)
        b = {"mask": mask, "label": lab, "center_x": cx, "center_y": cy, "width": w, "height": h, "image": img, "saliency": sal,
             "id": [str(i) for i in range(B)]}
        jit = lambda t: ((t.unsqueeze(1) + 0.03 * torch.randn(B, K, N, generator=g)).clamp(0, 1)) * mask.unsqueeze(1)  # noqa: E731
        r = {"mask": mask.unsqueeze(1).expand(B, K, N).clone(), "label": lab.unsqueeze(1).expand(B, K, N).clone(),
             "center_x": jit(cx), "center_y": jit(cy), "width": jit(w), "height": jit(h), "image": torch.zeros(B, K, 4, 1, 1)}
        b["retrieved"] = [r]
        out.append(b)
    return out
