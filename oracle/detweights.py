"""ORACLE-side test infrastructure: deterministic, name-keyed weights.

Checkpoints of the reference (18 M transformer-side + 25 M backbone parameters) are far too
large to commit as fixtures, so every parity test regenerates weights from (key name, shape)
with a per-tensor CRC32-seeded CPU generator.  The same rule is used (a) in this container
to load the stub-imported reference before recording golden outputs
(tests/golden/make_golden.py) and (b) on the GPU box to fill the oracle and the HIP model.
torch's CPU generator is deterministic for a fixed torch build (same image on both sides).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import torch

from .ralf_oracle import RESNET50_STAGES, pe1d_table


def _gen(key: str) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed(zlib.crc32(key.encode()))
    return g


def det_tensor(key: str, shape: Tuple[int, ...]) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    g = _gen(key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "pe":  # sinusoidal table buffers [1, max_len, d]
        return pe1d_table(shape[1], shape[2])[None].clone()
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith("flag_img"):
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith("flag_user_const"):
        return torch.ones(shape, dtype=torch.long)
    if leaf == "token_mask":
        return torch.zeros(shape, dtype=torch.bool)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "running_var":
        return 1.0 + 0.2 * torch.rand(shape, generator=g)
    if leaf == "token":
        return torch.randn(shape, generator=g)
    if key == "task_emb.weight":
        return 0.2 * torch.randn(shape, generator=g)
    if "emb" in key and len(shape) == 2:  # embeddings
        return 0.05 * torch.randn(shape, generator=g)
    if len(shape) >= 2:  # linear / conv / packed in_proj
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        a = math.sqrt(3.0 / fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * a
    if leaf == "weight":  # 1-D weight = LayerNorm / BatchNorm gamma
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    return 0.05 * torch.randn(shape, generator=g)  # biases


def det_state_dict(shapes: Dict[str, Iterable[int]]) -> Dict[str, torch.Tensor]:
    return {k: det_tensor(k, tuple(v)) for k, v in shapes.items()}


def resnet50_fpn_shapes(prefix: str = "encoder.extractor", d_model: int = 256) -> Dict[str, Tuple[int, ...]]:
    """state_dict layout of ResnetBackbone (image2layout/train/models/common/image.py:27-88):
    torchvision-FX-extracted timm resnet50 body (keys = torchvision ResNet naming, `fc` dropped),
    4-channel conv1, FPN convs and proj."""
    b = prefix + ".body"
    out: Dict[str, Tuple[int, ...]] = {}

    def bn(p, c):
        out[p + ".weight"] = (c,)
        out[p + ".bias"] = (c,)
        out[p + ".running_mean"] = (c,)
        out[p + ".running_var"] = (c,)
        out[p + ".num_batches_tracked"] = ()

    out[b + ".conv1.weight"] = (64, 4, 7, 7)
    bn(b + ".bn1", 64)
    inpl = 64
    for li, (planes, blocks, stride) in enumerate(RESNET50_STAGES, start=1):
        for bi in range(blocks):
            q = f"{b}.layer{li}.{bi}"
            out[q + ".conv1.weight"] = (planes, inpl, 1, 1)
            bn(q + ".bn1", planes)
            out[q + ".conv2.weight"] = (planes, planes, 3, 3)
            bn(q + ".bn2", planes)
            out[q + ".conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(q + ".bn3", planes * 4)
            if bi == 0:
                out[q + ".downsample.0.weight"] = (planes * 4, inpl, 1, 1)
                bn(q + ".downsample.1", planes * 4)
            inpl = planes * 4
    for name, cin, k in (("fpn_conv11_4", 1024, 1), ("fpn_conv11_5", 2048, 1), ("fpn_conv33", 256, 3)):
        out[f"{prefix}.{name}.weight"] = (256, cin, k, k)
        out[f"{prefix}.{name}.bias"] = (256,)
    out[prefix + ".proj.weight"] = (d_model, 512, 1, 1)
    out[prefix + ".proj.bias"] = (d_model,)
    return out
