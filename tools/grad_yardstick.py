"""A YARDSTICK under the bf16 backbone gradients.

The HIP path's bf16 weight gradients of the ResNet body agree with its fp32 ones to cosine 0.1-0.8 only
(profiles/r05_grad_agreement_trained.json).  Is that the precision's property or a rounding point of THIS implementation?
The checker here is stock PyTorch: the oracle model (oracle/ralf_oracle.py -- torch ops, test infrastructure) on the GPU in fp32 and
under torch.autocast(bfloat16), same weights and batch as tools/grad_agreement_trained.py, same 13 tensors.

  hip_bf16_vs_hip_fp32        the question
  autocast_vs_torch_fp32      the yardstick: what a stock bf16 mixed-precision run of the same model gives
  hip_fp32_vs_torch_fp32      sanity: two fp32 implementations (must be ~1)
  hip_bf16_vs_autocast        two bf16 implementations against each other

    python tools/grad_yardstick.py [steps=400] [out.json]
(on the GPU box; torch's own convolutions / GEMMs (MIOpen, rocBLAS) are used by the CHECKER only)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import grad_agreement_trained as ga  # noqa: E402
from oracle import ralf_oracle as O  # noqa: E402


def cos(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


def hip_grads(sd, batch, dev, dtype, N=10):
    m = bench.build_model(dev, N, dtype)
    m.load_state_dict(sd)
    m.rt.drop_p = lambda p: 0.0          # train mode (BatchNorm on batch statistics) without dropout
    o, l = m.train_loss(*batch)
    l["nll_loss"].backward()
    pad = m.tokenizer.name_to_id("pad")
    out = (l["nll_loss"].item(), {k: p.grad.float().clone() for k, p in m.named_parameters() if k in ga.KEYS}, pad)
    del m
    return out


def torch_grads(sd, batch, pad, autocast: bool):
    """the oracle on the GPU.  Under autocast its hand-written LayerNorm (mean / sqrt in the input's dtype) is replaced by F.layer_norm, which
    autocast runs in fp32 -- what a stock mixed-precision model does."""
    inputs, targets = batch
    sdo = {k: v.detach().clone().float() for k, v in sd.items()}
    for k in ga.KEYS:
        sdo[k].requires_grad_(True)
    ln = O.layer_norm
    if autocast:
        O.layer_norm = lambda x, s, p, eps=1e-5: torch.nn.functional.layer_norm(x, x.shape[-1:], s[p + ".weight"], s[p + ".bias"], eps)
    try:
        with torch.device(inputs["seq"].device), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            logits = O.ralf_forward(sdo, inputs, training_bn=True, p_drop=0.0)
            loss = O.xent_label_smoothing(logits.float(), targets["seq"], pad)
        grads = torch.autograd.grad(loss, [sdo[k] for k in ga.KEYS])
    finally:
        O.layer_norm = ln
    return loss.item(), {k: g.float() for k, g in zip(ga.KEYS, grads)}


def table(a, b):
    return {k: (round(cos(a[k], b[k]), 4), round((a[k].norm() / b[k].norm()).item(), 4)) for k in ga.KEYS}


def yardstick(steps, dev):
    sd, batch, curve = ga.trained_state(steps, dev)
    l32, g32, pad = hip_grads(sd, batch, dev, "float32")
    l16, g16, _ = hip_grads(sd, batch, dev, "bfloat16")
    t32, h32 = torch_grads(sd, batch, pad, False)
    t16, h16 = torch_grads(sd, batch, pad, True)
    return {"steps": steps, "loss_curve": curve,
            "loss": {"hip_fp32": l32, "hip_bf16": l16, "torch_fp32": t32, "torch_autocast_bf16": t16},
            "hip_bf16_vs_hip_fp32": table(g16, g32), "autocast_vs_torch_fp32": table(h16, h32),
            "hip_fp32_vs_torch_fp32": table(g32, h32), "hip_bf16_vs_autocast": table(g16, h16),
            "hip_bf16_vs_torch_fp32": table(g16, h32)}


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    dev = torch.device("cuda", 0)
    res = {}
    for s in sorted({0, steps}):
        res[f"after_{s}_steps"] = yardstick(s, dev)
        print(s, json.dumps(res[f"after_{s}_steps"]), flush=True)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(res, f, indent=1)
