"""bf16 throughput mode against fp32 parity mode, GRADIENTS of the whole model (ResNet-50 / FPN in front) at B = 64, 256 x 256 on TRAINED weights --
no damping of the residual branches: the weights after `steps` optimisation steps (bf16, learnable synthetic set, the benchmark's recipe) are loaded
into an fp32 and a bf16 model; per-tensor cosine and norm ratio of their gradients on a fresh batch.
    python tools/grad_agreement_trained.py [steps=1000] [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ralf_amd.engine import TrainStep  # noqa: E402
from ralf_amd.synthetic import make_learnable_set, to_device  # noqa: E402

KEYS = ["decoder.head.1.weight", "decoder.transformer.layers.0.multihead_attn.in_proj_weight", "transformer_encoder.layers.0.linear1.weight",
        "transformer_encoder.layers.5.self_attn.in_proj_weight", "head.net.1.weight", "attn.to_kv.weight",
        "encoder.extractor.proj.weight", "encoder.extractor.fpn_conv33.weight", "encoder.extractor.body.layer4.2.conv2.weight",
        "encoder.extractor.body.layer3.0.conv1.weight", "encoder.extractor.body.layer2.0.conv2.weight", "encoder.extractor.body.layer1.0.conv1.weight",
        "encoder.extractor.body.conv1.weight"]


def cos(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


def trained_state(steps, dev, B=64, N=10, HW=256):
    model = bench.build_model(dev, N, "bfloat16")
    batches = make_learnable_set(1024, B, N, H=HW, W=HW)
    devb = []
    for b in batches:
        i, t = model.preprocess(b)
        i, t = to_device(i, dev), to_device(t, dev)
        i["retrieved"] = {k: v for k, v in i["retrieved"].items() if k != "image"}
        devb.append((i, t))
    losses = []
    if steps:
        step = TrainStep(model, lr=1e-4, weight_decay=1e-4, max_norm=0.1, use_graph=True)
        losses = [step(*devb[k % len(devb)]).clone() for k in range(steps)]
        torch.cuda.synchronize()
        del step
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return sd, devb[-1], [float(x) for x in losses[:: max(1, steps // 10)]]


def gradient_table(sd, batch, dev, N=10):
    out = {}
    for name in ("float32", "bfloat16"):
        m = bench.build_model(dev, N, name)
        m.load_state_dict(sd)
        m.rt.drop_p = lambda p: 0.0          # train mode (BatchNorm on batch statistics) without dropout
        o, l = m.train_loss(*batch)
        l["nll_loss"].backward()
        out[name] = (o["logits"].float(), l["nll_loss"].item(), {k: p.grad.float() for k, p in m.named_parameters() if p.grad is not None})
        del m
    (lg32, l32, g32), (lg16, l16, g16) = out["float32"], out["bfloat16"]
    table = {k: (round(cos(g32[k], g16[k]), 4), round((g16[k].norm() / g32[k].norm()).item(), 4)) for k in KEYS}
    return {"loss_f32": l32, "loss_bf16": l16, "logits_cosine": cos(lg32, lg16), "grad_cos_normratio": table}


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    dev = torch.device("cuda", 0)
    res = {}
    for s in sorted({0, steps}):
        sd, batch, curve = trained_state(s, dev)
        res[f"after_{s}_steps"] = dict(gradient_table(sd, batch, dev), loss_curve=curve)
        print(s, json.dumps(res[f"after_{s}_steps"]))
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)
