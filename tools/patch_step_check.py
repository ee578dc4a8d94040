"""captured train steps (hipGraph, side stream) with lr = 0 on fixed weights and a fixed batch: the gradients of the LAST replay, saved (argv[1]) or compared
bit for bit with a saved run -- RALF_GEMM_PATCH=0 / 1 must agree wherever two runs of one setting do (the weight-matrix gradients of the backbone do)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ralf_amd.engine import TrainStep  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402

dev = torch.device("cuda", 0)
m = bench.build_model(dev, 10, "bfloat16")
inputs, targets = m.preprocess(make_batch(64, 10, seed=1))
inputs, targets = to_device(inputs, dev), to_device(targets, dev)
inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
step = TrainStep(m, lr=0.0, weight_decay=0.0, max_norm=0.1, use_graph=(os.environ.get("CHECK_EAGER", "0") != "1"))
losses = [float(step(inputs, targets)) for _ in range(nsteps)]
torch.cuda.synchronize()
g = {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
path = sys.argv[1]
print("RALF_GEMM_PATCH =", os.environ.get("RALF_GEMM_PATCH", "1"), "losses", losses)
if os.path.exists(path):
    ref = torch.load(path)
    bad = [k for k in g if not torch.equal(g[k], ref[k])]
    body = [k for k in bad if ".body." in k]
    print(f"{len(g) - len(bad)} of {len(g)} tensors bit-identical to {path}; differing backbone-body tensors: {len(body)}")
    for k in body[:12]:
        a, b = g[k].double().flatten(), ref[k].double().flatten()
        print(f"   differs: {k:60s} max |d| {(a - b).abs().max().item():.3e} of max |w| {b.abs().max().item():.3e}")
else:
    torch.save(g, path)
    print("saved", len(g), "tensors")
