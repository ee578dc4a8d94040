"""debug: which parameter gradients differ between the eager and the graphed first step (full model)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ralf_amd.engine import TrainStep
from ralf_amd.synthetic import make_batch, to_device

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ms, steps = [], []
for use_graph in (False, True):
    m = bench.build_model(dev, 10, "bfloat16")
    m.rt.drop_p = lambda p: 0.0
    if ms:
        m.load_state_dict(ms[0].state_dict())
    ms.append(m)
inputs, targets = ms[0].preprocess(make_batch(B, 10, seed=1))
inputs, targets = to_device(inputs, dev), to_device(targets, dev)
inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
a, b = TrainStep(ms[0], use_graph=False), TrainStep(ms[1], use_graph=True)
la, lb = a(inputs, targets).item(), b(inputs, targets).item()
torch.cuda.synchronize()
print("loss eager/graph", la, lb, "|G|", a.opt.G.norm().item(), b.opt.G.norm().item())
bad = []
n1, n2 = dict(ms[0].named_parameters()), dict(ms[1].named_parameters())
for k, p in n1.items():
    if p.grad is None:
        continue
    g1, g2 = p.grad, n2[k].grad
    r = ((g1 - g2).norm() / g1.norm().clamp_min(1e-20)).item()
    if not (r < 0.05):
        bad.append((r, k, tuple(p.shape), g1.norm().item(), g2.norm().item()))
bad.sort(reverse=True)
print(len(bad), "parameters differ")
for r, k, shp, a_, b_ in bad[:40]:
    print(f"  rel {r:10.3e}  {k:70s} {str(shp):22s} |eager| {a_:.3e} |graph| {b_:.3e}")
