"""debug: graphed vs eager TrainStep on the full model (small batch): gradient norm, weight delta, counters per step"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ralf_amd.engine import TrainStep
from ralf_amd.synthetic import make_batch, to_device

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
full = (sys.argv[2] if len(sys.argv) > 2 else "full") == "full"
for use_graph in (False, True):
    m = bench.build_model(dev, 10, "bfloat16")
    if not full:
        m.encoder = bench._BackboneStandIn(B, 256, 256, dev, m.rt.dtype)
    inputs, targets = m.preprocess(make_batch(B, 10, seed=1))
    inputs, targets = to_device(inputs, dev), to_device(targets, dev)
    inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
    step = TrainStep(m, use_graph=use_graph)
    P0 = step.opt.P.clone()
    for i in range(4):
        loss = step(inputs, targets).item()
        torch.cuda.synchronize()
        print(f"graph={use_graph} step {i}: loss {loss:.5f} |G| {step.opt.G.norm().item():.4e} |P-P0| {(step.opt.P - P0).norm().item():.4e} "
              f"|P16-P| {(step.opt.P16.float() - step.opt.P).norm().item():.3e} step_dev {int(step.opt.step_dev)} coef {float(step.opt.coef):.4e} gnorm {float(step.opt.grad_norm):.4e}", flush=True)
