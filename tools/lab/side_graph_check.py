"""lab: full model (ResNet-50 + FPN in front), B = 4, graph replay: losses of 4 steps with the weight gradients as graph branches (RALF_SIDE_GRAPH=0),
as a side graph (1), and twice each (run-to-run noise)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ralf_amd.engine import TrainStep  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402

dev = torch.device("cuda", 0)
m0 = bench.build_model(dev, 10, "bfloat16")
sd = {k: v.clone() for k, v in m0.state_dict().items()}
inputs, tgt = m0.preprocess(make_batch(4, 10, seed=3))
inputs, tgt = to_device(inputs, dev), to_device(tgt, dev)
inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
res = {}
for tag, mode in [("branches a", "0"), ("branches b", "0"), ("side graph a", "1"), ("side graph b", "1"), ("auto", "auto")]:
    os.environ["RALF_SIDE_GRAPH"] = mode
    m = bench.build_model(dev, 10, "bfloat16")
    m.load_state_dict(sd)
    step = TrainStep(m, use_graph=True, overlap_allreduce=False)
    ls = [step(inputs, tgt).item() for _ in range(4)]
    torch.cuda.synchronize()
    res[tag] = (ls, step.opt.G.clone(), step.opt.P.clone())
    print(f"{tag:14s} losses {' '.join(f'{x:.6f}' for x in ls)}  chosen side_graph={step.side_graph} {getattr(step, 'side_graph_ms', None)}")
g0, p0 = res["branches a"][1], res["branches a"][2]
for tag, (ls, g, p) in res.items():
    print(f"{tag:14s} |G - G0| / |G0| = {((g - g0).norm() / g0.norm()).item():.3e}   max |P - P0| = {(p - p0).abs().max().item():.3e}")
