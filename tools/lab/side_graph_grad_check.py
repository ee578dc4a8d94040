"""lab: whole model, B = 64, graph replay: the flat gradient buffer after ONE step with the weight gradients as graph branches (RALF_SIDE_GRAPH=0)
against the side graph (1), per parameter tensor, several replays each (a race between a deferred weight gradient and a later in-place write
of the chain would show as a tensor that differs by more than summation-order noise)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ralf_amd.engine import TrainStep  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m0 = bench.build_model(dev, 10, "bfloat16")
sd = {k: v.clone() for k, v in m0.state_dict().items()}
inputs, tgt = m0.preprocess(make_batch(B, 10, seed=3))
inputs, tgt = to_device(inputs, dev), to_device(tgt, dev)
inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
G = {}
for mode in ("0", "1"):
    os.environ["RALF_SIDE_GRAPH"] = mode
    m = bench.build_model(dev, 10, "bfloat16")
    m.load_state_dict(sd)
    step = TrainStep(m, lr=0.0, weight_decay=0.0, use_graph=True, overlap_allreduce=False)   # lr 0: every replay sees the same weights
    gs = []
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
        step(inputs, tgt)
        torch.cuda.synchronize()
        gs.append(step.opt.G.clone())
    G[mode] = (gs, step, m)
for it, (a, b) in enumerate(zip(G["0"][0], G["1"][0])):
    d = (a - b).abs()
    print(f"replay {it}: |G_side - G_branches| / |G_branches| = {(d.norm() / a.norm()).item():.3e}, max |diff| = {d.max().item():.3e} (max |G| = {a.abs().max().item():.3e})")
