"""one eager bf16 forward + backward of the full model at B = 64 on fixed weights and batch: every parameter gradient saved (argv[1]) or compared bit for bit
with a saved run (argv[1] exists) -- RALF_GEMM_PATCH=0 / 1 must give the same bits wherever the run itself is reproducible."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = bench.build_model(dev, 10, "bfloat16")
m.rt.drop_p = lambda p: 0.0
inputs, targets = m.preprocess(make_batch(64, 10, seed=1))
inputs, targets = to_device(inputs, dev), to_device(targets, dev)
o, l = m.train_loss(inputs, targets)
l["nll_loss"].backward()
torch.cuda.synchronize()
g = {k: p.grad.float().cpu() for k, p in m.named_parameters() if p.grad is not None}
g["loss"] = l["nll_loss"].detach().float().cpu()
path = sys.argv[1]
print("RALF_GEMM_PATCH =", os.environ.get("RALF_GEMM_PATCH", "1"), "loss", float(g["loss"]))
if os.path.exists(path):
    ref = torch.load(path)
    bad = [k for k in g if not torch.equal(g[k], ref[k])]
    print(f"{len(g) - len(bad)} of {len(g)} tensors bit-identical to {path}")
    for k in bad[:25]:
        a, b = g[k].double().flatten(), ref[k].double().flatten()
        print(f"   differs: {k:70s} cos {torch.nn.functional.cosine_similarity(a, b, dim=0).item():.6f} norm ratio {(a.norm() / b.norm()).item():.5f}")
else:
    torch.save(g, path)
    print("saved", len(g), "tensors")
