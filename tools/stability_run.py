import sys, torch
sys.path.insert(0, ".")
import bench
from ralf_amd.engine import TrainStep
from ralf_amd.synthetic import make_batch, to_device
dev = torch.device("cuda")
model = bench.build_model(dev, 10, "bfloat16")
step = TrainStep(model, lr=1e-4, weight_decay=1e-4, max_norm=0.1)
batches = []
for s in range(4):
    i, t = model.preprocess(make_batch(64, 10, seed=1 + s))
    i, t = to_device(i, dev), to_device(t, dev)
    i["retrieved"] = {k: v for k, v in i["retrieved"].items() if k != "image"}
    batches.append((i, t))
losses = []
for n in range(240):
    i, t = batches[n % 4]
    l = step(i, t)
    if n % 20 == 19:
        torch.cuda.synchronize()
        losses.append(round(float(l), 4))
        print(n + 1, losses[-1], "mem MB", torch.cuda.memory_allocated() // 2**20, "grad norm", round(float(step.opt.grad_norm), 3), flush=True)
assert all(x == x for x in losses) and losses[-1] < losses[0], losses
print("stable: loss", losses[0], "->", losses[-1])
