"""bf16 throughput mode against fp32 parity mode over a long run at the benchmark's own size (B = 64, 256 x 256, N = 10): the curves of
tests/test_convergence_gpu.py continued.  python tools/convergence_long.py OUT.json [steps=600]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ralf_amd.engine import TrainStep  # noqa: E402
from ralf_amd.synthetic import make_learnable_set, to_device  # noqa: E402

out, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda", 0)
B, N, HW = 64, 10, 256
batches = make_learnable_set(1024, B, N, H=HW, W=HW)
res = {"set": f"1024 learnable synthetic samples (synthetic.make_learnable_set), B={B}, {HW}x{HW}, N={N}; lr 1e-4 (backbone x0.1), clip 0.1, weight decay 1e-4, dropout 0.1, plain init",
       "steps": steps}
for dtype in ("float32", "bfloat16"):
    model = bench.build_model(dev, N, dtype)
    devb = []
    for b in batches:
        i, t = model.preprocess(b)
        i, t = to_device(i, dev), to_device(t, dev)
        i["retrieved"] = {k: v for k, v in i["retrieved"].items() if k != "image"}
        devb.append((i, t))
    step = TrainStep(model, lr=1e-4, weight_decay=1e-4, max_norm=0.1, use_graph=True)
    t0 = time.perf_counter()
    ls = [step(*devb[k % len(devb)]).clone() for k in range(steps)]
    torch.cuda.synchronize()
    res[dtype] = {"ms_per_step": (time.perf_counter() - t0) / steps * 1e3, "loss_every_10": [round(float(x), 4) for x in torch.stack(ls).float().cpu()[::10]]}
    c = torch.stack(ls).float().cpu()
    res[dtype]["mean_last_32"] = float(c[-32:].mean())
    del step, model
a, b = torch.tensor(res["float32"]["loss_every_10"]), torch.tensor(res["bfloat16"]["loss_every_10"])
sm = lambda v: torch.nn.functional.avg_pool1d(v[None, None], 4, 1)[0, 0]  # noqa: E731
res["max_rel_gap_of_40_step_means"] = float((sm(b) / sm(a) - 1).abs().max())
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ("float32", "bfloat16")}), res["float32"]["mean_last_32"], res["bfloat16"]["mean_last_32"], res["float32"]["ms_per_step"], res["bfloat16"]["ms_per_step"])
