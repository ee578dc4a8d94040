"""graph-replayed KV-cached decode (B = 256, cwh, top-k 5) for rocprofv3"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench
from ralf_amd.engine import GraphedDecode
from ralf_amd.helpers.task import get_condition
from ralf_amd.synthetic import make_batch
dev = torch.device("cuda"); B, N, task = 256, 10, "cwh"
model = bench.build_model(dev, N, "bfloat16", task).eval()
cond, _ = get_condition(make_batch(B, N, seed=9), task, model.tokenizer)
cond.retrieved = {k: v for k, v in cond.retrieved.items() if k != "image"}
cfg = {"name": "top_k", "top_k": 5, "temperature": 1.0}
dec = GraphedDecode(model, task, cfg, True)
for _ in range(4):
    model.sample(cond=cond, sampling_cfg=cfg, cond_type=task, decoder=dec)
torch.cuda.synchronize()
