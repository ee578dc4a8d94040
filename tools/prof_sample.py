import cProfile, pstats, sys, time, torch
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
for _ in range(3):
    model.sample(cond=cond, sampling_cfg=cfg, cond_type=task, decoder=dec)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    model.sample(cond=cond, sampling_cfg=cfg, cond_type=task, decoder=dec)
torch.cuda.synchronize()
print("sample(): %.1f ms per batch" % ((time.perf_counter() - t0) / 5 * 1e3))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); dec._graph.replay(); e1.record(); torch.cuda.synchronize()
print("graph replay alone: %.1f ms" % e0.elapsed_time(e1))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    model.sample(cond=cond, sampling_cfg=cfg, cond_type=task, decoder=dec)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
