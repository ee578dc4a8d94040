"""Which draws does the relationship-task decode make?  Per sample of the benchmark's workload: the randint ranges it asks for, in order
(range-one draws have a known value; the others serialise the exact mode).  python tools/relation_draw_stats.py [B=64] [sharpen]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sharpen = float(sys.argv[2]) if len(sys.argv) > 2 else None
model, cond, sub = bench.relation_workload(torch.device("cuda", 0), 10, B, "float32" if os.environ.get("FP32") else "bfloat16", sharpen)
cfg = {"name": "deterministic", "temperature": 1.0}
adv = model._relation_advance
steps = [0]


def counted(st, logits, may_draw, env):
    steps[0] += 1
    return adv(st, logits, may_draw, env)


model._relation_advance = counted
real = random.randint
calls = []
random.randint = lambda a, b: (calls.append((a, b)), real(a, b))[1]
random.seed(77)
per = []
import copy
for b in range(B):
    c2 = copy.copy(cond)
    c2.image, c2.seq = cond.image[b:b + 1], cond.seq[b:b + 1].clone()
    c2.mask = cond.mask[b:b + 1] if torch.is_tensor(getattr(cond, "mask", None)) else getattr(cond, "mask", None)
    c2.retrieved = {k: v[b:b + 1] for k, v in cond.retrieved.items()}
    if hasattr(cond, "id"):
        c2.id = cond.id[b:b + 1]
    n0, s0, t0 = len(calls), steps[0], time.perf_counter()
    model.sample(cond=c2, sampling_cfg=cfg, cond_type="relation", return_violation=False, use_backtrack=True, lockstep=False)
    per.append((steps[0] - s0, calls[n0:], time.perf_counter() - t0))
random.randint = real
wide = [sum(1 for a, b in c if b > a) for _, c, _ in per]
first_wide_step = []
print(f"B={B} sharpen={sharpen}: steps per sample mean {sum(p[0] for p in per) / B:.0f} max {max(p[0] for p in per)}; draws per sample mean {sum(len(p[1]) for p in per) / B:.1f}; "
      f"samples with a wide draw {sum(1 for w in wide if w)} / {B}; wide draws per sample mean {sum(wide) / B:.2f} max {max(wide)}; ms per sample {1e3 * sum(p[2] for p in per) / B:.1f}")
import collections
print("ranges of the wide draws:", collections.Counter(b for _, c, _ in per for a, b in c if b > a).most_common(12))
print("per sample (steps, draws, wide):", [(p[0], len(p[1]), w) for p, w in zip(per, wide)][:40])
