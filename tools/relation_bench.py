"""bench.bench_relation alone (BASELINE configs[4], relationship task at B = 256): exact mode in lock-step, the sequential loop's rate, rng=per_sample
    python tools/relation_bench.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    r = bench.bench_relation(torch.device("cuda", 0))
    print(json.dumps(r, indent=1))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(r, f, indent=1)
