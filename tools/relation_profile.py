"""cProfile of the host side of sample_relation(rng="per_sample") at B = 256 (bench.bench_relation's workload)"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

orig = bench.time.perf_counter
pr = cProfile.Profile()
import ralf_amd.models.ralf as R  # noqa: E402
_sr = R._GeneratorBase.sample_relation
calls = [0]


def wrapped(self, *a, **k):
    calls[0] += 1
    if k.get("rng") == "per_sample" and calls[0] >= 4:     # the timed per-sample call (after both warm-ups and the exact-mode call)
        pr.enable()
        try:
            return _sr(self, *a, **k)
        finally:
            pr.disable()
    return _sr(self, *a, **k)


R._GeneratorBase.sample_relation = wrapped
out = bench.bench_relation(torch.device("cuda", 0), 10, int(sys.argv[1]) if len(sys.argv) > 1 else 256)
print({k: v for k, v in out.items() if k != "note"})
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
