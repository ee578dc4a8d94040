"""encoder-decoder-only train step (backbone replaced by a fixed feature sequence) replayed a few times: the program profiled by
rocprofv3 for profiles/*_encoder_decoder_only_kernel_stats*.txt.
    python tools/encdec_once.py [steps] [batch] [elements] [single] [eager]
single: parameter-gradient kernels and the sub-network branches on the main stream (kernel durations are not inflated by overlap).
eager: no hipGraph (PMC passes: every dispatch attributed).  Set-up and the timed replays are separated by an idle second, which
tools/prof_summary.py uses to keep only the replays."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = sys.argv[1:]
flags = {a for a in args if not a.isdigit()}
nums = [int(a) for a in args if a.isdigit()]
steps = nums[0] if len(nums) > 0 else 8
B = nums[1] if len(nums) > 1 else 64
N = nums[2] if len(nums) > 2 else 10
t = bench.bench_encdec(torch.device("cuda", 0), N, B, "bfloat16", steps, use_graph="eager" not in flags, overlap="single" not in flags, pause_s=1.0)
print(f"encoder-decoder only: {t * 1e3:.3f} ms per step (B={B}, N={N}, {sorted(flags)})")
