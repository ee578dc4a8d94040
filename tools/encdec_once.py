"""encoder-decoder-only train step (backbone replaced by a fixed feature sequence) replayed a few times: the program profiled by
rocprofv3 for profiles/*_encoder_decoder_only_kernel_stats.txt.   python tools/encdec_once.py [steps] [batch] [elements]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
t = bench.bench_encdec(torch.device("cuda", 0), N, B, "bfloat16", steps)
print(f"encoder-decoder only: {t * 1e3:.3f} ms per step (B={B}, N={N})")
