"""encoder-decoder-only train step (backbone stand-in) for rocprofv3: python3 tools/encdec_once.py [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bench_encdec
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t = bench_encdec(torch.device("cuda", 0), 10, 64, "bfloat16", steps)
print(f"encoder-decoder step {t * 1e3:.2f} ms")
