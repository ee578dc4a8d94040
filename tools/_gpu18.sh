#!/bin/bash
O=gpurun_out/r02r; mkdir -p $O
timeout 600 python tools/debug_n32.py 32 > $O/n32.txt 2>&1; tail -30 $O/n32.txt | cut -c1-200
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_gemm_gpu.py -x -q 2>&1 | tail -5 | cut -c1-200
timeout 600 python tools/decode_once.py 3 2>&1 | tail -2
