#!/bin/bash
O=gpurun_out/r02s; mkdir -p $O
timeout 600 python tools/debug_n32.py 32 > $O/n32.txt 2>&1; tail -6 $O/n32.txt | cut -c1-300
RALF_GEMM_TILE=11 timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "not forced_tile" -p no:cacheprovider > $O/gemm11.txt 2>&1; grep -n "^E \|passed\|failed" $O/gemm11.txt | head -20 | cut -c1-250
timeout 900 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -12 | cut -c1-250
timeout 600 python tools/decode_once.py 3 2>&1 | tail -2
