#!/bin/bash
O=gpurun_out/r02aa; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -4 | cut -c1-250
timeout 600 python tools/gemm_bench.py 2>&1 | grep "^conv" 
for i in 1 2; do timeout 900 python bench.py --steps 20 --warmup 3 --skip-cpu --skip-knn --skip-decode > $O/bench$i.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench$i.json')); print(d['ms_per_step'], d['roofline_split']['encoder_decoder']['ms'], d['config']['final_loss'])"; done
