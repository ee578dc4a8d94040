#!/bin/bash
O=gpurun_out/r02q; mkdir -p $O
timeout 2000 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_all.txt
tail -5 $O/pytest_all.txt | cut -c1-200
timeout 300 python tools/debug_fault.py 0 graph 4 graph 64 2>&1 | tail -3
timeout 900 python bench.py --steps 10 --warmup 3 --skip-cpu --skip-knn --skip-decode > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline_split']['encoder_decoder']['ms'])"
