#!/bin/bash
O=gpurun_out/r02ac; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; echo "pytest rc=$?"
grep -n "^E \|passed\|failed" $O/pytest_all.txt | head -10 | cut -c1-250
for i in 1 2; do timeout 900 python bench.py --steps 20 --warmup 3 --skip-cpu --skip-knn --skip-decode > $O/bench$i.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench$i.json')); print(d['ms_per_step'], d['roofline_split']['encoder_decoder']['ms'])"; done
