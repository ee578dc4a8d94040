#!/bin/bash
O=gpurun_out/r02y; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_all.txt
grep -n "^E \|passed\|failed" $O/pytest_all.txt | head -20 | cut -c1-250
timeout 900 python bench.py --steps 10 --warmup 3 --skip-cpu --skip-knn --skip-decode > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline_split']['encoder_decoder']['ms'])"
timeout 600 python3 bench.py --dp-selftest --skip-cpu --skip-knn --skip-split --skip-decode > $O/dp.json 2> $O/dp.err; echo "dp rc=$?"; python3 -c "
import json; d=json.load(open('$O/dp.json')); print('dp-selftest ms', d['ms_per_step'], d['config'].get('data_parallel'))"
