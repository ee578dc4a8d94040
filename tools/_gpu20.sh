#!/bin/bash
O=gpurun_out/r02t; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_all.txt
grep -n "^E \|passed\|failed" $O/pytest_all.txt | head -20 | cut -c1-250
timeout 600 python tools/decode_once.py 3 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 3 --skip-cpu --skip-knn > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline_split']['encoder_decoder']['ms']); print({k:(v['ms_per_batch'],v['graph_ms']) for k,v in d['decode'].items() if isinstance(v,dict)})"
