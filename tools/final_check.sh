#!/bin/bash
O=gpurun_out/r02final; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; echo "pytest rc=$?"
grep -n "^E \|passed\|failed" $O/pytest_all.txt | head -10 | cut -c1-250
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['knn']['nq16']['hbm_frac'])"
