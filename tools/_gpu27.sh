#!/bin/bash
O=gpurun_out/r02x; mkdir -p $O
for i in 1 2 3; do timeout 600 python3 bench.py --dp-selftest --skip-cpu --skip-knn --skip-split --skip-decode > $O/dp$i.json 2> $O/dp$i.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('$O/dp$i.json')); print('dp-selftest ms', d['ms_per_step'])"; done
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -3
