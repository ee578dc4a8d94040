#!/bin/bash
O=gpurun_out/r02o; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | tail -3 | tee $O/pytest_gemm.txt
timeout 600 python tools/gemm_bench.py > $O/gemm_new.txt 2>&1
RALF_LIB=ralf_amd/libralf_hip_prev.so timeout 600 python tools/gemm_bench.py > $O/gemm_old.txt 2>&1
paste -d'|' $O/gemm_old.txt $O/gemm_new.txt | cut -c1-230 | head -70
timeout 900 python bench.py --steps 10 --warmup 3 --skip-cpu --skip-knn --skip-decode > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline_split'])"
( timeout 900 env AMD_LOG_LEVEL=1 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py -q -s -x > $O/eng_full_s.txt 2>&1; echo "rc=$?" >> $O/eng_full_s.txt )
grep -v "^\s*File\|^$\|Extension" $O/eng_full_s.txt | tail -30 | cut -c1-300
( timeout 900 python -m pytest "tests/test_engine_gpu.py::test_staged_backward_with_overlapped_exchange_matches_plain_step" tests/test_fullsize_gpu.py::test_b64_graph_replay_equals_eager_bf16_with_dropout -q -s -x > $O/pair_s.txt 2>&1; echo "rc=$?" >> $O/pair_s.txt )
grep -v "^\s*File\|^$\|Extension" $O/pair_s.txt | tail -12 | cut -c1-300
