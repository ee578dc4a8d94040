#!/bin/bash
O=gpurun_out/r02af; mkdir -p $O
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_retrieval_gpu.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"
grep -n "^E \|passed\|failed" $O/pytest.txt | head -10 | cut -c1-250
timeout 300 python tools/knn_graph_bench.py 1 16 32 2>&1 | grep V32
timeout 300 python tools/knn_graph_bench.py 1 16 32 2>&1 | grep V32
