#!/bin/bash
O=gpurun_out/r02w; mkdir -p $O
for v in 0 1 2 3; do RALF_KNN_V32=$v timeout 300 python tools/knn_graph_bench.py 32 2>&1 | grep V32 >> $O/knn_v32.txt; done
timeout 300 python tools/knn_graph_bench.py 1 16 2>&1 | grep V32 >> $O/knn_v32.txt
cat $O/knn_v32.txt
