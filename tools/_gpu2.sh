cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests/test_knn_gpu.py tests/test_abi.py -x -q 2>&1 | grep -v "^$" | tail -15 > gpurun_out/r02b/pytest_knn.txt
timeout 300 python tools/knn_bench.py > gpurun_out/r02b/knn_bench.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_knn_gpu.py 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r02b/pytest_all.txt
