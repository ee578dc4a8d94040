cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
timeout 600 python tools/rs_bench.py > gpurun_out/r02j/rs_bench.txt 2>&1
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/r02j/pytest_engine_full.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_engine_gpu.py --deselect tests/test_fullsize_gpu.py 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r02j/pytest_rest.txt
