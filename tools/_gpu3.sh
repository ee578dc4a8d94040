cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_gemm_gpu.py -x -q 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r02c/pytest_engine.txt
timeout 300 python tools/micro_ln.py > gpurun_out/r02c/micro_ln.txt 2>&1
timeout 300 python tools/encdec_once.py 20 > gpurun_out/r02c/encdec.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_knn -- python3 $GRAFT_REPO_ROOT/tools/knn_once2.py > $GRAFT_REPO_ROOT/gpurun_out/r02c/knn_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_kernels.py $(find /tmp/prof_knn -name '*.db' | head -1) knn > gpurun_out/r02c/knn_kernels.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_knn_gpu.py --deselect tests/test_engine_gpu.py --deselect tests/test_gemm_gpu.py 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r02c/pytest_rest.txt
