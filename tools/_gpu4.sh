cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
for i in 1 2; do
RALF_BRANCHES=0 RALF_STACKED_KV=0 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/base            : /' >> gpurun_out/r02d/encdec_ab.txt
RALF_BRANCHES=0 RALF_STACKED_KV=1 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/stackedKV       : /' >> gpurun_out/r02d/encdec_ab.txt
RALF_BRANCHES=1 RALF_STACKED_KV=1 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/stackedKV+branch: /' >> gpurun_out/r02d/encdec_ab.txt
done
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r02d/pytest_engine.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_knn -- python3 $GRAFT_REPO_ROOT/tools/knn_once2.py > $GRAFT_REPO_ROOT/gpurun_out/r02d/knn_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_kernels.py $(find /tmp/prof_knn -name '*.db' | head -1) knn > gpurun_out/r02d/knn_kernels.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_knn_gpu.py --deselect tests/test_engine_gpu.py --deselect tests/test_gemm_gpu.py --deselect tests/test_fullsize_gpu.py 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r02d/pytest_rest.txt
