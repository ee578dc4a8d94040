cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_knn_gpu.py -x -q 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r02e/pytest_gemm_knn.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_engine_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r02e/pytest_engine.txt
for i in 1 2; do
RALF_BRANCHES=0 RALF_GROUP_WGRADS=0 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/base           : /' >> gpurun_out/r02e/encdec_ab.txt
RALF_BRANCHES=1 RALF_GROUP_WGRADS=0 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/branches       : /' >> gpurun_out/r02e/encdec_ab.txt
RALF_BRANCHES=0 RALF_GROUP_WGRADS=1 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/grouped        : /' >> gpurun_out/r02e/encdec_ab.txt
RALF_BRANCHES=1 RALF_GROUP_WGRADS=1 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/branches+grouped: /' >> gpurun_out/r02e/encdec_ab.txt
done
timeout 300 python tools/knn_bench.py > gpurun_out/r02e/knn_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_ed -- python3 $GRAFT_REPO_ROOT/tools/encdec_once.py 8 > $GRAFT_REPO_ROOT/gpurun_out/r02e/ed.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ed -name '*.db' | head -1)
python tools/prof_summary.py $DB gpurun_out/r02e/ed_stats.txt "# encdec only (branches + grouped wgrads)" 13
python tools/prof_timeline.py $DB adamw > gpurun_out/r02e/ed_timeline.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_knn_gpu.py --deselect tests/test_engine_gpu.py --deselect tests/test_gemm_gpu.py --deselect tests/test_fullsize_gpu.py 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r02e/pytest_rest.txt
