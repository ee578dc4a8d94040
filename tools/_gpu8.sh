cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout 600 python -m pytest tests/test_knn_gpu.py -x -q 2>&1 | grep -v "^$" | tail -8 > gpurun_out/r02h/pytest_knn.txt
timeout 300 python tools/knn_bench.py 2>&1 | grep "D=" > gpurun_out/r02h/knn_bench.txt
echo "== default full (dropout on)" >> gpurun_out/r02h/debug.txt
timeout 300 python tools/debug_graph.py 16 full 2>&1 | grep "graph=" >> gpurun_out/r02h/debug.txt
timeout 1800 python -m pytest tests/test_fullsize_gpu.py tests/test_engine_gpu.py tests/test_model_gpu.py -x -q 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r02h/pytest_engine.txt
for i in 1 2; do
RALF_BRANCHES=0 RALF_GROUP_WGRADS=0 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/base           : /' >> gpurun_out/r02h/encdec_ab.txt
RALF_BRANCHES=1 RALF_GROUP_WGRADS=0 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/branches       : /' >> gpurun_out/r02h/encdec_ab.txt
RALF_BRANCHES=1 RALF_GROUP_WGRADS=1 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/branches+grouped: /' >> gpurun_out/r02h/encdec_ab.txt
RALF_BRANCHES=1 RALF_GROUP_WGRADS=1 RALF_WGRAD_GROUP_WGS=512 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/b+g wgs512      : /' >> gpurun_out/r02h/encdec_ab.txt
RALF_BRANCHES=1 RALF_GROUP_WGRADS=1 RALF_WGRAD_GROUP_TILES=96 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/b+g tiles96     : /' >> gpurun_out/r02h/encdec_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_ed -- python3 $GRAFT_REPO_ROOT/tools/encdec_once.py 8 > $GRAFT_REPO_ROOT/gpurun_out/r02h/ed.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ed -name '*.db' | head -1)
python tools/prof_summary.py $DB gpurun_out/r02h/ed_stats.txt "# encdec only (branches late-issued + grouped wgrads + LN-dropout)" 13
python tools/prof_timeline.py $DB adamw > gpurun_out/r02h/ed_timeline.txt 2>&1
