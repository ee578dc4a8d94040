cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r02i/pytest_all.txt
timeout 300 python tools/knn_bench.py 2>&1 | grep "D=" > gpurun_out/r02i/knn_bench.txt
for i in 1 2; do
RALF_BRANCHES=0 RALF_GROUP_WGRADS=0 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/base           : /' >> gpurun_out/r02i/encdec_ab.txt
RALF_BRANCHES=1 RALF_GROUP_WGRADS=1 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/branches+grouped: /' >> gpurun_out/r02i/encdec_ab.txt
RALF_BRANCHES=1 RALF_GROUP_WGRADS=1 RALF_WGRAD_GROUP_TILES=96 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/b+g tiles96     : /' >> gpurun_out/r02i/encdec_ab.txt
RALF_BRANCHES=1 RALF_GROUP_WGRADS=1 RALF_WGRAD_GROUP_TILES=48 timeout 300 python tools/encdec_once.py 30 2>&1 | tail -1 | sed 's/^/b+g tiles48     : /' >> gpurun_out/r02i/encdec_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_ed -- python3 $GRAFT_REPO_ROOT/tools/encdec_once.py 8 > $GRAFT_REPO_ROOT/gpurun_out/r02i/ed.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ed -name '*.db' | head -1)
python tools/prof_summary.py $DB gpurun_out/r02i/ed_stats.txt "# encdec only" 13
python tools/prof_timeline.py $DB adamw > gpurun_out/r02i/ed_timeline.txt 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err
