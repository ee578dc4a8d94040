set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02a/pytest.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_ed -- python3 $GRAFT_REPO_ROOT/tools/encdec_once.py 8 > $GRAFT_REPO_ROOT/gpurun_out/r02a/ed.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ed -name '*.db' | head -1)
python tools/prof_summary.py $DB gpurun_out/r02a/ed_stats.txt "# encdec only (round 2 start)" 8
python tools/prof_timeline.py $DB adamw > gpurun_out/r02a/ed_timeline.txt 2>&1
python tools/prof_gaps.py $DB > gpurun_out/r02a/ed_gaps.txt 2>&1
python tools/prof_streams.py $DB > gpurun_out/r02a/ed_streams.txt 2>&1
python bench.py --steps 10 --warmup 3 --skip-cpu > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
