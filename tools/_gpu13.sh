#!/bin/bash
O=gpurun_out/r02m; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_all.txt
tail -5 $O/pytest_all.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --skip-cpu --skip-knn --skip-split --skip-decode --no-overlap > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1); python tools/prof_kernels.py $DB > $O/step_stats_single_stream.txt 2>&1 || true
python tools/prof_timeline.py $DB adamw > $O/step_timeline_single_stream.txt 2>&1 || true
find $O/prof -name "*.csv" -size +5M -delete; find $O/prof -name "*.db" -delete
head -50 $O/step_stats_single_stream.txt | cut -c1-180
