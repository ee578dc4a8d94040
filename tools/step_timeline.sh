#!/bin/bash
# whole train step: rocprofv3 kernel trace -> per-kernel summary + ordered timeline of one steady-state step: tools/step_timeline.sh TAG [--no-overlap]
R=${1:-st}; MODE=$2
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/$R; mkdir -p $O
SUF=${MODE:+_single}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p -o t -- python3 $ROOT/bench.py --steps 5 --warmup 2 --skip-cpu --skip-knn --skip-split --skip-decode --skip-variants --profile-pause 1 $MODE > $O/step$SUF.log 2>&1
DB=$(find $O/p -name "*.db" | head -1)
GAP=300 python3 $ROOT/tools/prof_summary.py $DB $O/step_stats$SUF.txt "# bench.py --steps 5 $MODE ($R)" 5 300
python3 $ROOT/tools/prof_timeline.py $DB adamw 2 > $O/step_timeline$SUF.txt
rm -rf $O/p
head -4 $O/step_stats$SUF.txt | cut -c1-200
