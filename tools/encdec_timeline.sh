#!/bin/bash
# encoder-decoder-only step: rocprofv3 kernel trace -> per-kernel summary + ordered timeline of one steady-state step
#   tools/encdec_timeline.sh TAG [single]   (writes gpurun_out/TAG/encdec_{stats,timeline}[_single].txt)
R=${1:-tl}; MODE=$2
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/$R; mkdir -p $O
SUF=${MODE:+_$MODE}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p -o t -- python3 $ROOT/tools/encdec_once.py 8 $MODE > $O/encdec$SUF.log 2>&1
DB=$(find $O/p -name "*.db" | head -1)
GAP=300 python3 $ROOT/tools/prof_summary.py $DB $O/encdec_stats$SUF.txt "# encdec_once.py 8 $MODE ($R)" 8 300
python3 $ROOT/tools/prof_timeline.py $DB adamw 2 > $O/encdec_timeline$SUF.txt
rm -rf $O/p
grep "encoder-decoder only" $O/encdec$SUF.log
head -4 $O/encdec_stats$SUF.txt | cut -c1-200
