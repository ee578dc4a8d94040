#!/bin/bash
# per-kernel durations of the graph-replayed train step under two settings of an environment knob (single stream, windowed):
#   tools/ab_profile.sh OUTDIR "VAR=a ..." "VAR=b ..."
O=$1; shift
ROOT=$GRAFT_REPO_ROOT; mkdir -p $ROOT/$O
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  rm -rf $ROOT/$O/p_$i
  ( export $cfg; timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$O/p_$i -o t -- python3 $ROOT/bench.py --steps 5 --warmup 2 --skip-cpu --skip-knn --skip-split --skip-decode --skip-variants --profile-pause 1 --no-overlap > $ROOT/$O/run_$i.log 2>&1 )
  DB=$(find $ROOT/$O/p_$i -name "*.db" | head -1)
  python3 $ROOT/tools/prof_summary.py $DB $ROOT/$O/stats_$i.txt "# $cfg" 5 300
  rm -rf $ROOT/$O/p_$i
  head -3 $ROOT/$O/stats_$i.txt | cut -c1-250
done
