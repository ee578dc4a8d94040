#!/bin/bash
# per-queue idle analysis of a python tool's kernel trace:  tools/prof_q.sh TAG GAP_MS STEPS tools/prog.py [args] -> gpurun_out/TAG_queues.txt
R=$1; GAP=$2; STEPS=$3; shift 3
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; mkdir -p $O
PROG=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$R
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$R -o t -- python3 $PROG "$@" > $O/${R}.log 2>&1
DB=$(find /tmp/prof_$R -name "*.db" | head -1)
python3 $ROOT/tools/prof_queues.py $DB $O/${R}_queues.txt $GAP $STEPS
rm -rf /tmp/prof_$R
head -60 $O/${R}_queues.txt | cut -c1-200
