#!/bin/bash
# launch sequence of the last window of a python tool's kernel trace:  tools/prof_seq.sh TAG GAP_MS tools/prog.py [args] -> gpurun_out/TAG_sequence.txt
R=$1; GAP=$2; shift 2
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; mkdir -p $O
PROG=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$R
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$R -o t -- python3 $PROG "$@" > $O/${R}.log 2>&1
DB=$(find /tmp/prof_$R -name "*.db" | head -1)
python3 $ROOT/tools/prof_sequence.py $DB $O/${R}_sequence.txt $GAP
rm -rf /tmp/prof_$R
head -3 $O/${R}_sequence.txt
