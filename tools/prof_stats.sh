#!/bin/bash
# rocprofv3 --kernel-trace of a python tool, summarised per kernel:  tools/prof_stats.sh TAG STEPS tools/prog.py [args] -> gpurun_out/TAG_kernel_stats.txt
R=$1; STEPS=$2; shift 2
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; mkdir -p $O
PROG=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$R
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$R -o t -- python3 $PROG "$@" > $O/${R}.log 2>&1
DB=$(find /tmp/prof_$R -name "*.db" | head -1)
python3 $ROOT/tools/prof_summary.py $DB $O/${R}_kernel_stats.txt "# rocprofv3 --kernel-trace --stats -- python3 $(basename $PROG) $* (MI355X)" $STEPS $GAP
rm -rf /tmp/prof_$R
grep -v "simple_timer\|amdgpu.ids" $O/${R}.log | tail -3; head -20 $O/${R}_kernel_stats.txt | cut -c1-170
