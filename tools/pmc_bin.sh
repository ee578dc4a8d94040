#!/bin/bash
# one rocprofv3 counter pass over a native binary:  tools/pmc_bin.sh TAG "COUNTER COUNTER .." tools/_prog.bin [args]   -> gpurun_out/TAG.txt
# (environment variables of the program are inherited; the program itself follows `--`: no env / bash -c hop under the profiler)
R=$1; C=$2; shift 2
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; mkdir -p $O
PROG=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$R
timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$R -o t -- $PROG "$@" > $O/$R.log 2>&1
DB=$(find /tmp/pmc_$R -name "*.db" | head -1)
python3 $ROOT/tools/pmc_sq.py $DB $O/$R.txt "# rocprofv3 --pmc $C --kernel-trace -- $(basename $PROG) $* (LAB_SHAPES=$LAB_SHAPES LAB_VARIANTS=$LAB_VARIANTS)"
rm -rf /tmp/pmc_$R
head -14 $O/$R.txt | cut -c1-260
