#!/bin/bash
# one rocprofv3 counter pass over any tool:  tools/pmc_any.sh TAG "COUNTER COUNTER .." tools/prog.py [args]   -> gpurun_out/TAG/pmc.txt
R=$1; C=$2; shift 2
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/$R; mkdir -p $O
PROG=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc
timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc -o t -- python3 $PROG "$@" > $O/pmc.log 2>&1
DB=$(find $O/pmc -name "*.db" | head -1)
python3 $ROOT/tools/pmc_sq.py $DB $O/pmc.txt "# rocprofv3 --pmc $C --kernel-trace -- python3 $(basename $PROG) $*"
rm -rf $O/pmc
head -12 $O/pmc.txt | cut -c1-250
