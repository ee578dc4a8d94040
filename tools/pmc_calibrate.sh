#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts (tools/pmc_calibrate.py): tools/pmc_calibrate.sh TAG
R=${1:-cal}; ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/$R; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o t -- python3 $ROOT/tools/pmc_calibrate.py run > $O/cal_$C.log 2>&1
done
F=$(find $O/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*.db" | head -1)
python3 $ROOT/tools/pmc_calibrate.py report $F $W $O/pmc_calibration.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/pmc_calibration.txt
