#!/bin/bash
# instruction mix per kernel of the encoder-decoder-only step (eager, single stream): tools/pmc_sq.sh TAG
R=${1:-sq}; ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/$R; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
rm -rf $O/pmc
timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc -o t -- python3 $ROOT/tools/encdec_once.py 2 single eager > $O/pmc_sq.log 2>&1
DB=$(find $O/pmc -name "*.db" | head -1)
python3 $ROOT/tools/pmc_sq.py $DB $O/encdec_sq_pmc.txt "# rocprofv3 --pmc $C --kernel-trace -- python3 tools/encdec_once.py 2 single eager ($R)"
rm -rf $O/pmc
head -30 $O/encdec_sq_pmc.txt | cut -c1-220
