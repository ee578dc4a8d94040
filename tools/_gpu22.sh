#!/bin/bash
O=gpurun_out/r02v; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -k "fused_decode or decode_rows or sample" 2>&1 | tail -3 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/dprof -o dec -- python3 $GRAFT_REPO_ROOT/tools/decode_once.py 3 > $GRAFT_REPO_ROOT/$O/decode_prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/dprof -name "*.db" | head -1); python tools/prof_kernels.py $DB > $O/decode_kernels.txt 2>&1
find $O/dprof -name "*.db" -delete
grep "graph replay" $O/decode_prof.log; grep "attn_decode\|mask_sample\|ln_fwd" $O/decode_kernels.txt | cut -c1-150
