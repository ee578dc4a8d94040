#!/bin/bash
O=gpurun_out/r02n; mkdir -p $O
timeout 600 python tools/debug_decode256.py c > $O/decode_c.txt 2>&1; tail -25 $O/decode_c.txt | cut -c1-250
( timeout 900 env AMD_LOG_LEVEL=1 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py -q > $O/eng_full.txt 2>&1; echo "rc=$?" >> $O/eng_full.txt )
grep -v "^\s*File\|^$\|Extension" $O/eng_full.txt | head -30 | cut -c1-300
( timeout 900 python -m pytest tests/test_fullsize_gpu.py -q > $O/full_alone.txt 2>&1; echo "rc=$?" >> $O/full_alone.txt ); tail -5 $O/full_alone.txt | cut -c1-200
