#!/bin/bash
O=gpurun_out/r02u; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "fused_decode or folded or backbone" 2>&1 | tail -3 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/dprof -o dec -- python3 $GRAFT_REPO_ROOT/tools/decode_once.py 3 > $GRAFT_REPO_ROOT/$O/decode_prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/dprof -name "*.db" | head -1); python tools/prof_kernels.py $DB > $O/decode_kernels.txt 2>&1
python - "$DB" > $O/decode_timeline.txt 2>&1 <<'PY'
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select start, end, name from kernels order by start").fetchall()
n = len(rows); a = n - n // 4; seg = rows[a:a + 200]
prev = None
for s, e, name in seg:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", name)[:80]
    gap = (s - prev) / 1e3 if prev else 0
    prev = e
    print(f"{(e - s) / 1e3:7.1f} gap {gap:6.1f}  {name}")
PY
find $O/dprof -name "*.db" -delete
grep "graph replay" $O/decode_prof.log; sed -n 60,110p $O/decode_timeline.txt
