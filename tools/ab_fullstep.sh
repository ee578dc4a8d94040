#!/bin/bash
# same-box A/B of the WHOLE train step (backbone included): this tree against the round-4 tree staged under tools/lab/r04 (tools/lab is not tracked)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
F="--steps 30 --warmup 5 --skip-cpu --skip-knn --skip-split --skip-decode --skip-variants"
one() { # label, dir, env...
  L=$1; D=$2; shift 2
  (cd $D && timeout 200 env "$@" python3 bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['ms_per_step'],3), 'ms')") | tee -a gpurun_out/ab_fullstep.txt
}
echo "--- $(date +%H:%M:%S)" >> gpurun_out/ab_fullstep.txt
for rep in 1 2; do
  one "r05      " . X=1
  one "r04      " tools/lab/r04 X=1
  for v in "$@"; do one "r05 $v" . $v; done
done
