#!/bin/bash
# same-box A/B of the whole train step under bench.py flag / environment variants: tools/ab_bench_flags.sh "--no-overlap" "ENV=1 --flag" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
F="--steps 30 --warmup 5 --skip-cpu --skip-knn --skip-split --skip-decode --skip-variants"
one() { # variant string: leading VAR=val words are environment, the rest flags
  envs=(); flags=()
  for w in $1; do if [[ $w == *=* && $w != --* ]]; then envs+=("$w"); else flags+=("$w"); fi; done
  echo "[$1] $(timeout 200 env "${envs[@]}" X=1 python3 bench.py $F "${flags[@]}" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms')" || echo failed)" | tee -a gpurun_out/ab_bench_flags.txt
}
echo "--- $(date +%H:%M:%S)" >> gpurun_out/ab_bench_flags.txt
for rep in 1 2; do
  one ""
  for v in "$@"; do one "$v"; done
done
