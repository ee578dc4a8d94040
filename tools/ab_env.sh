#!/bin/bash
# same-box interleaved A/B of ONE environment switch on the whole train step:  tools/ab_env.sh OUT.json VAR=VALUE [N=3]
# (bench.py --steps ${STEPS:-30} --warmup 5, main timing only, alternating runs)
OUT=$1; SW=$2; N=${3:-3}
run() { env "$@" python3 bench.py --steps ${STEPS:-30} --warmup 5 --skip-variants --skip-cpu --skip-knn --skip-decode --skip-split 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        print(json.loads(l)['ms_per_step'])
"; }
A=(); B=()
for i in $(seq $N); do
  A+=("$(run X=1)")
  B+=("$(run $SW)")
done
python3 - "$OUT" "$SW" "${A[@]}" -- "${B[@]}" <<'PY'
import json, sys, statistics
out, sw = sys.argv[1], sys.argv[2]; rest = sys.argv[3:]; k = rest.index('--')
a, b = [float(x) for x in rest[:k]], [float(x) for x in rest[k + 1:]]
rep = {"command": "tools/ab_env.sh (bench.py --steps ${STEPS:-30} --warmup 5, alternating runs on one box)", "default_step_ms": a, sw + "_step_ms": b,
       "median_step_ms": {"default": statistics.median(a), sw: statistics.median(b)}}
json.dump(rep, open(out, "w"), indent=1)
print(json.dumps(rep))
PY
