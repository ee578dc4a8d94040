#!/bin/bash
# same-box interleaved A/B of this round's switchable kernels: the default configuration against the per-operation paths
# (RALF_FUSED_LAYERS / RALF_FUSED_FFN / RALF_FUSED_FFN_BWD / RALF_ATTN_BWD_FUSED = 0), N repetitions each, alternating.
#   tools/same_box_ab.sh OUT.json [N=4]
OUT=$1; N=${2:-4}
run() { env "$@" python3 bench.py --steps 30 --warmup 5 --skip-variants --skip-cpu --skip-knn --skip-decode 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline_split']['encoder_decoder']['ms'])
"; }
A=(); B=()
for i in $(seq $N); do
  A+=("$(run X=1)")
  B+=("$(run RALF_FUSED_LAYERS=0 RALF_FUSED_FFN=0 RALF_FUSED_FFN_BWD=0 RALF_ATTN_BWD_FUSED=0)")
done
python3 - "$OUT" "${A[@]}" -- "${B[@]}" <<'PY'
import json, sys, statistics
out = sys.argv[1]; rest = sys.argv[2:]; k = rest.index('--')
def parse(xs): return [tuple(float(v) for v in x.split()) for x in xs]
a, b = parse(rest[:k]), parse(rest[k + 1:])
rep = {"command": "tools/same_box_ab.sh (bench.py --steps 30 --warmup 5, alternating runs on one box)",
       "default": {"step_ms": [x[0] for x in a], "encoder_decoder_ms": [x[1] for x in a]},
       "per_operation_paths (RALF_FUSED_LAYERS=0 RALF_FUSED_FFN=0 RALF_FUSED_FFN_BWD=0 RALF_ATTN_BWD_FUSED=0)": {"step_ms": [x[0] for x in b], "encoder_decoder_ms": [x[1] for x in b]},
       "median_step_ms": [statistics.median(x[0] for x in a), statistics.median(x[0] for x in b)],
       "median_encoder_decoder_ms": [statistics.median(x[1] for x in a), statistics.median(x[1] for x in b)]}
json.dump(rep, open(out, "w"), indent=1)
print(json.dumps(rep["median_step_ms"]), json.dumps(rep["median_encoder_decoder_ms"]))
PY
