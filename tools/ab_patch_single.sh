run() { env "$@" python3 bench.py --steps 30 --warmup 5 --skip-variants --skip-cpu --skip-knn --skip-decode --skip-split $EXTRA 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        print(json.loads(l)['ms_per_step'])
"; }
for i in 1 2 3; do
  EXTRA=--no-overlap; echo "single stream: patch $(run X=1)  tap $(run RALF_GEMM_PATCH=0)"
done
