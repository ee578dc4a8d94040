#!/bin/bash
# same-box A/B of the encoder-decoder-only step under environment variants:  tools/ab_encdec.sh "VAR=1" "VAR=2 OTHER=3" ...
# (first run = baseline without variables); prints ms per step for each, twice (ABAB) to expose box drift.  Every run is bounded (a runtime
# knob that hangs the replay must not eat the call) and the lines land in gpurun_out/ab_encdec.txt as they are produced.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { echo "$1: $(timeout 90 env $1 python3 tools/encdec_once.py 20 2>/dev/null | grep -o '[0-9.]* ms per step' || echo 'failed / timed out')" | tee -a gpurun_out/ab_encdec.txt; }
echo "--- $(date +%H:%M:%S) $*" >> gpurun_out/ab_encdec.txt
for rep in 1 2; do
  one "BASE=1"
  for v in "$@"; do one "$v"; done
done
