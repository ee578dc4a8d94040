cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
echo "== GROUP_WGRADS=$1 BRANCHES=$2 full" >> gpurun_out/r02f/debug.txt
RALF_GROUP_WGRADS=$1 RALF_BRANCHES=$2 timeout 300 python tools/debug_graph.py 16 full 2>&1 | grep "graph=" >> gpurun_out/r02f/debug.txt
done
echo "== defaults, stand-in backbone" >> gpurun_out/r02f/debug.txt
timeout 300 python tools/debug_graph.py 16 encdec 2>&1 | grep "graph=" >> gpurun_out/r02f/debug.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "cgl or dropout_mask" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r02f/pytest_model.txt
