cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
timeout 600 python tools/rs_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r02k/rs_bench.txt
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
echo "== GROUP_WGRADS=$1 BRANCHES=$2" >> gpurun_out/r02k/staged.txt
RALF_GROUP_WGRADS=$1 RALF_BRANCHES=$2 timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "staged" 2>&1 | grep -v "^$" | grep -v "^  File\|^Extension" | tail -8 | cut -c1-300 >> gpurun_out/r02k/staged.txt
done
