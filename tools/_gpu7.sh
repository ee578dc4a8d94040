cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
echo "== default" >> gpurun_out/r02g/debug2.txt
timeout 300 python tools/debug_graph2.py 16 2>&1 | grep -v "amdgpu.ids" | tail -45 >> gpurun_out/r02g/debug2.txt
echo "== single flush" >> gpurun_out/r02g/debug2.txt
RALF_WGRAD_GROUP_TILES=1000000 timeout 300 python tools/debug_graph2.py 16 2>&1 | grep -v "amdgpu.ids" | tail -25 >> gpurun_out/r02g/debug2.txt
echo "== no split" >> gpurun_out/r02g/debug2.txt
RALF_WGRAD_GROUP_WGS=1 timeout 300 python tools/debug_graph2.py 16 2>&1 | grep -v "amdgpu.ids" | tail -25 >> gpurun_out/r02g/debug2.txt
