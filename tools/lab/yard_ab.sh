for i in 1 2 3; do
for p in 1 0; do
RALF_GEMM_PATCH=$p python - <<'PY'
import os, sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import grad_yardstick as gy
res = gy.yardstick(400, torch.device("cuda", 0))
h = res["hip_bf16_vs_hip_fp32"]
print("PATCH", os.environ["RALF_GEMM_PATCH"], {k.split("body.")[-1]: v for k, v in h.items() if ".body." in k}, flush=True)
PY
done; done
