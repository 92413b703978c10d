#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5ln
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "norm or ln" > gpurun_out/r5ln/test.log 2>&1; grep -n "passed\|failed\|FAILED" gpurun_out/r5ln/test.log | tail -5
timeout 900 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "parity_fp32 or bf16_math_mode or bench_config" > gpurun_out/r5ln/test2.log 2>&1; grep -n "passed\|failed\|FAILED" gpurun_out/r5ln/test2.log | tail -5
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from kokoro_ruslan_amd import lib as kk
rows, H = 4096, 512
dy = torch.randn(rows, H, device='cuda').bfloat16(); x = torch.randn(rows, H, device='cuda'); g = torch.randn(H, device='cuda')
mean = x.mean(1); rstd = 1 / x.std(1)
dx = torch.zeros(rows, H, device='cuda'); nb = kk.load().kk_norm_bwd_blocks(rows, H); part = torch.empty(nb, 2 * H, device='cuda')
def run():
    kk.call("kk_layernorm_bwd", dy, x, g, mean, rstd, dx, 1, None, None, part, rows, H, 1)
for _ in range(3): run()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(10): run()
gr.replay(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): gr.replay()
e.record(); torch.cuda.synchronize()
print(f"kk_layernorm_bwd 4096x512 bf16 dy, partial rows: {s.elapsed_time(e) / 200 * 1e3:.2f} us per launch")
PY
