#!/bin/bash
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "cast_ranges or rccl or exchange" 2>&1 | tail -4) > gpurun_out/r5h_tests.log
for a in "512 1 1" "512 0 1" "512 1 0" "1024 0 1"; do python tools/probes/attn_bwd_trace.py $a 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r5h_attn_bwd_trace.txt
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { env "$@" python bench.py --steps 100 --repeats 2 --no-cpu-baseline --no-extra-shapes --no-roofline 2>>gpurun_out/dp_ab.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1]); print('%-58s %.3f ms  %s' % (' '.join(sys.argv[1:]) or 'plain', d['ms_per_step'], d['config']['grad_allreduce']))" "$@"; }
for i in 1 2 3; do
  run KK_NONE=1
  run KK_DP_FORCE=1
  run KK_DP_FORCE=1 KK_DP_PAYLOAD=bf16
done > gpurun_out/r5h_dp_ab.txt 2>&1
cat gpurun_out/r5h_tests.log gpurun_out/r5h_attn_bwd_trace.txt gpurun_out/r5h_dp_ab.txt
