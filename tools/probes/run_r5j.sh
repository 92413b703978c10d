#!/bin/bash
# round 5: what a fork of the step's graph onto the communication branch costs on ONE GPU, by the number of exchange groups
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { env "$@" python bench.py --steps 100 --repeats 2 --no-cpu-baseline --no-extra-shapes --no-roofline 2>>gpurun_out/dp_ab.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1]); print('%-85s %.3f ms  %s' % (' '.join(sys.argv[1:]) or 'plain', d['ms_per_step'], d['config']['grad_allreduce']))" "$@"; }
for i in 1 2; do
  run KK_NONE=1
  for g in 13 6 4 3 2 1; do
    run KK_DP_FORCE=1 KK_DP_PROBE_KERNEL=1 KK_DP_GROUPS=$g
  done
  for g in 13 3 2 1; do
    run KK_DP_FORCE=1 KK_DP_PAYLOAD=bf16 KK_DP_GROUPS=$g
  done
  run KK_DP_FORCE=1 KK_DP_PAYLOAD=bf16 KK_DP_GROUPS=3 KK_DP_STREAM=main
  run KK_DP_FORCE=1 KK_DP_GROUPS=3
done > gpurun_out/r5j_dp_ab.txt 2>&1
(timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "exchange or rccl" 2>&1 | tail -3) >> gpurun_out/r5j_dp_ab.txt
cat gpurun_out/r5j_dp_ab.txt
