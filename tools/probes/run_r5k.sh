#!/bin/bash
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { env "$@" python bench.py --steps 100 --repeats 2 --no-cpu-baseline --no-extra-shapes --no-roofline 2>>gpurun_out/dp_ab.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1]); print('%-95s %.3f ms  %s' % (' '.join(sys.argv[1:]) or 'plain', d['ms_per_step'], d['config']['grad_allreduce']))" "$@"; }
(run KK_NONE=1
for g in dec3 dec3,dec0 13; do
  run KK_DP_FORCE=1 KK_DP_PROBE_KERNEL=1 KK_DP_GROUPS=$g KK_DP_DEFER=1
done
run KK_DP_FORCE=1 KK_DP_PAYLOAD=bf16 KK_DP_GROUPS=dec3,dec0 KK_DP_DEFER=1
) > gpurun_out/r5k_defer.txt 2>&1; cat gpurun_out/r5k_defer.txt
