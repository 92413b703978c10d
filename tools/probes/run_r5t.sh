#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5dp; mkdir -p $o
run() { env $1 python bench.py --steps 100 --repeats 2 --no-cpu-baseline --no-extra-shapes --no-roofline $2 2>>$o/dp_ab.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1]); print('%-70s %.3f ms  %s' % (' '.join(sys.argv[1:]) or 'plain', d['ms_per_step'], d['config']['grad_allreduce']))" "$1" "$2"; }
for i in 1 2; do
  run KK_NONE=1 ""
  run KK_NONE=1 "--set tail_aside=0"
  run "KK_DP_FORCE=1 KK_DP_PROBE_KERNEL=1" ""
  run "KK_DP_FORCE=1 KK_DP_PROBE_KERNEL=1" "--set tail_aside=0"
  run "KK_DP_FORCE=1 KK_DP_PAYLOAD=bf16" ""
  run "KK_DP_FORCE=1 KK_DP_PAYLOAD=bf16" "--set tail_aside=0"
  run "KK_DP_FORCE=1 KK_DP_PAYLOAD=bf16 KK_DP_GROUPS=1" ""
done | tee $o/dp_ab2.txt
