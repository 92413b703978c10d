# What the data-parallel exchange costs the step on ONE GPU (no multi-GPU box: scaling stays unmeasured).
# Interleaved runs of bench.py: plain step | in-graph 1-rank RCCL exchange (fp32 and bf16 payload) | the torch.distributed
# all-reduce between the backward and optimizer graphs (KK_DP_LEGACY=1) | the fallback branch taken on purpose.
run() { env "$@" python bench.py --steps 100 --repeats 2 --no-cpu-baseline --no-extra-shapes --no-roofline 2>>gpurun_out/dp_ab.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1]); print('%-58s %.3f ms  %s' % (' '.join(sys.argv[1:]) or 'plain', d['ms_per_step'], d['config']['grad_allreduce']))" "$@"; }
for i in 1 2 3; do
  run KK_NONE=1
  run KK_DP_FORCE=1
  run KK_DP_FORCE=1 KK_DP_PAYLOAD=bf16
  run KK_DP_FORCE=1 KK_DP_LEGACY=1
  run KK_DP_FORCE=1 KK_BENCH_TEST_FALLBACK=1
done
