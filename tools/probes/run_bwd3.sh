#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_attn_v2_fp64_gpu.py -q -k "attn or attention" 2>&1 | tail -12 > gpurun_out/t_bwd3.log
for i in 1 2; do for v in 0 3; do for sh in "512 64" "1024 128"; do set -- $sh
  echo -n "KK_ATTN_BWD3=$v  $1: "; KK_ATTN_BWD3=$v python bench.py --lib tuning --frames $1 --phonemes $2 --steps 60 --repeats 3 --no-cpu-baseline --no-extra-shapes --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['timed_regions']['ms_per_step'])"
done; done; done > gpurun_out/bwd3_step_ab.txt 2>&1
