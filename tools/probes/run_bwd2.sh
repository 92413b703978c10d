#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "two_pass" 2>&1 | tail -5 > gpurun_out/t_bwd2.log
for i in 1 2; do
for tp in 1 0; do
  echo "attn_two_pass=$tp"; python bench.py --workload dyn16384 --set attn_two_pass=$tp --no-cpu-baseline --no-extra-shapes --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])"
done; done > gpurun_out/two_pass_ab.txt 2>&1
