#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/probes/dyn_stress.py 60 bf16 > gpurun_out/r5x_dyn.txt 2>&1
tail -2 gpurun_out/r5x_dyn.txt | cut -c1-300
: > gpurun_out/r5x_soak.txt
for i in $(seq 1 10); do timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_attn_v2_fp64_gpu.py -q -k "attn or attention or linear_tail or regulate or embed_ln" -p no:cacheprovider 2>&1 | grep -E "passed|failed" >> gpurun_out/r5x_soak.txt; done
sort gpurun_out/r5x_soak.txt | uniq -c
