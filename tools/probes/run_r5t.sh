#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attn_v2_fp64_gpu.py tests/test_kernels_gpu.py -x -q -k "attn or attention" > gpurun_out/r5t_pytest1.txt 2>&1
grep -n "passed\|failed\|Error" gpurun_out/r5t_pytest1.txt | tail -5
timeout 600 python tools/probes/attn_bwd_trace.py > gpurun_out/r5t_trace.txt 2>&1
grep -A6 "S=512 causal=0 keep_bits=1 dK/dV" gpurun_out/r5t_trace.txt | awk -F'\\|\\|' '{print $NF}'
timeout 600 python tools/probes/attn_keep_ab.py > gpurun_out/r5t_attn_ab.txt 2>&1
grep "B=" gpurun_out/r5t_attn_ab.txt
