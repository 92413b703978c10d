#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "linear_tail" > gpurun_out/r5n_pytest.txt 2>&1
grep -n "passed\|failed\|Error\|differs" gpurun_out/r5n_pytest.txt | tail -12
timeout 300 python tools/probes/linear_tail_trace.py > gpurun_out/r5n_trace.txt 2>&1
cat gpurun_out/r5n_trace.txt | tail -30 | grep -v "wave [1-6]"
KK_LIB=tuning timeout 600 python tools/probes/linear_tail_bench.py > gpurun_out/r5n_bench.txt 2>&1
cat gpurun_out/r5n_bench.txt | tail -9
