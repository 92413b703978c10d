#!/bin/bash
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_attn_v2_fp64_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "attn or attention or engine or bench or step or train" 2>&1 | tail -25) > gpurun_out/r5c_tests.log
python tools/probes/attn_keep_ab.py > gpurun_out/r5c_attn_keep_ab.txt 2>&1
tail -4 gpurun_out/r5c_tests.log; cat gpurun_out/r5c_attn_keep_ab.txt
