#!/bin/bash
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_attn_v2_fp64_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "attn or attention or engine or bench or step or train" 2>&1 | tail -25) > gpurun_out/r5d_tests.log
python tools/probes/g16x_longk_probe.py 8192 > gpurun_out/r5d_g16x_longk_8192.txt 2>&1
python tools/probes/g16x_longk_probe.py 4096 > gpurun_out/r5d_g16x_longk_4096.txt 2>&1
tail -4 gpurun_out/r5d_tests.log; cat gpurun_out/r5d_g16x_longk_8192.txt gpurun_out/r5d_g16x_longk_4096.txt
