#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "regulate_embed or bucket" > gpurun_out/r5w_pytest1.txt 2>&1
grep -n "passed\|failed\|Error" gpurun_out/r5w_pytest1.txt | tail -5
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q > gpurun_out/r5w_pytest2.txt 2>&1
grep -n "passed\|failed\|Error" gpurun_out/r5w_pytest2.txt | tail -5
bash tools/probes/ab.sh gpurun_out/r5w_ab512 3 "SET:fuse_memory_fwd=0" "SET:fuse_memory_fwd=1" > gpurun_out/r5w_ab512.txt 2>&1
cat gpurun_out/r5w_ab512.txt
