#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q > gpurun_out/r5o_pytest.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5o_pytest.txt | tail -3
EXTRA="--frames 1024 --phonemes 128" bash tools/probes/ab.sh gpurun_out/r5o_ab1024 3 "SET:fuse_linear_tail=0" "SET:fuse_linear_tail=1" > gpurun_out/r5o_ab1024.txt 2>&1
cat gpurun_out/r5o_ab1024.txt
