#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5q_pytest.txt 2>&1
grep -n "passed\|failed\|Error" gpurun_out/r5q_pytest.txt | tail -5
EXTRA="--frames 1024 --phonemes 128" bash tools/probes/ab.sh gpurun_out/r5q_ab1024 2 "SET:self_cleaning_acc=0" "SET:self_cleaning_acc=1" > gpurun_out/r5q_ab1024.txt 2>&1
cat gpurun_out/r5q_ab1024.txt
