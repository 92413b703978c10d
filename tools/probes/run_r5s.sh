#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5s_pytest.txt 2>&1
grep -n "passed\|failed\|Error" gpurun_out/r5s_pytest.txt | tail -5
