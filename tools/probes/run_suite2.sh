#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|AssertionError" | head -20; done > gpurun_out/suite2.txt 2>&1
