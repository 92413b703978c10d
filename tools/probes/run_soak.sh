#!/bin/bash
# repeat the attention tests (races show up as rare single-element errors) and the whole suite
mkdir -p gpurun_out
: > gpurun_out/soak.txt
for i in $(seq 1 25); do timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_attn_v2_fp64_gpu.py -q -k "attn or attention" -p no:cacheprovider 2>&1 | grep -E "passed|failed" >> gpurun_out/soak.txt; done
for i in 1 2; do timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E " passed| failed|^FAILED" >> gpurun_out/soak.txt; done
