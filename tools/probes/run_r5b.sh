#!/bin/bash
# round 5, run b: keep-bit attention — the attention / engine tests, then the step with and without the stored bits (interleaved)
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_attn_v2_fp64_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "attn or attention or engine or bench or step or train" 2>&1 | tail -25) > gpurun_out/r5b_tests.log
bash tools/probes/ab.sh gpurun_out/r5b_ab512 2 "LIBV=product SET:attn_keep_bits=1" "LIBV=product SET:attn_keep_bits=0" > gpurun_out/r5b_ab512.txt 2>&1
EXTRA="--frames 1024 --phonemes 128" bash tools/probes/ab.sh gpurun_out/r5b_ab1024 2 "LIBV=product SET:attn_keep_bits=1" "LIBV=product SET:attn_keep_bits=0" > gpurun_out/r5b_ab1024.txt 2>&1
tail -5 gpurun_out/r5b_tests.log; cat gpurun_out/r5b_ab512.txt gpurun_out/r5b_ab1024.txt
