#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5w; mkdir -p $o
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py tests/test_trainer_gpu.py -x -q -m gpu -k "ema or optimizer or nonfinite or trainer or multi_step or device_step" > $o/test.log 2>&1; tail -5 $o/test.log
