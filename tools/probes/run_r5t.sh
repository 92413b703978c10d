#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5z; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bucket" > $o/test.log 2>&1; tail -3 $o/test.log
