#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5y; mkdir -p $o
timeout 900 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu -k "legacy" > $o/test2.log 2>&1; tail -30 $o/test2.log
