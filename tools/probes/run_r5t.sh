#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5dp
timeout 900 python -m pytest tests -q -m gpu -k "rccl or exchange or loss_normalisers or comm" > gpurun_out/r5dp/test.log 2>&1; grep -n "passed\|failed\|FAILED" gpurun_out/r5dp/test.log | tail -5
