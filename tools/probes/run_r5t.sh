#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5y; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu > $o/test.log 2>&1; grep -n "passed\|failed\|FAILED" $o/test.log | tail -8
