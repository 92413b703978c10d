#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5final; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu > $o/test.log 2>&1; grep -n "passed\|failed\|FAILED" $o/test.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/round_artifacts.sh r05 > $o/artifacts.log 2>&1; tail -3 $o/artifacts.log | cut -c1-200
