#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5v; mkdir -p $o
for d in 0 1 2 3; do echo "DBG=$d"; KK_BE_DBG=$d KK_BE_ROWS=512 python tools/probes/bucket_embed_bench.py 2>&1 | grep KK_BE; done | tee $o/bench_dbg.txt
