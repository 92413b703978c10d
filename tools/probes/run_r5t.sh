#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5u; mkdir -p $o
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "memory_tail or bench_config or graphed_accumulation or zero_fill" > $o/test.log 2>&1; tail -3 $o/test.log
bash tools/probes/ab.sh $o/ab512 3 "SET:reduce_beside_tail=0" "SET:reduce_beside_tail=1" | tee $o/ab512.txt
EXTRA="--frames 1024 --phonemes 128" bash tools/probes/ab.sh $o/ab1024 2 "SET:reduce_beside_tail=0" "SET:reduce_beside_tail=1" | tee $o/ab1024.txt
