#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5t; mkdir -p $o
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "memory_tail" > $o/test.log 2>&1; tail -3 $o/test.log
EXTRA="--workload dyn16384" bash tools/probes/ab.sh $o/abdyn 2 "SET:tail_aside=0" "SET:tail_aside=3" "SET:tail_aside=4" | tee $o/abdyn.txt
EXTRA="--frames 768 --phonemes 96" bash tools/probes/ab.sh $o/ab768 2 "SET:tail_aside=0" "SET:tail_aside=3" "SET:tail_aside=4" | tee $o/ab768.txt
EXTRA="--batch 16 --frames 512 --phonemes 64" bash tools/probes/ab.sh $o/ab16x512 2 "SET:tail_aside=0" "SET:tail_aside=3" "SET:tail_aside=4" | tee $o/ab16x512.txt
