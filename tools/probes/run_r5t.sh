#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5w1; mkdir -p $o
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "memory_tail or zero_fill or graphed_accumulation or bench_config or workspace" > $o/test.log 2>&1; tail -5 $o/test.log
EXTRA="--frames 1024 --phonemes 128" bash tools/probes/ab.sh $o/ab1024 3 "SET:wgrad1_aside=0" "SET:wgrad1_aside=1" | tee $o/ab1024.txt
EXTRA="--workload dyn16384" bash tools/probes/ab.sh $o/abdyn 2 "SET:wgrad1_aside=0" "SET:wgrad1_aside=1" | tee $o/abdyn.txt
EXTRA="--frames 768 --phonemes 96" bash tools/probes/ab.sh $o/ab768 2 "SET:wgrad1_aside=0" "SET:wgrad1_aside=1" | tee $o/ab768.txt
