#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5sk; mkdir -p $o
bash tools/probes/ab.sh $o/ab512 3 "SET:kv_dgrad_split=0" "SET:kv_dgrad_split=2" "SET:kv_dgrad_split=4" | tee $o/ab512.txt
EXTRA="--frames 1024 --phonemes 128" bash tools/probes/ab.sh $o/ab1024 2 "SET:kv_dgrad_split=0" "SET:kv_dgrad_split=2" | tee $o/ab1024.txt
