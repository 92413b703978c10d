#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5ln; mkdir -p $o
bash tools/probes/ab.sh $o/ab512 4 "LIBV=lnatom" "LIBV=tuning" | tee $o/ab512.txt
EXTRA="--frames 1024 --phonemes 128" bash tools/probes/ab.sh $o/ab1024 2 "LIBV=lnatom" "LIBV=tuning" | tee $o/ab1024.txt
