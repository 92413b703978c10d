#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5wg; mkdir -p $o
EXTRA="--workload dyn16384" bash tools/probes/ab.sh $o/abdyn 2 "SET:wgrad0_aside=0" "SET:wgrad0_aside=-1" | tee $o/abdyn.txt
EXTRA="--frames 768 --phonemes 96" bash tools/probes/ab.sh $o/ab768 2 "SET:wgrad0_aside=0" "SET:wgrad0_aside=-1" | tee $o/ab768.txt
EXTRA="--batch 16 --frames 512 --phonemes 64" bash tools/probes/ab.sh $o/ab16x512 2 "SET:wgrad0_aside=0" "SET:wgrad0_aside=-1" | tee $o/ab16x512.txt
