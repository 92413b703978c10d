#!/bin/bash
mkdir -p gpurun_out
bash tools/probes/ab.sh gpurun_out/r5l_ab512 2 "KK_SIB_RIF=1" "KK_SIB_RIF=2" > gpurun_out/r5l_ab512.txt 2>&1
EXTRA="--frames 1024 --phonemes 128" bash tools/probes/ab.sh gpurun_out/r5l_ab1024 2 "KK_SIB_RIF=1" "KK_SIB_RIF=2" > gpurun_out/r5l_ab1024.txt 2>&1
cat gpurun_out/r5l_ab512.txt gpurun_out/r5l_ab1024.txt
