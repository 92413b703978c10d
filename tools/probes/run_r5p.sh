#!/bin/bash
mkdir -p gpurun_out
for o in 0 64; do echo "== KK_PROBE_OR=$o"; KK_PROBE_OR=$o timeout 300 python tools/probes/g16x_longk_probe.py 8192 2>&1 | grep "T="; done > gpurun_out/r5p_probe.txt
for o in 0 64; do echo "== KK_PROBE_OR=$o"; KK_PROBE_OR=$o timeout 300 python tools/probes/g16x_longk_probe.py 4096 2>&1 | grep "T="; done >> gpurun_out/r5p_probe.txt
cat gpurun_out/r5p_probe.txt
EXTRA="--frames 1024 --phonemes 128" bash tools/probes/ab.sh gpurun_out/r5p_ab1024 2 "KK_G16X_DBG=0" "KK_G16X_DBG=64" > gpurun_out/r5p_ab1024.txt 2>&1
cat gpurun_out/r5p_ab1024.txt
bash tools/probes/ab.sh gpurun_out/r5p_ab512 2 "KK_G16X_DBG=0" "KK_G16X_DBG=64" > gpurun_out/r5p_ab512.txt 2>&1
cat gpurun_out/r5p_ab512.txt
