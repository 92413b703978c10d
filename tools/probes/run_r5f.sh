#!/bin/bash
mkdir -p gpurun_out
for pf in 0 2 3 4 6 8; do
  echo "=== KK_G16X_PF=$pf cooperative"
  KK_G16X_PF=$pf KK_G16X_PF_COOP=1 python tools/probes/g16x_longk_probe.py 8192 2>&1 | grep -v "amdgpu.ids\|NS4"
done > gpurun_out/r5f_pfcoop.txt
cat gpurun_out/r5f_pfcoop.txt
