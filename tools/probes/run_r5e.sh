#!/bin/bash
mkdir -p gpurun_out
KK_G16X_NS4=0 python tools/probes/g16x_longk_probe.py 8192 > gpurun_out/r5e_ns3.txt 2>&1
KK_G16X_NS4=1 python tools/probes/g16x_longk_probe.py 8192 > gpurun_out/r5e_ns4.txt 2>&1
KK_G16X_NS4=1 python tools/probes/g16x_longk_probe.py 4096 > gpurun_out/r5e_ns4_4096.txt 2>&1
cat gpurun_out/r5e_ns3.txt gpurun_out/r5e_ns4.txt gpurun_out/r5e_ns4_4096.txt | grep -v amdgpu.ids
