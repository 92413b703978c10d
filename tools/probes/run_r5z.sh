#!/bin/bash
mkdir -p gpurun_out
for d in 0 1024; do echo "== KK_ATTN_DBG=$d"; KK_ATTN_DBG=$d timeout 300 python tools/probes/attn_keep_ab.py 2>&1 | grep "B="; done > gpurun_out/r5z_hash4.txt
cut -c1-75 gpurun_out/r5z_hash4.txt
