#!/bin/bash
mkdir -p gpurun_out
{
for f in 1 0; do echo "== KK_ATTN_FWD3=$f"; KK_ATTN_FWD3=$f timeout 600 python tools/probes/hn_epi_repeat.py 1500; done
} > gpurun_out/hn_repeat.txt 2>&1
