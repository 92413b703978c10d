#!/bin/bash
# Gap between dependent launches in a replayed hipGraph under HIP runtime environment settings (one box, back to back).
mkdir -p gpurun_out
out=gpurun_out/env_gap.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python tools/probes/graph_gap_probe.py >> $out 2>&1; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run GPU_FLUSH_ON_EXECUTION=1
run ROC_SKIP_KERNEL_ARG_COPY=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run ROC_USE_FGS_KERNARG=0
run DEBUG_HIP_DYNAMIC_QUEUES=0
cat $out
