#!/bin/bash
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 KK_DP_FORCE=1 KK_DP_PAYLOAD=bf16
bash tools/rocprof_bench.sh r5i_dp16 > gpurun_out/r5i_prof.log 2>&1
head -40 gpurun_out/prof_r5i_dp16_kernel_stats.txt | cut -c1-160
grep -n "cast_ranges\|nccl\|rccl\|Reduce" gpurun_out/prof_r5i_dp16_kernel_stats.txt | cut -c1-200
