#!/bin/bash
# usage (in the container, after `gpurun ... bash tools/probes/run_round.sh <tag>` merged gpurun_out/<tag>/): copy the judged files
# into profiles/ under the round's names
tag=${1:-r04}; src=gpurun_out/$tag
cp $src/bench_8x512.json profiles/${tag}_bench_8x512_bf16.json
cp $src/bench_8x1024.json profiles/${tag}_bench_8x1024_bf16.json
for s in 512 1024; do
  cp $src/prof_${tag}_${s}_kernel_stats.txt profiles/${tag}_rocprofv3_kernel_stats_8x${s}_bf16.txt
  cp $src/pmc_${tag}_${s}_FETCH_SIZE.txt profiles/${tag}_pmc_FETCH_SIZE_8x${s}.txt
  cp $src/pmc_${tag}_${s}_WRITE_SIZE.txt profiles/${tag}_pmc_WRITE_SIZE_8x${s}.txt
  cp $src/pmc_${tag}_${s}_MFMA.txt profiles/${tag}_pmc_MFMA_busy_8x${s}.txt
  cp $src/step_timeline_8x${s}.txt profiles/${tag}_step_timeline_timestamps_8x${s}.txt
done
cp $src/pmc_${tag}_512_hbm_traffic_8x512x64.json $src/pmc_${tag}_1024_hbm_traffic_8x1024x128.json profiles/ 2>/dev/null
mv profiles/pmc_${tag}_512_hbm_traffic_8x512x64.json profiles/${tag}_pmc_hbm_traffic_8x512x64.json
mv profiles/pmc_${tag}_1024_hbm_traffic_8x1024x128.json profiles/${tag}_pmc_hbm_traffic_8x1024x128.json
cp $src/prof_${tag}_512_timeline.txt profiles/${tag}_rocprofv3_step_timeline_8x512.txt
cp $src/prof_${tag}_512_gaps.txt profiles/${tag}_rocprofv3_timeline_gaps_8x512.txt
ls profiles | grep ${tag}_
