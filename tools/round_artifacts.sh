# usage (on the GPU box): bash tools/round_artifacts.sh r02   — bench lines, rocprofv3 kernel stats, PMC traffic, step timeline
tag=${1:-r02}
out=gpurun_out/$tag; mkdir -p $out
python bench.py --steps 200 --warmup 10 > $out/bench_8x512.json 2> $out/bench_8x512.err
python bench.py --steps 200 --warmup 10 --frames 1024 --phonemes 128 --no-cpu-baseline --no-extra-shapes > $out/bench_8x1024.json 2> $out/bench_8x1024.err
python tools/step_timeline.py > $out/step_timeline_8x512.txt 2>/dev/null
python tools/step_timeline.py 1024 128 > $out/step_timeline_8x1024.txt 2>/dev/null
bash tools/rocprof_bench.sh ${tag}_512 > $out/prof_512.log 2>&1
bash tools/rocprof_bench.sh ${tag}_1024 --frames 1024 --phonemes 128 > $out/prof_1024.log 2>&1
bash tools/rocprof_pmc.sh ${tag}_512 512 64 > $out/pmc_512.log 2>&1
bash tools/rocprof_pmc.sh ${tag}_1024 1024 128 > $out/pmc_1024.log 2>&1
cp gpurun_out/prof_${tag}_* gpurun_out/pmc_${tag}_*hbm_traffic* gpurun_out/pmc_${tag}_*_SIZE.txt gpurun_out/pmc_${tag}_*_MFMA.txt $out/ 2>/dev/null
tail -1 $out/bench_8x512.json | cut -c1-300; tail -1 $out/bench_8x1024.json | cut -c1-300; ls $out
