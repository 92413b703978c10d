# rocprofv3 kernel stats of the replayed step at both bench shapes (no A/B)
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sh in "512 64" "1024 128"; do set -- $sh
rm -rf $R/gpurun_out/prof/*
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o step -- python $R/bench.py --frames $1 --phonemes $2 --steps 15 --warmup 3 --repeats 1 --no-cpu-baseline --no-extra-shapes --no-roofline > $R/gpurun_out/prof_bench_$1.log 2>&1
DB=$(find $R/gpurun_out/prof -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/prof_stats_$1.txt 2>&1
done
rm -rf $R/gpurun_out/prof
