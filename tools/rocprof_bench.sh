# usage (on the GPU box): bash tools/rocprof_bench.sh <tag> [extra bench.py args]
tag=${1:-run}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o r -- python bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-extra-shapes "$@" > gpurun_out/prof_${tag}_bench.json 2> gpurun_out/prof_${tag}_err.log
db=$(find gpurun_out/prof_$tag -name '*.db' | head -1)
python tools/rocpd_stats.py $db > gpurun_out/prof_${tag}_kernel_stats.txt
python tools/rocpd_gaps.py $db 0.3 0.8 > gpurun_out/prof_${tag}_gaps.txt
python tools/rocpd_timeline.py $db 8 > gpurun_out/prof_${tag}_timeline.txt
cat gpurun_out/prof_${tag}_gaps.txt; tail -1 gpurun_out/prof_${tag}_bench.json | cut -c1-200
rm -rf gpurun_out/prof_$tag
