# rocprofv3 kernel stats of the replayed step + interleaved A/B of the large-tile family (tuning flavour)
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o step -- python $R/bench.py --steps 15 --warmup 3 --repeats 1 --no-cpu-baseline --no-extra-shapes --no-roofline > $R/gpurun_out/prof_bench.log 2>&1
cd $R
ls gpurun_out/prof | head
DB=$(find gpurun_out/prof -name "*.db" | head -1); echo db=$DB; python tools/rocpd_stats.py $DB > gpurun_out/prof_stats.txt 2>&1; head -45 gpurun_out/prof_stats.txt
for i in 1 2 3; do
  KK_G16X=0 python bench.py --lib tuning --steps 100 --repeats 2 --no-cpu-baseline --no-extra-shapes --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('G16X=0 ', d['ms_per_step'])"
  KK_G16X=15 python bench.py --lib tuning --steps 100 --repeats 2 --no-cpu-baseline --no-extra-shapes --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('G16X=15', d['ms_per_step'])"
done
