cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof1 -o r01 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > gpurun_out/prof1_bench.json 2> gpurun_out/prof1_err.log
ls -R gpurun_out/prof1 | head -30
python - <<'PY'
import csv, glob
fs = glob.glob('gpurun_out/prof1/**/*kernel_stats.csv', recursive=True)
print(fs)
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    print(list(rows[0].keys()))
    for r in rows[:30]:
        print(r)
PY
