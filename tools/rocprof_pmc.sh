# usage (on the GPU box): bash tools/rocprof_pmc.sh <tag> [frames phonemes]   — HBM traffic counters, one pass per counter
# (guide: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc only with --kernel-trace); batch 8
tag=${1:-pmc}; T=${2:-512}; P=${3:-64}
rm -f gpurun_out/pmc_${tag}_raw.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_${tag}_$c -o r -- python bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline --frames $T --phonemes $P > gpurun_out/pmc_${tag}_$c.json 2> gpurun_out/pmc_${tag}_$c.err
  db=$(find gpurun_out/pmc_${tag}_$c -name '*.db' | head -1)
  python tools/rocpd_pmc.py $db $c gpurun_out/pmc_${tag}_raw.json > gpurun_out/pmc_${tag}_$c.txt
  rm -rf gpurun_out/pmc_${tag}_$c
done
head -12 gpurun_out/pmc_${tag}_FETCH_SIZE.txt; head -12 gpurun_out/pmc_${tag}_WRITE_SIZE.txt
python tools/pmc_summary.py gpurun_out/pmc_${tag}_raw.json 8 $T $P bf16 gpurun_out/pmc_${tag}_hbm_traffic_8x${T}x${P}.json
