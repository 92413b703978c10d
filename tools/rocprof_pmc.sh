# usage (on the GPU box): bash tools/rocprof_pmc.sh <tag> [frames phonemes]   — HBM traffic counters, one pass per counter
# (guide: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc only with --kernel-trace), then one pass for the matrix-pipe
# occupancy of every kernel (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES, SQ_INSTS_VALU, SQ_INSTS_MFMA); batch 8
tag=${1:-pmc}; T=${2:-512}; P=${3:-64}
rm -f gpurun_out/pmc_${tag}_raw.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE MFMA; do
  ctr=$c; [ $c = MFMA ] && ctr="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
  rocprofv3 --pmc $ctr --kernel-trace -d gpurun_out/pmc_${tag}_$c -o r -- python bench.py --steps 2 --warmup 2 --repeats 1 --no-graph --no-cpu-baseline --no-extra-shapes --frames $T --phonemes $P > gpurun_out/pmc_${tag}_$c.json 2> gpurun_out/pmc_${tag}_$c.err
  db=$(find gpurun_out/pmc_${tag}_$c -name '*.db' | head -1)
  : > gpurun_out/pmc_${tag}_$c.txt
  for one in $ctr; do python tools/rocpd_pmc.py $db $one gpurun_out/pmc_${tag}_raw.json >> gpurun_out/pmc_${tag}_$c.txt; done
  rm -rf gpurun_out/pmc_${tag}_$c
done
head -12 gpurun_out/pmc_${tag}_FETCH_SIZE.txt; head -12 gpurun_out/pmc_${tag}_WRITE_SIZE.txt; head -12 gpurun_out/pmc_${tag}_MFMA.txt
python tools/pmc_summary.py gpurun_out/pmc_${tag}_raw.json 8 $T $P bf16 gpurun_out/pmc_${tag}_hbm_traffic_8x${T}x${P}.json
