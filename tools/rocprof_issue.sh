# usage (on the GPU box): bash tools/rocprof_issue.sh <tag> [frames phonemes]   — what each kernel of the (eager) step spends its waves' lifetime on:
# SQ_WAVES / SQ_WAVE_CYCLES and the cycles in which the vector ALU, the matrix core, the scalar unit and the LDS execute an instruction, one counter
# group per pass (tools/issue_table.py turns the raw sums into a table: a kernel whose shares add up to ~100 % is bound by instruction issue).
tag=${1:-issue}; T=${2:-512}; P=${3:-64}
out=gpurun_out/issue_${tag}; mkdir -p $out
rm -f $out/raw.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for ctr in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace -d $out/p$i -o r -- python bench.py --steps 2 --warmup 2 --repeats 1 --no-graph --no-cpu-baseline --no-extra-shapes --no-roofline --frames $T --phonemes $P > $out/p$i.json 2> $out/p$i.err
  db=$(find $out/p$i -name '*.db' | head -1)
  for one in $ctr; do python tools/rocpd_pmc.py $db $one $out/raw.json > /dev/null; done
  rm -rf $out/p$i
done
python tools/issue_table.py $out/raw.json > $out/issue_8x${T}.txt; head -50 $out/issue_8x${T}.txt
