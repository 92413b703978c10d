# usage (GPU box): bash tools/probes/attn_pmc.sh <S> <causal>  — where a SIMD's cycles go in the attention launches (stand-alone, 8 x 8 heads):
# busy cycles, cycles the vector ALU / the matrix core / the LDS are executing, cycles waves wait.  One counter group per pass.
S=${1:-512}; C=${2:-0}
out=gpurun_out/r06_attn_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for pass in "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '+')
  KK_ATTN_DBG=0 rocprofv3 --pmc $pass --kernel-trace -d $out/pmc_$tag -o r -- python tools/probes/attn_grid_timeline.py $S $C > $out/log_${S}_${C}_$tag.log 2>&1
  db=$(find $out/pmc_$tag -name '*.db' | head -1)
  for one in $pass; do python tools/rocpd_pmc.py $db $one | grep -v "^#" | grep "attn_" | cut -c1-70,100-150 > $out/pmc_${S}_${C}_$one.txt; echo "== $one"; cat $out/pmc_${S}_${C}_$one.txt; done
  rm -rf $out/pmc_$tag
done
