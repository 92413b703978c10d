# usage (GPU box): bash tools/probes/attn_pmc.sh <tag> <fwd|dq|dkv> <S> <causal> <p>   — SQ / LDS counters of one attention kernel
tag=$1; which=$2; S=$3; causal=$4; p=$5
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/attnpmc_${tag}.txt; : > $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); d=gpurun_out/attnpmc_${tag}_$i
  rocprofv3 --pmc $grp --kernel-trace -d $d -o r -- python tools/probes/attn_one.py $which $S $causal $p 10 > $d.log 2>&1 || { echo "pass $i ($grp) failed" >> $out; tail -3 $d.log >> $out; }
  db=$(find $d -name '*.db' | head -1)
  for c in $grp; do [ -n "$db" ] && python tools/rocpd_pmc.py $db $c | grep -i "attn_" | awk -v c=$c '{print c, $0}' | cut -c1-40,120-200 >> $out; done
  rm -rf $d
done
cat $out
