# usage (GPU box): bash tools/probes/attn_pmc2.sh <tag> <fwd|dq|dkv> <S> <causal> <p>   — cache / memory-path counters of one attention kernel
tag=$1; which=$2; S=$3; causal=$4; p=$5
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/attnpmc2_${tag}.txt; : > $out
i=0
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_BUFFER_LOAD_WAVEFRONTS_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=gpurun_out/attnpmc2_${tag}_$i
  rocprofv3 --pmc $grp --kernel-trace -d $d -o r -- python tools/probes/attn_one.py $which $S $causal $p 10 > $d.log 2>&1 || { echo "pass $i ($grp) failed" >> $out; tail -3 $d.log >> $out; }
  db=$(find $d -name '*.db' | head -1)
  for c in $grp; do [ -n "$db" ] && python tools/rocpd_pmc.py $db $c | grep -i "attn_f\|attn_b" | awk -v c=$c '{print c, $2, $3, $4}' >> $out; done
  rm -rf $d $d.log
done
cat $out
