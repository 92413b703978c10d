# usage (GPU box): bash tools/probes/chain_pmc.sh <T>   — L2 hit / miss / fabric-read counters of the four launches and of the persistent launch
T=${1:-512}
out=gpurun_out/r06_chain; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $pass | tr ' ' '+')
  rocprofv3 --pmc $pass --kernel-trace -d $out/pmc_$tag -o r -- python tools/probes/chain_probe.py $T 0 pmc > $out/pmc_${T}_$tag.log 2>&1
  db=$(find $out/pmc_$tag -name '*.db' | head -1)
  for one in $pass; do python tools/rocpd_pmc.py $db $one | grep -v "^#" | cut -c1-60,100-160 > $out/pmc_${T}${KK_LIB}_$one.txt; cat $out/pmc_${T}${KK_LIB}_$one.txt; done
  rm -rf $out/pmc_$tag
done
