#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r5soak; mkdir -p $o
timeout 600 python tools/probes/dyn_stress.py 60 bf16 > $o/dyn_stress.log 2>&1; echo "dyn_stress rc=$?"; tail -3 $o/dyn_stress.log
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_engine_gpu.py -q -m gpu -p no:cacheprovider -k "memory_tail or graphed_accumulation or bench_config or zero_fill or rccl or dynamic" 2>&1 | grep -E " passed| failed|^FAILED"; done | tee $o/soak.txt
timeout 900 python -m pytest tests/test_trainer_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -E " passed| failed|^FAILED" | tee -a $o/soak.txt
