#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_encstack_gpu.py -x -q > gpurun_out/r5v_pytest.txt 2>&1
grep -n "passed\|failed\|Error" gpurun_out/r5v_pytest.txt | tail -3
for i in 1 2; do python tools/step_timeline.py 2>/dev/null | grep "enc5 fwd\|optimizer done"; done
python tools/step_timeline.py 1024 128 2>/dev/null | grep "enc5 fwd\|optimizer done"
WGS=0 python tools/probes/enc_stack_phases.py 2>/dev/null | tail -14
