#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final_tests.txt
