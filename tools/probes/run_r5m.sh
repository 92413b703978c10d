#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5m_pytest.txt 2>&1
tail -5 gpurun_out/r5m_pytest.txt
python bench.py --steps 60 --warmup 10 > gpurun_out/r5m_b512.txt 2>&1; tail -1 gpurun_out/r5m_b512.txt | cut -c1-300
python bench.py --steps 60 --warmup 10 --frames 1024 --phonemes 128 > gpurun_out/r5m_b1024.txt 2>&1; tail -1 gpurun_out/r5m_b1024.txt | cut -c1-300
