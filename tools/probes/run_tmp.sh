timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "two_pass or pair_launch" 2>&1 | tail -4 > gpurun_out/t_bwd2.log
