set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" > gpurun_out/t_gemm.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_gemm.log
tail -15 gpurun_out/t_gemm.log
for on in 0 271 15; do python tools/gemm_shapes_bench.py 4096 $on --lib tuning 2>&1 | grep -v amdgpu.ids > gpurun_out/gsb_4096_$on.log; done
for on in 0 271 15; do python tools/gemm_shapes_bench.py 8192 $on --lib tuning 2>&1 | grep -v amdgpu.ids > gpurun_out/gsb_8192_$on.log; done
