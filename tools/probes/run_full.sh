mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_all.log
tail -8 gpurun_out/t_all.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 3000 gpurun_out/bench.json
