mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_all.log
tail -8 gpurun_out/t_all.log
timeout 1200 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 1500 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
bash tools/probes/dp_cost_ab.sh > gpurun_out/dp_ab.txt 2>&1; cat gpurun_out/dp_ab.txt
