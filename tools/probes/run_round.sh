#!/bin/bash
# full GPU suite (no -x) + the round's artifact set: bash tools/probes/run_round.sh <tag>; then tools/collect_profiles.sh <tag> in the container
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/t_all.log
bash tools/round_artifacts.sh ${1:-r04} > gpurun_out/round_artifacts.log 2>&1
