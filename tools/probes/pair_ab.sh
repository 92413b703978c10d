#!/bin/bash
# A/B of the attention-backward pair launch (KK_ATTN_BWD_PAIR) on one box: new kernel tests, then interleaved bench runs.
out=gpurun_out/pair; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "dgrad_delta or pair_launch or attention" > $out/tests.log 2>&1; tail -5 $out/tests.log
for i in 1 2; do
  for v in 0 1; do
    KK_ATTN_BWD_PAIR=$v timeout 600 python bench.py --no-cpu-baseline --steps 200 > $out/bench_pair${v}_$i.json 2> $out/bench_pair${v}_$i.err
    python - <<PY
import json
try:
    d = json.load(open("$out/bench_pair${v}_$i.json"))
    print("pair=$v run $i:", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["skipped"], d["final_losses"][:2])
except Exception as e:
    print("pair=$v run $i failed", e); print(open("$out/bench_pair${v}_$i.err").read()[-1500:])
PY
  done
done
KK_ATTN_BWD_PAIR=1 timeout 600 python bench.py --no-cpu-baseline --steps 100 --frames 1024 --phonemes 128 > $out/bench1024_pair1.json 2>$out/b1024_1.err
KK_ATTN_BWD_PAIR=0 timeout 600 python bench.py --no-cpu-baseline --steps 100 --frames 1024 --phonemes 128 > $out/bench1024_pair0.json 2>$out/b1024_0.err
python - <<PY
import json
for v in (0, 1):
    try:
        d = json.load(open(f"$out/bench1024_pair{v}.json")); print("8x1024 pair", v, d["value"], d["ms_per_step"])
    except Exception as e: print("1024 failed", v, e)
PY
