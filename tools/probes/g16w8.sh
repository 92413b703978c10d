# A/B of the 8-wave 128x64 tile for single GEMMs, interleaved repeats on one box
for rep in 1 2 3; do for cfg in "0 4096" "3 128" "3 256" "3 64"; do set -- $cfg
echo -n "rep $rep W8=$1 thr12864=$2: "; KK_G16_W8=$1 KK_GEMM16_TUNE=4096,$2,30384 python bench.py --steps 100 --warmup 5 --no-cpu-baseline ${EXTRA} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
