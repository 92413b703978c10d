for r in 1 2 3; do
  for w in "r5 _r5" "r6 ."; do
    set -- $w
    (cd $2 && python bench.py --steps 200 --repeats 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], [x['ms_per_step'] for x in d['extra_shapes']])")
  done
done
