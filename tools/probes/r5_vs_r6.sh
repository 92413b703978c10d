# Round 6 against round 5 on ONE box.  In the container:  git worktree add _r5 28c0d2e && (cd _r5 && python -m kokoro_ruslan_amd.build) ; then
#   gpurun -- 'rm -rf _r5/.git; bash tools/probes/r5_vs_r6.sh'      (the worktree travels with the snapshot; remove it afterwards: rm -rf _r5; git worktree prune)
# Each side runs ITS OWN bench.py, engine and product library; three interleaved rounds.  Result: profiles/r06_vs_r05_one_box.txt
for r in 1 2 3; do
  for w in "r5 _r5" "r6 ."; do
    set -- $w
    (cd $2 && python bench.py --steps 200 --repeats 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], [x['ms_per_step'] for x in d['extra_shapes']])")
  done
done
