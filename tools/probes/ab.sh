#!/bin/bash
# Interleaved A/B of environment switches of the TUNING library on one box:  bash tools/probes/ab.sh <out-dir> <reps> "<ENV=a ...>" "<ENV=b ...>" ...
# SET:attr=value in a variant string sets an engine attribute (bench.py --set).
# LIBV=<name> in a variant string selects another library flavour (a --variant build).
# NEVER compare across flavours: the tools flavour (tuning, and every --variant build unless KK_VARIANT_PRODUCT=1 was set when it was built) is ~1.1 % (8 x 512) /
# 0.7 % (8 x 1024) slower in the step than the product flavour of the same source (profiles/r06_attn_hn_core_ab.txt).  A compile-time change is measured either
# as two tools-flavour variants (LIBV=tuning against LIBV=<variant>) or as two product-flavour ones (LIBV=product against a KK_VARIANT_PRODUCT=1 variant).
# Every variant runs bench.py (--lib tuning, headline region only) <reps> times, round-robin, so box and clock drift hit all alike.
out=$1; reps=$2; shift 2
mkdir -p $out
for r in $(seq 1 $reps); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    lib=$(echo "$v" | tr ' ' '\n' | grep '^LIBV=' | cut -d= -f2); lib=${lib:-tuning}
    sets=$(echo "$v" | tr ' ' '\n' | grep '^SET:' | sed 's/^SET:/--set /' | tr '\n' ' ')      # SET:attr=value -> bench.py --set attr=value
    env $(echo "$v" | tr ' ' '\n' | grep -v '^SET:' | tr '\n' ' ') timeout 300 python bench.py --lib $lib $sets --steps 200 --repeats 3 --no-cpu-baseline --no-extra-shapes --no-roofline ${EXTRA} > $out/v${i}_r$r.json 2> $out/v${i}_r$r.err
    python - "$v" $out/v${i}_r$r.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(f"{sys.argv[1]:40s} {d['ms_per_step']:.3f} ms  {d['timed_regions']['ms_per_step']}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
done
