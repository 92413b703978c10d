#!/usr/bin/env python3
"""Per-kernel issue shares from the raw sums of tools/rocprof_issue.sh:   python tools/issue_table.py raw.json
Cycle counters are in quads of 4 clocks except SQ_VALU_MFMA_BUSY_CYCLES (clocks); the counters sample 1/32 of the chip's SIMDs' waves alike, so
ratios to SQ_WAVE_CYCLES need no scaling: share = active cycles / (wave cycles / waves-per-SIMD-slot) is reported per SIMD as
active / (SQ_BUSY_CYCLES-equivalent) — here simply active cycles per SIMD over the mean wave lifetime times the waves resident per SIMD."""
import json, re, sys

raw = json.load(open(sys.argv[1]))
rows = []
for name, c in raw.items():
    if "SQ_WAVES" not in c or "SQ_WAVE_CYCLES" not in c:
        continue
    n, waves = c["SQ_WAVES"]
    wc = c["SQ_WAVE_CYCLES"][1]
    if waves <= 0 or wc <= 0:
        continue
    g = lambda k: c.get(k, [0, 0.0])[1]
    busy = g("SQ_BUSY_CYCLES")                 # quads during which the sampled SQs had any wave, summed over dispatches
    # SIMD-time available to the sampled slice: busy quads x 4 SIMDs per CU x CUs per SQ instance is not exposed; use wave-quads instead:
    # share of a wave's life in which IT (or its SIMD neighbours, for the pipes) executes category X
    life = wc
    short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)
    short = re.sub(r"\.kd$", "", short)[:58]
    rows.append((life, short, n, waves / n, wc / waves,
                 100 * g("SQ_ACTIVE_INST_VALU") / life, 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / 4 / life, 100 * g("SQ_ACTIVE_INST_SCA") / life,
                 100 * g("SQ_ACTIVE_INST_LDS") / life, 100 * g("SQ_ACTIVE_INST_MISC") / life, 100 * g("SQ_ACTIVE_INST_ANY") / life,
                 g("SQ_INSTS_VALU") / waves, g("SQ_INSTS_SALU") / waves, g("SQ_INSTS_MFMA") / waves))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("# share of the waves' summed lifetime (SQ_WAVE_CYCLES) in which an instruction of the category is executing; with w waves per SIMD the pipes can")
print("# be busy at most 100 % / w of that each, so  w x (VALU + MFMA + scalar + LDS + misc) ~ 100 %  means the kernel is bound by instruction issue.")
print(f"{'kernel':58s} {'disp':>5s} {'waves':>7s} {'life q':>8s} {'wave-time %':>11s} | {'VALU':>5s} {'MFMA':>5s} {'scal':>5s} {'LDS':>5s} {'misc':>5s} {'any':>5s} | per wave: {'VALU':>6s} {'SALU':>6s} {'MFMA':>6s}")
for r in rows[:60]:
    print(f"{r[1]:58s} {r[2]:5d} {r[3]:7.0f} {r[4]:8.0f} {100 * r[0] / tot:11.1f} | {r[5]:5.1f} {r[6]:5.1f} {r[7]:5.1f} {r[8]:5.1f} {r[9]:5.1f} {r[10]:5.1f} |           {r[11]:6.0f} {r[12]:6.0f} {r[13]:6.0f}")
