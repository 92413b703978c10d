#!/usr/bin/env python3
"""The real timeline of one captured train step: KK_TRACE=1 puts one-thread time-stamp kernels at the marks of the step
(encoder / decoder layers, branches, optimizer), also inside the hipGraphs; after a few replays the stamps of the last
step are read back.  Unlike rocprofv3 (which serialises the graph's branches) this shows what overlaps with what.

    python tools/step_timeline.py [frames=512] [phonemes=64]
"""
import os
import sys

os.environ["KK_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
eng = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
eng.train_dropout = True
for kv in os.environ.get("KK_TIMELINE_SET", "").split():      # tools: attr=int pairs for A/B timelines
    setattr(eng, kv.split("=")[0], int(kv.split("=")[1]))
batch = {k: v.cuda() for k, v in synthetic_batch(8, T, P, seed=1).items()}
for _ in range(12):
    eng.train_step_graphed(batch)
torch.cuda.synchronize()
rows = eng.timeline()
prev = {}
print(f"# 8x{T} frames x {P} phonemes, bf16, one replayed step; us since the step's first mark (+ since the previous mark of the same branch)")
for t, name in rows:
    br = name.split(":")[0] if ":" in name else "main"
    print(f"{t:9.1f}  (+{t - prev.get(br, 0.0):7.1f})  {name}")
    prev[br] = t
