#!/usr/bin/env python3
"""How long does the HOST spend issuing one captured train step (two hipGraph launches + the batch hand-over), against
the GPU time of the step?  If the two are close the step is bound by hipGraphLaunch, not by the kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

B, T, P = 8, 512, 64
eng = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
eng.train_dropout = True
batch = {k: v.cuda() for k, v in synthetic_batch(B, T, P, seed=1).items()}
for _ in range(5):
    eng.train_step_graphed(batch)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
host = 0.0
for _ in range(n):
    h0 = time.perf_counter()
    eng.train_step_graphed(batch)
    host += time.perf_counter() - h0
issue_done = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"host time per step (issue only): {host / n * 1e3:.3f} ms; all {n} steps issued after {issue_done * 1e3:.1f} ms; "
      f"GPU done after {total * 1e3:.1f} ms = {total / n * 1e3:.3f} ms/step")
ent = next(iter(eng._graphs.values()))
for name in ("fb", "opt"):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    for _ in range(20):
        ent[name].replay()
    h = (time.perf_counter() - h0) / 20
    torch.cuda.synchronize()
    print(f"{name}.replay(): host {h * 1e3:.3f} ms per launch (queue kept full)")
