#!/usr/bin/env python3
"""Do consecutive hipGraph launches on one stream really run in order when the second graph has parallel branches?
KK_TRACE time stamps: "optimizer done" (last node of the single-stream optimizer graph of step n) against the first
stamps of the forward/backward graph of step n+1 (main chain and the kv branch, which zeroes the gradient arena)."""
import os
import sys

os.environ["KK_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

e = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
e.train_dropout = True
b = {k: v.cuda() for k, v in synthetic_batch(8, 512, 64, seed=1234).items()}
for _ in range(4):
    e.train_step_graphed(b)
ent = next(iter(e._graphs.values()))
viol = 0
for trial in range(200):
    for _ in range(3):                       # keep the queue full: graphs launched back to back
        ent["fb"].replay()
        ent["opt"].replay()
    ent["fb"].replay()                        # step n+1's forward/backward only: its stamps overwrite, "optimizer done" stays
    torch.cuda.synchronize()
    t = e._mark_buf.cpu().tolist()
    m = e._marks
    opt_done, start, zg = t[m["optimizer done"]], t[m["step.start"]], t[m["kv: zero_grad done"]]
    if start < opt_done or zg < opt_done:
        viol += 1
        if viol <= 5:
            print(f"trial {trial}: step.start {(start - opt_done) / 100:.2f} us, kv zero_grad done {(zg - opt_done) / 100:.2f} us relative to 'optimizer done' of the previous step")
    ent["opt"].replay()
print("order violations:", viol, "of 200")
