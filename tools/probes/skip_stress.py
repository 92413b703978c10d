#!/usr/bin/env python3
"""Rare-skip stress: N runs of 600 graphed steps, (a) free-running like bench.py, (b) with a device synchronize between the
optimizer graph of step n and the forward/backward graph of step n+1.  Prints skipped-step counts per run."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
b = {k: v.cuda() for k, v in synthetic_batch(8, 512, 64, seed=1234).items()}
for mode in ("free", "sync") * runs:
    e = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
    e.train_dropout = True
    ev = None
    tail = e.arena.G[e.arena.names[-1]].view(-1)
    dbg_g = torch.zeros(16, device="cuda")
    dbg_gmax = torch.zeros(1, device="cuda")
    dbg_ss = torch.zeros(8, dtype=torch.float64, device="cuda")
    for i in range(600):
        if mode == "fence" and ev is not None:
            ev.synchronize()                       # the previous launch of the forward/backward graph has completed
        e.train_step_graphed(b)
        hit = e.opt_state[6] != 0                      # LAST_SKIP (device-side: no host sync in this loop)
        dbg_g = torch.where(hit, tail[:16], dbg_g)
        dbg_gmax = torch.where(hit, tail.abs().max().view(1), dbg_gmax)
        dbg_ss = torch.where(hit, e.grad_sumsq[-8:], dbg_ss)
        if mode == "sync":
            torch.cuda.synchronize()
        if mode == "fence":
            ev = torch.cuda.Event()
            ev.record()
    st = e.opt_stats()
    if st["skipped"]:
        j = int(st["bad_seg"]) - 1
        print("   g tail at a skipped step:", dbg_g.cpu().tolist()[:8], "max|tail|", float(dbg_gmax), "grad_sumsq[-8:]", dbg_ss.cpu().tolist())
        print(f"   last skipped boundary {int(st['bad_attempt'])}: {int(st['bad_count'])} segments non-finite, first = #{j} {e.arena.names[j]}")
    print(f"{mode}: skipped {int(st['skipped'])} of {int(st['attempt'])}; last losses {[round(x, 3) for x in e.losses.cpu().tolist()]}", flush=True)
    del e
    torch.cuda.empty_cache()
