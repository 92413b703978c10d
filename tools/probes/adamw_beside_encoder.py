#!/usr/bin/env python3
"""What would the previous step's AdamW pass cost the encoder forward if it ran beside it (the cross-step
overlap that was not built)?  A replayed train step (KK_TRACE time stamps) with and without an AdamW + EMA pass over a second
set of arenas of the same size launched on another stream right before the replay.  frac = share of the arena the pass covers."""
import os, sys
os.environ["KK_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd import lib as kk
from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

eng = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
eng.train_dropout = True
batch = {k: v.cuda() for k, v in synthetic_batch(8, 512, 64, seed=1).items()}
for _ in range(6):
    eng.train_step_graphed(batch)
torch.cuda.synchronize()
a, hp = eng.arena, eng.hp
dummy = [torch.zeros_like(a.p) for _ in range(5)]
d16 = torch.zeros_like(a.p16)
psq = torch.zeros_like(eng.p_sumsq)
side = torch.cuda.Stream()
marks = ("enc5 fwd done", "memory ready", "cross K/V fwd done", "dec0 fwd done", "dec5 fwd done", "optimizer done")
for frac in (0.0, 0.55, 1.0):
    nb = int(a.nblocks * frac)
    res = []
    for rep in range(5):
        torch.cuda.synchronize()
        if nb:
            with torch.cuda.stream(side):
                kk.call("kk_adamw_ema", dummy[0], dummy[1], dummy[2], dummy[3], dummy[4], a.block_seg, nb, eng.seg_gscale, eng.seg_decay,
                        eng.seg_stepsize, a.seg_flags, eng.step_consts, hp.adam_betas[0], hp.adam_betas[1], hp.ema_decay, psq, a.nseg, d16, 0)
        eng.train_step_graphed(batch)
        torch.cuda.synchronize()
        t = dict((n, v) for v, n in eng.timeline())
        res.append([t[m] for m in marks])
    med = [sorted(r[i] for r in res)[len(res) // 2] for i in range(len(marks))]
    print(f"AdamW over {frac:4.2f} of the arena beside the step start:", ", ".join(f"{m} {v:7.1f}" for m, v in zip(marks, med)), flush=True)
