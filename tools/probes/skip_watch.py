#!/usr/bin/env python3
"""Hunt for rare optimizer skips: replay the captured forward/backward graph, inspect the gradient arena BEFORE the
optimizer graph runs, then replay the optimizer graph.  Prints the first anomalies (non-finite gradients by tensor)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 620
e = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode=mode, total_steps=20000, seed=0)
e.train_dropout = True
b = {k: v.cuda() for k, v in synthetic_batch(8, 512, 64, seed=1234).items()}
for _ in range(3):
    e.train_step_graphed(b)
ent = next(iter(e._graphs.values()))
shown = 0
variant = sys.argv[3] if len(sys.argv) > 3 else "a"
for i in range(steps):
    ent["fb"].replay()
    if variant == "a":                                   # big temporaries between the graphs (raises the incidence)
        bad = ~torch.isfinite(e.arena.g)
        nbad = int(bad.sum())
    ent["opt"].replay()
    if float(e.opt_state[6]) != 0.0 and shown < 8:       # LAST_SKIP
        shown += 1
        gs = e.grad_sumsq
        idx = (~torch.isfinite(gs)).nonzero().flatten().tolist()
        print(f"step {i}: SKIP; {len(idx)} segments with non-finite sumsq; g finite: {bool(torch.isfinite(e.arena.g).all())}; "
              f"losses {[round(x, 3) for x in e.losses.cpu().tolist()]}")
        for j in idx[:6]:
            gg = e.arena.G[e.arena.names[j]]
            print(f"    {e.arena.names[j]} {tuple(gg.shape)}: sumsq {float(gs[j])}, finite {bool(torch.isfinite(gg).all())}, max|.| {float(gg.abs().max()):.3e}, "
                  f"#|.|>1e10 {int((gg.abs() > 1e10).sum())}")
torch.cuda.synchronize()
st = e.opt_stats()
print(mode, "done:", st["skipped"], "skipped of", st["attempt"])
