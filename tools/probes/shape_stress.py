import os, sys, torch
sys.path.insert(0, '/root/repo')
from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch
for (T, P, graph, steps) in [(1024, 128, True, 300), (512, 64, False, 300), (437, 53, True, 300)]:
    e = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
    e.train_dropout = True
    b = {k: v.cuda() for k, v in synthetic_batch(8, T, P, seed=7).items()}
    for _ in range(steps):
        (e.train_step_graphed if graph else e.train_step)(b)
    st = e.opt_stats()
    print(f"T={T} P={P} graph={graph}: skipped {int(st['skipped'])} of {int(st['attempt'])}, losses {[round(x,3) for x in e.losses.cpu().tolist()]}, finite params {bool(torch.isfinite(e.arena.p).all())}", flush=True)
    del e; torch.cuda.empty_cache()
