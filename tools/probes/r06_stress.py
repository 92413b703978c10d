"""Round-6 shape stress: the features added this round (weight warming, keep bits from their own launch, tile-record gradient norms) over ragged shapes —
default dims in the dynamic-batching range (eager and graph replay), and a tiny model at grids of a few workgroups.  Every step is synchronised and its
losses checked finite, so a hang or a fault is attributable to a printed shape.     timeout 600 python tools/probes/r06_stress.py [n]"""
import sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.enable()
faulthandler.dump_traceback_later(1500, exit=True)       # (whole-run watchdog: ~10 s per new shape at default dims)
import numpy as np
import torch
from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def check(e, tag):
    torch.cuda.synchronize()
    ls = [float(x) for x in e.losses.tolist()] if hasattr(e.losses, "tolist") else list(e.losses)
    assert all(np.isfinite(ls)), (tag, ls)


rs = np.random.RandomState(6)
e = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16")
e.train_dropout = True
shapes = []
for i in range(n):
    T = int(rs.randint(90, 1500))
    B = int(min(32, max(1, 16384 // T) if i % 5 else rs.randint(1, 4)))
    shapes.append((B, T, int(rs.randint(12, 60))))
for i, (B, T, Pn) in enumerate(shapes):
    b = {k: v.cuda() for k, v in synthetic_batch(B, T, Pn, seed=i, ragged=True).items()}
    print(f"default dims, eager {i}: B {B} T {T} P {Pn}", flush=True)
    e.train_step(b)
    check(e, (B, T, Pn))
for rep in range(3):
    for i, (B, T, Pn) in enumerate(shapes[:8]):
        b = {k: v.cuda() for k, v in synthetic_batch(B, T, Pn, seed=100 + i, ragged=True).items()}
        print(f"default dims, graphed pass {rep} shape {i}: B {B} T {T} P {Pn}", flush=True)
        e.train_step_graphed(b)
        check(e, (B, T, Pn))
print("default dims ok", e.opt_stats(), flush=True)
del e
torch.cuda.empty_cache()
tiny = ModelDims(vocab=40, mel=20, hidden=128, heads=2, enc_layers=2, dec_layers=2, enc_ff=96, dec_ff=96, var_filter=64, var_kernel=3, var_bins=16, max_len=512)
for G in (1, 2):
    e = KokoroEngine(tiny, StepHyper(gradient_accumulation_steps=G), math_mode="bf16", total_steps=20000)
    e.train_dropout = True
    for i in range(n):
        B, T, Pn = int(rs.randint(1, 5)), int(rs.randint(17, 260)), int(rs.randint(3, 25))
        b = {k: v.cuda() for k, v in synthetic_batch(B, T, Pn, vocab=tiny.vocab, mel=tiny.mel, seed=300 + i, ragged=True).items()}
        print(f"tiny dims G={G} {i}: B {B} T {T} P {Pn}", flush=True)
        (e.train_step_graphed if i % 2 else e.train_step)(b)
        check(e, (B, T, Pn))
    print("tiny dims ok", G, e.opt_stats(), flush=True)
