import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch
b = {k: v.cuda() for k, v in synthetic_batch(8, 512, 64, seed=1234).items()}
def run(graph, n=4, pre=False):
    e = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
    if pre:
        e.zero_grad(); e.forward_backward(b)
    e.train_dropout = True
    out = []
    for _ in range(n):
        (e.train_step_graphed if graph else e.train_step)(b)
        out.append(e.losses.clone())
    torch.cuda.synchronize()
    return torch.stack(out).cpu(), e.arena.p.clone(), e.encoder_stack_error()
runs = {}
for name, graph, pre in [("eager_a", False, True), ("graph_a", True, False), ("eager_b", False, False), ("graph_b", True, False), ("eager_c", False, True), ("graph_c", True, False)]:
    runs[name] = run(graph, pre=pre)
ref = runs["eager_a"]
for name, (l, p, err) in runs.items():
    dl = (l - ref[0]).abs().max(dim=1).values.tolist()
    print(f"{name}: per-step max |loss - eager_a| {[f'{x:.2e}' for x in dl]}  max|p - eager_a.p| {float((p - ref[1]).abs().max()):.3e}  err {err}   dur {l[:, 2].tolist()}")
