"""Does the fused encoder launch ever read stale data?  Between launches the encoder weights and the batch are changed by
amounts far above bf16 noise; the fused result is compared with the per-kernel sequence on the same state.  Eager and
graph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

B, T, P = 8, 512, 64
eng = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
eng.train_dropout = True
batches = [{k: v.cuda() for k, v in synthetic_batch(B, T, P, seed=s).items()} for s in range(4)]
static = {k: v.clone() for k, v in batches[0].items()}
Ne, H = B * P, eng.dims.hidden
gen = torch.Generator(device="cuda").manual_seed(1)
enc_names = [n for n in eng.arena.param_names if n.startswith("transformer_encoder_layers") and n.endswith("weight") and eng.arena.P[n].dim() == 2]

def perturb(i):
    for n in enc_names:
        w = eng.arena.P[n]
        w.mul_(1.0 + 0.05 * torch.randn(w.shape, device="cuda", generator=gen))
    eng.sync_shadow()
    for k, v in batches[i % 4].items():
        static[k].copy_(v)

def run(fused, graph=None):
    eng.enc_fused = fused
    eng.rng.fill_(55)
    if graph is None:
        eng.forward_backward(static, backward=False)
    else:
        graph.replay()
    return eng._buf("enc.norm.y", Ne, H).clone()

for mode in ("eager", "graph"):
    g = None
    if mode == "graph":
        eng.enc_fused = True
        eng.rng.fill_(55)
        eng.forward_backward(static, backward=False)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            eng.forward_backward(static, backward=False)
    worst = 0.0
    for i in range(40):
        perturb(i)
        a = run(True, g)
        b = run(False)
        err = float((a - b).norm() / b.norm())
        worst = max(worst, err)
        if err > 2e-2:
            print(f"  {mode} iteration {i}: rel. error {err:.3e}  <-- stale?")
    print(f"{mode}: worst fused-vs-unfused rel. error over 40 perturbed launches {worst:.3e}; barrier error flag {eng.encoder_stack_error()}")
