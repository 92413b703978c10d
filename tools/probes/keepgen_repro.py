"""Steps with the keep bits hashed in the forward / generated beside the encoder / generated a step ahead beside the optimizer must be the
same steps: same masks, so the same losses up to the run-to-run noise of the atomic scatter-adds (compare the two hashing runs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import kokoro_oracle as O
from kokoro_ruslan_amd import engine as eng_mod
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch
d = O.ModelDims(); P = O.init_params(d, 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
b = {k: v.cuda() for k, v in synthetic_batch(8, T, T // 8, seed=1234).items()}
def mk(gen, pre):
    e = eng_mod.KokoroEngine(ModelDims(**d.__dict__), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", init=False, total_steps=20000)
    e.load_params(P); e.train_dropout = True; e.attn_keep_gen = gen; e.keep_pregen = pre
    return e
out = {}
for name, gen, pre, graphed in (("hash/eager", 0, 0, 0), ("hash/eager again", 0, 0, 0), ("gen-in-step/eager", 1, 0, 0), ("pregen/eager", 0, 1, 0),
                                ("hash/graph", 0, 0, 1), ("pregen/graph", 1, 1, 1)):
    e = mk(gen, pre)
    ls = []
    routes = set()
    from kokoro_ruslan_amd import lib as kk
    real = kk.call
    def rec(n, *a):
        real(n, *a)
        if n.startswith("kk_attn_fwd") or n == "kk_attn_keep_gen":
            routes.add(n)
    kk.call = rec
    for _ in range(6):
        (e.train_step_graphed if graphed else e.train_step)(b)
        ls.append(e.losses.clone())
    kk.call = real
    torch.cuda.synchronize()
    out[name] = torch.stack(ls)
    print(f"{name:20s} total loss per step {[round(float(x[0]), 5) for x in ls]}  calls {sorted(routes)}  skipped {e.opt_stats()['skipped']}", flush=True)
ref = out["hash/eager"]
for k, v in out.items():
    print(f"{k:20s} max rel deviation from hash/eager over 6 steps: {float(((v - ref).abs() / ref.abs()).max()):.2e}")
