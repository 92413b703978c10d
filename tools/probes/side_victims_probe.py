"""Which launches of the decoder backward does the side branch slow down?  One decoder layer's backward (layer 3) gets a time-stamp
launch after every kernel of the main chain; the replayed step is measured as shipped and with the side branch's backward skipped
(timing only).  Prints per-kernel durations side by side.
    PYTHONPATH=. python tools/probes/side_victims_probe.py"""
import os
import sys

os.environ["KK_TRACE"] = "1"
import torch
from kokoro_ruslan_amd import engine as E, lib as kk
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

batch = {k: v.cuda() for k, v in synthetic_batch(8, 512, 64, seed=1).items()}
orig_call = kk.call
LAYER = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def run(skip_side_bwd):
    eng = E.KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
    eng.train_dropout = True
    eng.zero_skip_overwritten = False
    stamps = torch.zeros(256, dtype=torch.int64, device="cuda")
    state = {"bwd": False, "win": False, "names": [], "n": 0}
    orig_mark = eng._mark

    def mark(name):
        if name == f"dec{LAYER + 1} bwd done":
            state["win"], state["n"], state["names"] = True, 0, []
            orig_call("kk_timestamp", stamps[0:])
        elif name == f"dec{LAYER} bwd done":
            state["win"] = False
        orig_mark(name)
    eng._mark = mark

    def call(name, *a):
        if name == "kk_losses_bwd":
            state["bwd"] = True
        elif name == "kk_seg_sumsq":
            state["bwd"] = False
        if skip_side_bwd and state["bwd"] and eng._tmp_ns == "side.":
            return
        r = orig_call(name, *a)
        if state["win"] and eng._tmp_ns == "" and name != "kk_timestamp":
            state["n"] += 1
            state["names"].append(name)
            orig_call("kk_timestamp", stamps[state["n"]:])
        return r
    kk.call = call
    E.kk.call = call
    try:
        for _ in range(12):
            eng.train_step_graphed(batch)
        torch.cuda.synchronize()
        acc = None
        for _ in range(10):
            eng.train_step_graphed(batch)
            torch.cuda.synchronize()
            t = stamps.cpu().tolist()
            d = [(t[i + 1] - t[i]) / 100.0 for i in range(state["n"])]
            acc = d if acc is None else [x + y for x, y in zip(acc, d)]
        return state["names"], [x / 10 for x in acc]
    finally:
        kk.call = orig_call
        E.kk.call = orig_call


n1, with_side = run(False)
n2, without = run(True)
print(f"decoder layer {LAYER} backward, main chain, us per launch (incl. the stamp launch that follows it): with the side branch | without | delta")
for nm, a, b in zip(n1, with_side, without):
    print(f"  {nm:28s} {a:7.1f} {b:7.1f}  {a - b:+6.1f}")
print(f"  {'sum':28s} {sum(with_side):7.1f} {sum(without):7.1f}  {sum(with_side) - sum(without):+6.1f}")
