"""What does the backward's side branch (predictors' + text encoder's backward beside the decoder backward) cost the step?
Three graph-replayed variants of the 8x512 step on one box: as shipped; with every launch of the side branch's BACKWARD skipped (wrong
gradients — timing only: the ceiling of anything that makes that branch cheaper); and with no overlap at all (engine.overlap = False).
    PYTHONPATH=. python tools/probes/side_cost_probe.py [frames phonemes]"""
import sys
import time
import torch
from kokoro_ruslan_amd import engine as E, lib as kk
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
batch = {k: v.cuda() for k, v in synthetic_batch(8, T, P, seed=1).items()}
orig_call = kk.call


def run(tag, skip_side_bwd=False, overlap=True, skip_fwd_side=False, only=None):
    eng = E.KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
    eng.train_dropout = True
    eng.overlap = overlap
    state = {"bwd": False, "side": False, "part": "pred"}
    orig_mark = eng._mark

    def mark(name):                      # the side backward = predictors + heads, then the text encoder
        if name == "side: predictors bwd done":
            state["part"] = "enc"
        elif name == "side: backward start":
            state["part"] = "pred"
        orig_mark(name)
    eng._mark = mark
    import contextlib
    orig_on = eng._on_stream

    @contextlib.contextmanager
    def on_stream(stream, ns, enable=True, after=None):      # (marks the side branch's launches also when overlap is off)
        was, state["side"] = state["side"], state["side"] or ns == "side."
        try:
            with orig_on(stream, ns, enable, after):
                yield
        finally:
            state["side"] = was
    eng._on_stream = on_stream

    def call(name, *a):
        if name == "kk_losses_bwd":
            state["bwd"] = True
        elif name == "kk_seg_sumsq":
            state["bwd"] = False
        if skip_side_bwd and state["bwd"] and state["side"] and (only is None or state["part"] == only):
            return
        if skip_fwd_side and not state["bwd"] and state["side"]:
            return
        return orig_call(name, *a)
    kk.call = call
    E.kk.call = call
    try:
        eng.zero_skip_overwritten = False           # (the overwrite record would not match a step with launches missing)
        for _ in range(8):
            eng.train_step_graphed(batch)
        torch.cuda.synchronize()
        reps = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(100):
                eng.train_step_graphed(batch)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) * 10)
        print(f"{tag:60s} {sorted(reps)[2]:.3f} ms/step  {[round(x, 3) for x in reps]}", flush=True)
    finally:
        kk.call = orig_call
        E.kk.call = orig_call


run("as shipped")
run("side backward: predictors + heads skipped (timing only)", skip_side_bwd=True, only="pred")
run("side backward: text encoder skipped (timing only)", skip_side_bwd=True, only="enc")
run("side branch's backward launches skipped (timing only)", skip_side_bwd=True)
run("side branch skipped in forward AND backward (timing only)", skip_side_bwd=True, skip_fwd_side=True)
run("no overlap (one stream)", overlap=False)
run("as shipped (again)")
# single-stream graphs take the runtime's packet-capture path (1.5 us between dependent launches against 2.5 in a forked graph):
run("no overlap, side backward skipped (timing only)", skip_side_bwd=True, overlap=False)
run("no overlap, side skipped in forward AND backward (timing only)", skip_side_bwd=True, skip_fwd_side=True, overlap=False)
