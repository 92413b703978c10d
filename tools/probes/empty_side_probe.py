#!/usr/bin/env python3
"""Is the main chain's +7 % beside the side branch (decoder layer forward 144 us against 133 us with engine.overlap = 0, also in layers 3-5
where nothing runs beside it) the FORK in the graph or the side branch's WORK?  The captured step with every launch of the side stream
skipped (results are garbage; only the time stamps matter): the graph keeps its fork / join structure, the branch is empty."""
import os, sys
os.environ["KK_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd import lib as kk
from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

mode = sys.argv[1] if len(sys.argv) > 1 else "empty"
eng = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
eng.train_dropout = True
eng.attn_keep_gen = False
if mode.startswith("nooverlap"):
    eng.overlap = False
batch = {k: v.cuda() for k, v in synthetic_batch(8, 512, 64, seed=1).items()}
eng.train_step_graphed(batch)          # eager pass: workspaces, tables
real = kk.call
if mode in ("empty", "emptykv"):
    def call(name, *a):
        st = torch.cuda.current_stream()
        if name != "kk_timestamp" and (st == eng._side or (mode == "emptykv" and st == eng._kv)):
            return
        real(name, *a)
    kk.call = call
ctx = torch.cuda.stream(torch.cuda.Stream()) if mode.endswith("_s") else torch.cuda.stream(torch.cuda.current_stream())
with ctx:                                   # (mode *_s: the whole step on a CREATED stream instead of the default one)
    for _ in range(10):
        eng.train_step_graphed(batch)
torch.cuda.synchronize()
kk.call = real
prev = {}
for t, name in eng.timeline():
    br = name.split(":")[0] if ":" in name else "main"
    if br == "main" and ("fwd done" in name or "bwd done" in name or "optimizer" in name or "losses" in name):
        print(f"{mode:10s} {t:9.1f}  (+{t - prev.get(br, 0.0):7.1f})  {name}")
    prev[br] = t
