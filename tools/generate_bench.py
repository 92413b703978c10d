#!/usr/bin/env python3
"""KokoroEngine.generate (autoregressive decode with the KV cache) at the default model size, random weights, stop head
disabled: mel frames per second of wall time, fp32 parity mode and bf16 mode."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ids = torch.randint(1, 59, (1, 64), generator=torch.Generator().manual_seed(0)).cuda()
for mode in ("f32", "bf16"):
    e = KokoroEngine(ModelDims(), StepHyper(), math_mode=mode, total_steps=100, seed=0)
    for graph in (False, True):
        e.generate(ids, max_len=20, stop_threshold=2.0, decode_graph=graph)                 # warm-up: workspaces, kernel attributes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mel = e.generate(ids, max_len=frames, stop_threshold=2.0, decode_graph=graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{mode} {'one replayed graph per frame' if graph else 'eager launches':30s}: {mel.shape[1]} frames in {dt * 1e3:.1f} ms = "
              f"{mel.shape[1] / dt:.0f} frames/s ({dt / mel.shape[1] * 1e3:.3f} ms/frame), finite={bool(torch.isfinite(mel).all())}")
