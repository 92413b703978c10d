#!/usr/bin/env python3
"""What does ONE launch on a forked branch cost the main chain of the same hipGraph?  Main chain: 200 dependent chip-filling kernels
(~6 us each).  Branch: N dependent one-thread kernels (kk_timestamp) or N small-grid kernels, forked after the 10th chain launch.
The host is kept ahead (replays queued back to back); time per replay by events."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from kokoro_ruslan_amd import lib as kk

x = torch.ones(1 << 22, device="cuda")
MAIN = os.environ.get("MAIN", "stream")      # stream: a 32 MB read-modify-write; gemm: a 4096 x 512 x 512 bf16 GEMM (cache-resident operands)
ga, gb = torch.randn(4096, 512, device="cuda").bfloat16(), torch.randn(512, 512, device="cuda").bfloat16()
gc = torch.empty(4096, 512, device="cuda", dtype=torch.bfloat16)


def main_op():
    if MAIN == "gemm":
        torch.matmul(ga, gb.t(), out=gc)
    else:
        x.mul_(1.0)


y = torch.ones(1 << 14, device="cuda")
buf = torch.zeros(1024, dtype=torch.int64, device="cuda")
side = torch.cuda.Stream()
NMAIN = 200


def step(n_side, kind):
    for _ in range(10):
        main_op()
    if n_side:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(n_side):
                if kind == "stamp":
                    kk.call("kk_timestamp", buf[i:])
                else:
                    y.mul_(1.0)
    for _ in range(NMAIN - 10):
        main_op()
    if n_side:
        torch.cuda.current_stream().wait_stream(side)


def measure(n_side, kind):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(n_side, kind)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step(n_side, kind)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1000.0


base = measure(0, "stamp")
print(f"main chain alone (single-stream graph): {base:8.1f} us = {base / NMAIN:.2f} us per launch")
for kind in ("stamp", "small"):
    for n in (1, 50, 100, 200, 400):
        t = measure(n, kind)
        print(f"branch of {n:3d} {kind:5s} launches: {t:8.1f} us  (+{t - base:7.1f} us, {(t - base) / n:6.2f} us per branch launch)")
