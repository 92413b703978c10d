#!/usr/bin/env python3
"""Gap between dependent launches inside a captured hipGraph: a chain of one-thread time-stamp kernels on one stream,
(a) alone, (b) with a second stream forked beside it (which switches the graph to the multi-queue launch path)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from kokoro_ruslan_amd import lib as kk

N = 200
buf = torch.zeros(N + 8, dtype=torch.int64, device="cuda")
side = torch.cuda.Stream()
x = torch.zeros(1 << 20, device="cuda")


def chain(fork):
    if fork:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(20):
                x.add_(1.0)
    for i in range(N):
        kk.call("kk_timestamp", buf[i:])
    if fork:
        torch.cuda.current_stream().wait_stream(side)


for fork in (False, True):
    chain(fork)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(fork)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t = buf[:N].cpu().tolist()
    d = sorted((t[i + 1] - t[i]) / 100.0 for i in range(20, N - 1))
    print(f"fork={fork}: gap between dependent one-thread kernels in a graph: median {d[len(d) // 2]:.2f} us, p10 {d[len(d) // 10]:.2f}, p90 {d[9 * len(d) // 10]:.2f}")
chain(False)
torch.cuda.synchronize()
t = buf[:N].cpu().tolist()
d = sorted((t[i + 1] - t[i]) / 100.0 for i in range(20, N - 1))
print(f"eager: median {d[len(d) // 2]:.2f} us")
