#!/usr/bin/env python3
"""Single-queue graphs joined by events against one forked graph: the per-launch gap of a chain and the cost of a boundary.
A: ONE graph with a forked branch (multi-queue launch path).  B: the same work as single-stream graphs on two streams joined by
events (each graph takes the packet-capture path).  Time stamps by one-thread kernels (kk_timestamp, 100 MHz)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from kokoro_ruslan_amd import lib as kk

N, M = 150, 100                       # chain launches per segment, side launches
buf = torch.zeros(4 * N + M + 8, dtype=torch.int64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def seg(lo, n):
    for i in range(n):
        kk.call("kk_timestamp", buf[lo + i:])


def capture(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            fn()
    return g


def report(tag, host_s):
    t = buf.cpu().tolist()
    def gaps(lo, n):
        d = sorted((t[lo + i + 1] - t[lo + i]) / 100.0 for i in range(n - 1))
        return d[len(d) // 2]
    print(f"{tag}: chain seg0 gap {gaps(0, N):.2f} us, seg1 gap {gaps(N, N):.2f}, seg2 gap {gaps(2 * N, N):.2f}, side gap {gaps(3 * N, M):.2f}; "
          f"boundary seg0->seg1 {(t[N] - t[N - 1]) / 100.0:.2f} us, seg1->seg2 {(t[2 * N] - t[2 * N - 1]) / 100.0:.2f} us, "
          f"side start after seg0 end {(t[3 * N] - t[N - 1]) / 100.0:.2f} us, total {(t[3 * N - 1] - t[0]) / 100.0:.1f} us, host {host_s * 1e6:.0f} us per replay")


# A: one graph, fork after seg0, join before seg2
def whole():
    seg(0, N)
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        seg(3 * N, M)
    seg(N, N)
    torch.cuda.current_stream().wait_stream(s2)
    seg(2 * N, N)


gA = capture(whole, s1)
with torch.cuda.stream(s1):
    for _ in range(3):
        gA.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        gA.replay()
    h = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
report("A one forked graph      ", h)

# B: four single-stream graphs
g0, g1, g2 = capture(lambda: seg(0, N), s1), capture(lambda: seg(N, N), s1), capture(lambda: seg(2 * N, N), s1)
gs = capture(lambda: seg(3 * N, M), s2)
e1, e2 = torch.cuda.Event(), torch.cuda.Event()


def stepB():
    with torch.cuda.stream(s1):
        g0.replay()
        e1.record(s1)
    with torch.cuda.stream(s2):
        s2.wait_event(e1)
        gs.replay()
        e2.record(s2)
    with torch.cuda.stream(s1):
        g1.replay()
        s1.wait_event(e2)
        g2.replay()


for _ in range(3):
    stepB()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    stepB()
h = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
report("B single-stream segments", h)
