#!/usr/bin/env python3
"""GPU-side period of a chain of ~5 us dependent kernels: (P) one single-stream graph (packet-capture path), (A) the same chain in a
graph that also holds a forked branch (multi-queue path), (B) chain and branch as single-stream graphs on two streams joined by events."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

N, M = 300, 100
x = torch.ones(1 << 20, device="cuda")
y = torch.ones(1 << 18, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def chain(n):
    for _ in range(n):
        x.mul_(1.0)


def side(n):
    for _ in range(n):
        y.mul_(1.0)


def capture(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            fn()
    return g


def timed(tag, step, launches):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    h = (time.perf_counter() - t0) / 30
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 30
    print(f"{tag}: {t * 1e6:8.1f} us per replay = {t * 1e6 / launches:.2f} us per chain launch; host {h * 1e6:.0f} us")


def forked():
    chain(10)
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        side(M)
    chain(N - 20)
    torch.cuda.current_stream().wait_stream(s2)
    chain(10)


gP = capture(lambda: chain(N), s1)
gA = capture(forked, s1)
g0, g1, g2 = capture(lambda: chain(10), s1), capture(lambda: chain(N - 20), s1), capture(lambda: chain(10), s1)
gs = capture(lambda: side(M), s2)
e1, e2 = torch.cuda.Event(), torch.cuda.Event()


def stepP():
    with torch.cuda.stream(s1):
        gP.replay()


def stepA():
    with torch.cuda.stream(s1):
        gA.replay()


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


def stepE():
    with torch.cuda.stream(s1):
        chain(N)


timed("P single-stream graph, chain only   ", stepP, N)
timed("A one forked graph, chain + branch  ", stepA, N)
timed("B single-stream graphs, two streams ", stepB, N)
timed("E eager chain                       ", stepE, N)
