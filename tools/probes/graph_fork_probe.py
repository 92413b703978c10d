#!/usr/bin/env python3
"""Do parallel branches of a captured hipGraph overlap on this runtime?  Two chains of long single-workgroup GEMMs,
captured (a) back to back on one stream, (b) forked onto a second stream after a common first kernel and joined at
the end.  Overlap => (b) takes about half of (a)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from kokoro_ruslan_amd import lib as kk

bf = torch.bfloat16
K = 32768
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
A = [torch.randn(64, K, device="cuda").to(bf) for _ in range(2)]
B = [torch.randn(64, K, device="cuda").to(bf) for _ in range(2)]
C = [torch.zeros(64, 64, device="cuda") for _ in range(2)]


def gemm(i):
    kk.call("kk_gemm", 0, 0, 64, 64, K, 1.0, A[i], K, B[i], K, 0.0, C[i], 64, None, None, 0, 0, 1, 1, 3)


side = torch.cuda.Stream()


def serial():
    gemm(0)
    for _ in range(n):
        gemm(0)
    for _ in range(n):
        gemm(1)


def forked():
    gemm(0)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(n):
            gemm(1)
    for _ in range(n):
        gemm(0)
    torch.cuda.current_stream().wait_stream(side)


def run(fn, label):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t) / 20 * 1e6:8.1f} us per replay")
    t = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{label} (eager): {(time.perf_counter() - t) / 20 * 1e6:8.1f} us")


run(serial, "one stream ")
run(forked, "two streams")
