#!/usr/bin/env python3
"""Micro-benchmark of kk_gemm on the Linear shapes of the train step (M = B*T rows).  Prints TFLOP/s per shape for
the three layouts under a few tuning settings.  python tools/gemm_bench.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kokoro_ruslan_amd import lib as kk

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SHAPES = [(512, 1536), (512, 512), (512, 1024), (512, 3072), (1536, 512)]     # (K_in, N_out)
math_mode = 1


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


DT = 0


def run(label):
    out = []
    act = torch.bfloat16 if DT else torch.float32
    for K, N in SHAPES:
        x, w = torch.randn(M, K, device="cuda").to(act), torch.randn(N, K, device="cuda").to(act)
        y, dy = torch.empty(M, N, device="cuda", dtype=act), torch.randn(M, N, device="cuda").to(act)
        dx, dw = torch.empty(M, K, device="cuda", dtype=act), torch.zeros(N, K, device="cuda")
        fl = 2.0 * M * N * K
        t1 = timeit(lambda: kk.call("kk_gemm", 0, 0, M, N, K, 1.0, x, K, w, K, 0.0, y, N, None, None, 0, 0, 1, math_mode, DT))
        t2 = timeit(lambda: kk.call("kk_gemm", 0, 1, M, K, N, 1.0, dy, N, w, K, 0.0, dx, K, None, None, 0, 0, 1, math_mode, DT))
        t3 = timeit(lambda: kk.call("kk_gemm", 1, 1, N, K, M, 1.0, dy, N, x, K, 1.0, dw, K, None, None, 0, 0, 0, math_mode, DT & 3))
        out.append(f"K{K}xN{N}: fwd {fl / t1 / 1e12:6.1f} dgrad {fl / t2 / 1e12:6.1f} wgrad {fl / t3 / 1e12:6.1f}")
    print(f"[{label}] M={M}  " + " | ".join(out))


for dt, thr, swz in ((0, 512, 1), (7, 512, 1), (7, 256, 1), (7, 128, 1), (7, 1, 1), (7, 100000, 1)):
    DT = dt
    kk.gemm_tune(thr, swz)
    run(f"dtypes={dt} tm128 if tiles>={thr}, xcd_swizzle={swz}")
