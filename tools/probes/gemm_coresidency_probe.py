#!/usr/bin/env python3
"""Do two eight-wave 128x64 GEMM workgroups on one CU slow each other down?  dX[4096, N] = dY[4096, K] . W[K, N] (kk_gemm, bf16) for
N = 512 (256 tiles: one workgroup per CU), 1024 (two per CU, co-resident) and 1536 (three: a second round), K = 512 .. 6144; per-launch
period of dependent launches in a replayed graph.  If N = 1024 costs what N = 512 costs, a CU has room for twice the waves on this loop —
the premise of splitting a one-tile-per-CU GEMM's reduction over two wave groups of one workgroup."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from kokoro_ruslan_amd import lib as kk

T, bf, REP = 4096, torch.bfloat16, 30


def period(fn):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / (10 * REP)


for K in (512, 1536, 3072, 6144):
    row = []
    for N in (512, 1024, 1536):
        dy, w = torch.randn(T, K, device="cuda").to(bf), (torch.randn(K, N, device="cuda") * 0.05).to(bf)
        dx = torch.empty(T, N, device="cuda", dtype=bf)
        t = period(lambda: kk.call("kk_gemm", 0, 1, T, N, K, 1.0, dy, K, w, N, 0.0, dx, N, None, None, 0, 0, 0, 1, 7))
        row.append(f"N={N}: {t:6.1f} us ({2.0 * T * N * K / t * 1e-6:4.0f} TFLOP/s)")
    print(f"dgrad K={K:5d}  " + "   ".join(row), flush=True)
