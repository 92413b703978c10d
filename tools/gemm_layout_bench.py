#!/usr/bin/env python3
"""The DMA-staged bf16 GEMM core at one problem size in its four operand layouts (k-contiguous / k-strided A and B),
no split-K: separates what the transpose-read (k-strided) path costs from what the tile shape costs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kokoro_ruslan_amd import lib as kk

bf = torch.bfloat16


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, N, K in [(4096, 512, 4096), (2048, 2048, 4096), (4096, 4096, 512), (4096, 512, 512)]:
    out = []
    for ta in (0, 1):
        for tb in (0, 1):
            A = torch.randn((K, M) if ta else (M, K), device="cuda").to(bf)
            B = torch.randn((K, N) if tb else (N, K), device="cuda").to(bf)
            Cm = torch.zeros(M, N, device="cuda")
            for stages in (2, 3):
                kk.gemm_tune16(1, 4096, 4096, stages * 10000 + 384)
                t = timeit(lambda: kk.call("kk_gemm", ta, tb, M, N, K, 1.0, A, A.stride(0), B, B.stride(0), 0.0, Cm, N, None, None,
                                           0, 0, 1, 1, 3))
                out.append(f"ta{ta} tb{tb} NS{stages}: {t:6.1f}us {2.0 * M * N * K / t / 1e6:4.0f}TF")
    print(f"M={M} N={N} K={K}: " + " | ".join(out))
