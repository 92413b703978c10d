#!/usr/bin/env python3
"""kk_gemm on the train step's bf16 x bf16 shapes: old register-staged core vs the DMA-staged core (microseconds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kokoro_ruslan_amd import lib as kk

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bf = torch.bfloat16


def timeit(fn, n=40):
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


def cases():
    for K, N in [(512, 1536), (512, 512), (512, 1024), (512, 3072), (1536, 512)]:
        x, w = torch.randn(T, K, device="cuda").to(bf), torch.randn(N, K, device="cuda").to(bf)
        y = torch.empty(T, N, device="cuda", dtype=bf)
        yield f"fwd   K={K:5d} N={N:5d}", 2.0 * T * N * K, (lambda x=x, w=w, y=y, K=K, N=N: kk.call(
            "kk_gemm", 0, 0, T, N, K, 1.0, x, K, w, K, 0.0, y, N, None, None, 0, 0, 0, 1, 7))
    for K, N in [(512, 1536), (512, 512), (512, 1024), (512, 3072), (1536, 512)]:
        dy, w = torch.randn(T, N, device="cuda").to(bf), torch.randn(N, K, device="cuda").to(bf)
        dx = torch.empty(T, K, device="cuda", dtype=bf)
        yield f"dgrad K={K:5d} N={N:5d}", 2.0 * T * N * K, (lambda dy=dy, w=w, dx=dx, K=K, N=N: kk.call(
            "kk_gemm", 0, 1, T, K, N, 1.0, dy, N, w, K, 0.0, dx, K, None, None, 0, 0, 0, 1, 7))
    for M, N in [(3072, 512), (512, 512), (1536, 512), (512, 1536), (1024, 512), (256, 1536)]:
        dy, x = torch.randn(T, M, device="cuda").to(bf), torch.randn(T, N, device="cuda").to(bf)
        dw = torch.zeros(M, N, device="cuda")
        yield f"wgrad M={M:5d} N={N:5d}", 2.0 * T * N * M, (lambda dy=dy, x=x, dw=dw, M=M, N=N: kk.call(
            "kk_gemm", 1, 1, M, N, T, 1.0, dy, M, x, N, 1.0, dw, N, None, None, 0, 0, 0, 1, 3))


configs = [("old", (0, 0, 0, 0)), ("dma default", (1, 4096, 4096, 384))]
for spec in sys.argv[2:]:
    a, b, c = (int(v) for v in spec.split(","))
    configs.append((f"dma {spec}", (1, a, b, c)))
rows = {}
for label, cfg in configs:
    kk.gemm_tune16(*cfg)
    for name, fl, fn in cases():
        t = timeit(fn)
        rows.setdefault(name, []).append(f"{label}: {t:6.1f}us {fl / t / 1e6:5.0f}TF")
for name, r in rows.items():
    print(name, " | ".join(r))
