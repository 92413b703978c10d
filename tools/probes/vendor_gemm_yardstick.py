#!/usr/bin/env python3
"""Yardstick only (not a product path): what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on this model's GEMM shapes, as
dependent launches inside a replayed graph, next to the repo's bf16 core (kk_gemm).  Per-launch period, TFLOP/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from kokoro_ruslan_amd import lib as kk

dev = "cuda"
shapes = [  # (tag, layout, M, N, K)   fwd: Y = X.W^T ; dgrad: dX = dY.W ; wgrad: dW = dY^T.X
    ("w_o fwd", "fwd", 4096, 512, 512), ("qkv fwd", "fwd", 4096, 1536, 512), ("linear1 fwd", "fwd", 4096, 3072, 512),
    ("linear2 fwd", "fwd", 4096, 512, 1536), ("w_o dgrad", "dgrad", 4096, 512, 512), ("linear1 dgrad", "dgrad", 4096, 512, 3072),
    ("linear2 dgrad", "dgrad", 4096, 1536, 512), ("w_o wgrad", "wgrad", 512, 512, 4096), ("qkv wgrad", "wgrad", 1536, 512, 4096),
    ("linear1 wgrad", "wgrad", 3072, 512, 4096), ("linear2 wgrad", "wgrad", 512, 1536, 4096),
]
REP = 40


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


for tag, lay, M, N, K in shapes:
    if lay == "fwd":
        A, B = torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        vend = lambda: torch.matmul(A, B.t(), out=C)
        ta, tb = 0, 1
    elif lay == "dgrad":
        A, B = torch.randn(M, K, device=dev).bfloat16(), torch.randn(K, N, device=dev).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        vend = lambda: torch.matmul(A, B, out=C)
        ta, tb = 0, 0
    else:
        A, B = torch.randn(K, M, device=dev).bfloat16(), torch.randn(K, N, device=dev).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.float32)
        C16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        vend = lambda: torch.matmul(A.t(), B, out=C16)
        ta, tb = 1, 0
    tv = period(vend)
    fl = 2.0 * M * N * K
    print(f"{tag:14s} {lay:5s} M={M:5d} N={N:5d} K={K:5d}: vendor {tv:6.2f} us = {fl / tv * 1e-6:6.0f} TFLOP/s", flush=True)
