#!/usr/bin/env python3
"""Split-K sweep of kk_gemm on the train step's shapes (bf16 storage): microseconds per launch for each split count."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kokoro_ruslan_amd import lib as kk
from gemm_bench import timeit

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bf = torch.bfloat16
print("wgrad dW[M,N] += dY^T[M,T] X[T,N]   (ta=1, tb=1, K=T tokens)")
for M, N in [(3072, 512), (512, 512), (1536, 512), (512, 1536), (1024, 512), (256, 1536), (80, 512)]:
    dy, x = torch.randn(T, M, device="cuda").to(bf), torch.randn(T, N, device="cuda").to(bf)
    dw = torch.zeros(M, N, device="cuda")
    row = []
    for sp in (0, 1, 2, 4, 8, 16, 32):
        t = timeit(lambda: kk.call("kk_gemm", 1, 1, M, N, T, 1.0, dy, M, x, N, 1.0, dw, N, None, None, 0, 0, sp, 1, 3))
        row.append(f"s{sp}:{t * 1e6:6.1f}")
    print(f"  M={M:5d} N={N:5d} " + " ".join(row))
print("fwd Y[T,N] = X[T,K] W[N,K]^T  (bf16 out: no split; fp32 out: split allowed)")
for K, N in [(512, 1536), (512, 512), (512, 1024), (512, 3072), (1536, 512)]:
    x, w = torch.randn(T, K, device="cuda").to(bf), torch.randn(N, K, device="cuda").to(bf)
    y16, y32 = torch.empty(T, N, device="cuda", dtype=bf), torch.empty(T, N, device="cuda")
    t16 = timeit(lambda: kk.call("kk_gemm", 0, 0, T, N, K, 1.0, x, K, w, K, 0.0, y16, N, None, None, 0, 0, 1, 1, 7))
    row = [f"bf16out:{t16 * 1e6:6.1f}"]
    for sp in (1, 2, 4):
        t = timeit(lambda: kk.call("kk_gemm", 0, 0, T, N, K, 1.0, x, K, w, K, 0.0, y32, N, None, None, 0, 0, sp, 1, 3))
        row.append(f"f32 s{sp}:{t * 1e6:6.1f}")
    print(f"  K={K:5d} N={N:5d} " + " ".join(row))
print("dgrad dX[T,K] = dY[T,N] W[N,K]  (ta=0, tb=1)")
for K, N in [(512, 1536), (512, 512), (512, 1024), (512, 3072), (1536, 512)]:
    dy, w = torch.randn(T, N, device="cuda").to(bf), torch.randn(N, K, device="cuda").to(bf)
    d16, d32 = torch.empty(T, K, device="cuda", dtype=bf), torch.empty(T, K, device="cuda")
    t16 = timeit(lambda: kk.call("kk_gemm", 0, 1, T, K, N, 1.0, dy, N, w, K, 0.0, d16, K, None, None, 0, 0, 1, 1, 7))
    row = [f"bf16out:{t16 * 1e6:6.1f}"]
    for sp in (1, 2, 4):
        t = timeit(lambda: kk.call("kk_gemm", 0, 1, T, K, N, 1.0, dy, N, w, K, 0.0, d32, K, None, None, 0, 0, sp, 1, 3))
        row.append(f"f32 s{sp}:{t * 1e6:6.1f}")
    print(f"  K={K:5d} N={N:5d} " + " ".join(row))
