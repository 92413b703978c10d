#!/usr/bin/env python3
"""Weight-gradient GEMMs of the train step (dW[M,N] += dY[T,M]^T . X[T,N], bf16 operands, fp32 accumulate into dW) under
different tile / split-K / pipeline-depth settings of the DMA-staged core.  usage: wgrad_bench.py [T] thr128,thr12864,target ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kokoro_ruslan_amd import lib as kk

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
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


def cases():
    for M, N in [(4096, 512), (512, 2048), (1536, 512), (512, 512), (6144, 512), (1024, 512)]:
        dy, x = torch.randn(T, M, device="cuda").to(bf), torch.randn(T, N, device="cuda").to(bf)
        dw = torch.zeros(M, N, device="cuda")
        yield f"wgrad M={M:5d} N={N:5d}", 2.0 * T * N * M, (lambda dy=dy, x=x, dw=dw, M=M, N=N: kk.call(
            "kk_gemm", 1, 1, M, N, T, 1.0, dy, M, x, N, 1.0, dw, N, None, None, 0, 0, 0, 1, 3))


configs = [("default", (1, 4096, 4096, 768))]
for spec in sys.argv[2:]:
    a, b, c = (int(v) for v in spec.split(","))
    configs.append((spec, (1, a, b, c)))
rows = {}
for label, cfg in configs:
    kk.gemm_tune16(*cfg)
    for name, fl, fn in cases():
        t = timeit(fn)
        rows.setdefault(name, []).append(f"{label}: {t:6.1f}us {fl / t / 1e6:4.0f}TF")
for name, r in rows.items():
    print(name, " | ".join(r))

# one encoder layer's / one decoder layer's weight gradients as a grouped launch
for label, shapes in [("enc layer", [(512, 2048), (4096, 512), (512, 512), (1536, 512)]),
                      ("dec layer", [(512, 2048), (4096, 512), (512, 512), (512, 512), (512, 512), (1536, 512)])]:
    probs = [(torch.randn(T, M, device="cuda").to(bf), torch.randn(T, N, device="cuda").to(bf), torch.zeros(M, N, device="cuda"))
             for M, N in shapes]
    fl = sum(2.0 * T * M * N for M, N in shapes)
    table = kk.wgrad_table(probs)
    out = []
    for stages in (2, 3):
        kk.gemm_tune16(1, 4096, 4096, stages * 10000 + 384)
        for split in (1, 2, 101, 102, 201, 202):
            kk.gemm_tune_group(split)
            t = timeit(lambda: kk.call("kk_gemm_wgrad_group", table, len(probs), 0, 0, None, None, None))
            out.append(f"NS {stages} split {split}: {t:6.1f}us {fl / t / 1e6:4.0f}TF")
    kk.gemm_tune_group(0)
    print("group", label, " | ".join(out))
