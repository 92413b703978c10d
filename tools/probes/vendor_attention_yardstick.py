#!/usr/bin/env python3
"""Yardstick only (not a product path): PyTorch-ROCm's own fused attention (F.scaled_dot_product_attention: flash / efficient backends)
forward and backward on the decoder's attention shapes, as dependent launches of a replayed graph, next to this repo's kernels
(tools/attn_bench.py gives theirs on the same box).  bf16, B 8, h 8, d 64, dropout 0.2."""
import os
import sys

import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

REP = 20


def period(fn):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(REP):
                    fn()
            run = g.replay
            per = REP
        except Exception as e:                      # a backend that cannot be captured: eager launches (host-bound below ~10 us)
            print("   (not capturable:", type(e).__name__, "- eager timing)")
            torch.cuda.synchronize()
            run, per = fn, 1
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / (10 * per)


for S in (512, 1024):
    for causal in (False, True):
        for p in (0.0, 0.2):
            q, k, v = (torch.randn(8, 8, S, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
            do = torch.randn(8, 8, S, 64, device="cuda", dtype=torch.bfloat16)
            for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION)):
                try:
                    with sdpa_kernel(be):
                        tf = period(lambda: F.scaled_dot_product_attention(q, k, v, dropout_p=p, is_causal=causal))

                        def fb():
                            o = F.scaled_dot_product_attention(q, k, v, dropout_p=p, is_causal=causal)
                            torch.autograd.grad(o, (q, k, v), do)
                        tfb = period(fb)
                    fl = 4.0 * 8 * 8 * S * S * 64 * (0.5 if causal else 1.0)
                    print(f"S={S:5d} causal={int(causal)} p={p}: {name:9s} fwd {tf:7.1f} us ({fl / tf * 1e-6:5.0f} TFLOP/s)   fwd+bwd {tfb:7.1f} us  -> bwd {tfb - tf:7.1f} us "
                          f"({2 * fl / (tfb - tf) * 1e-6:5.0f} TFLOP/s, 4-matmul count)", flush=True)
                except Exception as e:
                    print(f"S={S} causal={int(causal)} p={p}: {name}: {type(e).__name__}: {str(e)[:100]}", flush=True)
