#!/usr/bin/env python3
"""Per-launch time of the attention forward storing keep bits (kk_attn_fwd_kb) against the one READING generated bits (kk_attn_fwd_rb),
and of the generator for the decoder's 12 sites, in replayed graphs.   python tools/probes/keepgen_bench.py [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd import lib as kk

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B, h, H, p = 8, 8, 512, 0.2
g = torch.Generator().manual_seed(1)
mk = lambda: (torch.randn(B * T, H, generator=g) * 0.7).cuda().to(torch.bfloat16)
seed = torch.tensor([5], dtype=torch.int32, device="cuda")
nb = kk.load().kk_attn_keep_bytes(B, h, T, T)
N = 12
Q, K, V = [mk() for _ in range(N)], [mk() for _ in range(N)], [mk() for _ in range(N)]
O = [torch.zeros(B * T, H, dtype=torch.bfloat16, device="cuda") for _ in range(N)]
L = [torch.zeros(B, h, T, device="cuda") for _ in range(N)]
KB = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(N)]
sites = kk.keep_sites([(KB[i], 2003 + 8 * i, p, B, h, T, T, i % 2 == 0) for i in range(N)])


def fwd(name, i):
    kk.call(name, Q[i], K[i], V[i], O[i], L[i], B, h, T, T, H, H, H, H, None, 1 if i % 2 == 0 else 0, 0.125, seed, 2003 + 8 * i, p, kk.KK_MATH_BF16, 1, KB[i])


def timed(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters


kk.call("kk_attn_keep_gen", sites, N, seed, 0, 0)
for r in range(3):
    tk = timed(lambda: [fwd("kk_attn_fwd_kb", i) for i in range(N)])
    tr = timed(lambda: [fwd("kk_attn_fwd_rb", i) for i in range(N)])
    tg = timed(lambda: kk.call("kk_attn_keep_gen", sites, N, seed, 0, 0))
    print(f"T={T}: 12 forwards (6 causal + 6 full) storing bits {tk:8.1f} us   reading bits {tr:8.1f} us   ({(tk - tr) / N:5.2f} us per launch)   generator, 12 sites, alone {tg:7.1f} us", flush=True)
