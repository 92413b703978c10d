#!/usr/bin/env python3
"""dQ + dK/dV as two launches against kk_attn_bwd's pair launch, decoder shape (B 8, h 8, S, bf16 storage, dropout 0.2, key mask,
head-norm epilogues), timed from hipGraph replays of 20 back-to-back launches (ctypes launches are host-bound below ~10 us)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
from oracle import kokoro_oracle as O
B, h = 8, 8
H = h * 64
bf = torch.bfloat16
def graph_time(fn, n=20, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / (n * reps) * 1e3
for S in (512, 1024):
    qkv = torch.randn(B * S, 3 * H, device="cuda").to(bf)
    raw = torch.randn(B * S, 3 * H, device="cuda").to(bf)
    q, k, v = qkv, qkv[:, H:], qkv[:, 2 * H:]
    o, do = torch.empty(B * S, H, device="cuda", dtype=bf), torch.randn(B * S, H, device="cuda").to(bf)
    lse, delta = torch.empty(B, h, S, device="cuda"), torch.empty(B, h, S, device="cuda")
    dqkv = torch.empty_like(qkv)
    seed = torch.tensor([7], dtype=torch.int32, device="cuda")
    km = torch.zeros(B, S, dtype=torch.uint8, device="cuda")
    gains = [torch.ones(64, device="cuda") for _ in range(3)]
    c, s = (t.cuda() for t in O.rope_tables(S, 64))
    nb = kk.load().kk_attn_bwd_blocks(B, h, S)
    part = torch.zeros(3, nb, 64, device="cuda")
    for causal in (0, 1):
        p = 0.2
        kk.call("kk_attn_fwd", q, k, v, o, lse, B, h, S, S, 3 * H, 3 * H, 3 * H, H, km, causal, 0.125, seed, 3, p, 1, 1)
        kk.call("kk_attn_delta", o, do, delta, B, h, S, H, H, 1)
        hq = kk.attn_headnorm([(raw, gains[0], part[0], c, s)])
        hkv = kk.attn_headnorm([(raw[:, H:], gains[1], part[1], c, s), (raw[:, 2 * H:], gains[2], part[2], None, None)])
        def dq(): kk.call("kk_attn_bwd_dq", q, k, v, do, lse, delta, dqkv, B, h, S, S, 3 * H, 3 * H, 3 * H, H, 3 * H, km, causal, 0.125, seed, 3, p, 1, 1, None, 0, hq)
        def dkv(): kk.call("kk_attn_bwd_dkv", q, k, v, do, lse, delta, dqkv[:, H:], dqkv[:, 2 * H:], B, h, S, S, 3 * H, 3 * H, 3 * H, H, 3 * H, 3 * H, km, causal, 0.125, seed, 3, p, 1, 1, hkv)
        def both(): dq(); dkv()
        def pair(): kk.call("kk_attn_bwd", q, k, v, do, lse, delta, dqkv, dqkv[:, H:], dqkv[:, 2 * H:], B, h, S, S, 3 * H, 3 * H, 3 * H, H, 3 * H, 3 * H, 3 * H, km, causal, 0.125, seed, 3, p, 1, 1, hq, hkv)
        def fwd(): kk.call("kk_attn_fwd", q, k, v, o, lse, B, h, S, S, 3 * H, 3 * H, 3 * H, H, km, causal, 0.125, seed, 3, p, 1, 1)
        print(f"S={S} causal={causal}: fwd {graph_time(fwd):6.1f}  dq {graph_time(dq):6.1f}  dkv {graph_time(dkv):6.1f}  dq+dkv {graph_time(both):6.1f}  pair {graph_time(pair):6.1f} us", flush=True)
