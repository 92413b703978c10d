#!/usr/bin/env python3
"""Attention with stored keep bits against the hashing kernels: kk_attn_fwd / kk_attn_fwd_kb and kk_attn_bwd / kk_attn_bwd_kb at the
decoder's shapes (dropout 0.2, head-norm epilogues, graph replays of R independent operand sets).
    python tools/probes/attn_keep_ab.py [--lib NAME]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd import lib as kk
from oracle import kokoro_oracle as O

if "--lib" in sys.argv:
    kk.use_library(sys.argv[sys.argv.index("--lib") + 1])
R = 6
P = 0.2


def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps / R * 1e3


shapes = ((8, 8, 512, 1), (8, 8, 512, 0), (8, 8, 1024, 1), (8, 8, 1024, 0), (12, 8, 1333, 1))
for B, h, S, causal in shapes:
    H = h * 64
    sets = []
    nbytes = kk.load().kk_attn_keep_bytes(B, h, S, S)
    for i in range(R):
        q, kv = torch.randn(B * S, H, device="cuda").bfloat16(), torch.randn(B * S, 2 * H, device="cuda").bfloat16()
        raw_q, raw_kv = torch.randn_like(q), torch.randn_like(kv)
        do = torch.randn(B * S, H, device="cuda").bfloat16()
        o, lse = torch.empty_like(q), torch.empty(B, h, S, device="cuda")
        seed = torch.tensor([3], dtype=torch.int32, device="cuda")
        km = torch.zeros(B, S, dtype=torch.uint8, device="cuda"); km[:, S - 20:] = 1
        km = None if causal else km
        keep = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        kk.call("kk_attn_fwd_kb", q, kv, kv[:, H:], o, lse, B, h, S, S, H, 2 * H, 2 * H, H, km, causal, 0.125, seed, 5, P, 1, 1, keep)
        delta = torch.empty(B, h, S, device="cuda")
        kk.call("kk_attn_delta", o, do, delta, B, h, S, H, H, 1)
        nb = kk.load().kk_attn_bwd_blocks(B, h, S)
        gains = [torch.ones(64, device="cuda") for _ in range(3)]
        c, s = (t.cuda() for t in O.rope_tables(S, 64))
        pq, pkv = torch.zeros(1, nb, 64, device="cuda"), torch.zeros(2, nb, 64, device="cuda")
        hq = kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)])
        hkv = kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)])
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        sets.append((q, kv, do, o, lse, delta, dq, dkv, km, seed, hq, hkv, keep, raw_q, raw_kv, gains, c, s, pq, pkv))

    def fwd(kb):
        def f():
            for (q, kv, do, o, lse, delta, dq, dkv, km, seed, hq, hkv, keep, *_) in sets:
                a = (q, kv, kv[:, H:], o, lse, B, h, S, S, H, 2 * H, 2 * H, H, km, causal, 0.125, seed, 5, P, 1, 1)
                if kb: kk.call("kk_attn_fwd_kb", *a, keep)
                else: kk.call("kk_attn_fwd", *a)
        return f

    def bwd(kb):
        def f():
            for (q, kv, do, o, lse, delta, dq, dkv, km, seed, hq, hkv, keep, *_) in sets:
                a = (q, kv, kv[:, H:], do, lse, delta, dq, dkv, dkv[:, H:], B, h, S, S, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km, causal, 0.125, seed, 5, P, 1, 1, hq, hkv)
                if kb: kk.call("kk_attn_bwd_kb", *a, keep)
                else: kk.call("kk_attn_bwd", *a)
        return f
    t = {}
    for rnd in range(2):                                  # interleaved
        for name, fn in (("fwd", fwd(False)), ("fwd_kb", fwd(True)), ("bwd", bwd(False)), ("bwd_kb", bwd(True))):
            t.setdefault(name, []).append(graph_time(fn))
    ff = 2 * 2.0 * B * h * S * S * 64 * (0.5 if causal else 1.0)
    fb = 2 * ff
    m = {k: min(v) for k, v in t.items()}
    print(f"B={B} S={S} causal={causal}: fwd {m['fwd']:6.2f} -> kb {m['fwd_kb']:6.2f} us ({ff / m['fwd_kb'] / 1e6:4.0f} TFLOP/s)   "
          f"bwd {m['bwd']:7.2f} -> kb {m['bwd_kb']:7.2f} us ({fb / m['bwd_kb'] / 1e6:4.0f} TFLOP/s = {fb / m['bwd_kb'] / 2.5e7:.1f} %)", flush=True)
