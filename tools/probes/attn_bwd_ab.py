#!/usr/bin/env python3
"""Attention backward: kk_attn_bwd (pair launch) against kk_attn_bwd_ws (dK/dV + dS, then the dQ pass), decoder shapes, graph replays.
    python tools/probes/attn_bwd_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd import lib as kk
from oracle import kokoro_oracle as O

if "--lib" in sys.argv:
    kk.use_library(sys.argv[sys.argv.index("--lib") + 1])
R = 6
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

for B, h, S, causal in ((8, 8, 512, 1), (8, 8, 512, 0), (8, 8, 1024, 1), (8, 8, 1024, 0), (16, 8, 512, 1), (4, 8, 2048, 0)):
    H = h * 64
    sets = []
    for i in range(R):
        q, kv = torch.randn(B * S, H, device="cuda").bfloat16(), torch.randn(B * S, 2 * H, device="cuda").bfloat16()
        raw_q, raw_kv = torch.randn_like(q), torch.randn_like(kv)
        do = torch.randn(B * S, H, device="cuda").bfloat16()
        o, lse = torch.empty_like(q), torch.empty(B, h, S, device="cuda")
        seed = torch.tensor([3], dtype=torch.int32, device="cuda")
        km = torch.zeros(B, S, dtype=torch.uint8, device="cuda"); km[:, S - 20:] = 1
        km = None if causal else km
        kk.call("kk_attn_fwd", q, kv, kv[:, H:], o, lse, B, h, S, S, H, 2 * H, 2 * H, H, km, causal, 0.125, seed, 5, 0.2, 1, 1)
        delta = torch.empty(B, h, S, device="cuda")
        kk.call("kk_attn_delta", o, do, delta, B, h, S, H, H, 1)
        nb = kk.load().kk_attn_bwd_blocks(B, h, S)
        gains = [torch.ones(64, device="cuda") for _ in range(3)]
        c, s = (t.cuda() for t in O.rope_tables(S, 64))
        pq, pkv = torch.zeros(1, nb, 64, device="cuda"), torch.zeros(2, nb, 64, device="cuda")
        hq = kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)])
        hkv = kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)])
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        sets.append((q, kv, do, lse, delta, dq, dkv, km, seed, hq, hkv, raw_q, raw_kv, gains, c, s, pq, pkv))
    need = kk.load().kk_attn_bwd_ws_bytes(B, h, S, S)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    def run(two):
        def f():
            for (q, kv, do, lse, delta, dq, dkv, km, seed, hq, hkv, *_) in sets:
                a = (q, kv, kv[:, H:], do, lse, delta, dq, dkv, dkv[:, H:], B, h, S, S, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km, causal, 0.125, seed, 5, 0.2, 1, 1, hq, hkv)
                if two: kk.call("kk_attn_bwd_ws", *a, ws, need)
                else: kk.call("kk_attn_bwd", *a)
        return f
    def dkv_only():
        for (q, kv, do, lse, delta, dq, dkv, km, seed, hq, hkv, *_) in sets:
            kk.call("kk_attn_bwd_dkv", q, kv, kv[:, H:], do, lse, delta, dkv, dkv[:, H:], B, h, S, S, H, 2 * H, 2 * H, H, 2 * H, 2 * H, km, causal, 0.125, seed, 5, 0.2, 1, 1, hkv)
    ta, tb, tc = graph_time(run(False)), graph_time(run(True)), graph_time(dkv_only)
    ta2, tb2 = graph_time(run(False)), graph_time(run(True))
    fl = 4 * 2.0 * B * h * S * S * 64 * (0.5 if causal else 1.0)
    print(f"B={B} S={S} causal={causal}: pair {ta:7.2f} / {ta2:7.2f} us   two-pass {tb:7.2f} / {tb2:7.2f} us   (dK/dV kernel alone {tc:7.2f})   "
          f"{fl / min(tb, tb2) / 1e6:6.0f} TFLOP/s two-pass vs {fl / min(ta, ta2) / 1e6:6.0f} pair; ws {need / 2**20:.0f} MB", flush=True)
