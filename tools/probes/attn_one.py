#!/usr/bin/env python3
"""One attention kernel, a few launches (for rocprofv3 --pmc passes):  attn_one.py fwd|dq|dkv|pair S causal p [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
which, S, causal, p = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10
B, h = 8, 8
H = h * 64
bf = torch.bfloat16
qkv = torch.randn(B * S, 3 * H, device="cuda").to(bf)
q, k, v = qkv, qkv[:, H:], qkv[:, 2 * H:]
o, do = torch.empty(B * S, H, device="cuda", dtype=bf), torch.randn(B * S, H, device="cuda").to(bf)
lse, delta = torch.empty(B, h, S, device="cuda"), torch.empty(B, h, S, device="cuda")
dqkv = torch.empty_like(qkv)
seed = torch.tensor([7], dtype=torch.int32, device="cuda")
km = torch.zeros(B, S, dtype=torch.uint8, device="cuda")
kk.call("kk_attn_fwd", q, k, v, o, lse, B, h, S, S, 3 * H, 3 * H, 3 * H, H, km, causal, 0.125, seed, 3, p, 1, 1)
kk.call("kk_attn_delta", o, do, delta, B, h, S, H, H, 1)
torch.cuda.synchronize()
for _ in range(n):
    if which == "fwd":
        kk.call("kk_attn_fwd", q, k, v, o, lse, B, h, S, S, 3 * H, 3 * H, 3 * H, H, km, causal, 0.125, seed, 3, p, 1, 1)
    elif which == "dq":
        kk.call("kk_attn_bwd_dq", q, k, v, do, lse, delta, dqkv, B, h, S, S, 3 * H, 3 * H, 3 * H, H, 3 * H, km, causal, 0.125, seed, 3, p, 1, 1, None, 0, None)
    elif which == "pair":
        kk.call("kk_attn_bwd", q, k, v, do, lse, delta, dqkv, dqkv[:, H:], dqkv[:, 2 * H:], B, h, S, S, 3 * H, 3 * H, 3 * H, H, 3 * H, 3 * H, 3 * H,
                km, causal, 0.125, seed, 3, p, 1, 1, None, None)
    else:
        kk.call("kk_attn_bwd_dkv", q, k, v, do, lse, delta, dqkv[:, H:], dqkv[:, 2 * H:], B, h, S, S, 3 * H, 3 * H, 3 * H, H, 3 * H, 3 * H, km, causal, 0.125, seed, 3, p, 1, 1, None)
torch.cuda.synchronize()
