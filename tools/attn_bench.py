#!/usr/bin/env python3
"""Isolated timings of the three attention kernels at the decoder shape (B 8, h 8, S 512, bf16 storage)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kokoro_ruslan_amd import lib as kk
B, h, S = 8, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = h * 64
bf = torch.bfloat16
qkv = torch.randn(B * S, 3 * H, device="cuda").to(bf)
q, k, v = qkv, qkv[:, H:], qkv[:, 2 * H:]
o, do = torch.empty(B * S, H, device="cuda", dtype=bf), torch.randn(B * S, H, device="cuda").to(bf)
lse, delta = torch.empty(B, h, S, device="cuda"), torch.empty(B, h, S, device="cuda")
dqkv = torch.empty_like(qkv)
seed = torch.tensor([7], dtype=torch.int32, device="cuda")
km = torch.zeros(B, S, dtype=torch.uint8, device="cuda")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for causal in (0, 1):
    for p in (0.0, 0.2):
        for mask in (None, km):
            f = lambda: kk.call("kk_attn_fwd", q, k, v, o, lse, B, h, S, S, 3 * H, 3 * H, 3 * H, H, mask, causal, 0.125, seed, 3, p, 1, 1)
            t1 = timeit(f)
            kk.call("kk_attn_delta", o, do, delta, B, h, S, H, H, 1)
            t2 = timeit(lambda: kk.call("kk_attn_bwd_dq", q, k, v, do, lse, delta, dqkv, B, h, S, S, 3 * H, 3 * H, 3 * H, H, 3 * H, mask, causal, 0.125, seed, 3, p, 1, 1, None, 0, None))
            t3 = timeit(lambda: kk.call("kk_attn_bwd_dkv", q, k, v, do, lse, delta, dqkv[:, H:], dqkv[:, 2 * H:], B, h, S, S, 3 * H, 3 * H, 3 * H, H, 3 * H, 3 * H, mask, causal, 0.125, seed, 3, p, 1, 1, None))
            print(f"S={S} causal={causal} p={p} mask={'y' if mask is not None else 'n'}: fwd {t1:6.1f} us  dq {t2:6.1f} us  dkv {t3:6.1f} us")
