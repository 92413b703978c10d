#!/usr/bin/env python3
"""Repeat tests/test_kernels_gpu.py::test_attention_backward_headnorm_epilogue at one shape: are the forward and the two backward
flavours run-to-run deterministic, and where do the flavours differ?   python tools/probes/hn_epi_repeat.py [reps]"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd import lib as kk
from oracle import kokoro_oracle as O

B, h, Sq, Sk, causal, rope, p = 2, 4, 200, 200, 1, 1, 0.0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = lambda t: t.cuda()
g = torch.Generator().manual_seed(B * Sq + Sk + causal)
H = h * 64
dt = torch.bfloat16
raw_q = dev(torch.randn(B * Sq, H, generator=g)).to(dt)
raw_kv = dev(torch.randn(B * Sk, 2 * H, generator=g)).to(dt)
gains = [dev(1.0 + 0.2 * torch.randn(64, generator=g)) for _ in range(3)]
c, s = (dev(t) for t in O.rope_tables(max(Sq, Sk), 64))
q_n, kv_n = torch.empty_like(raw_q), torch.empty_like(raw_kv)
kk.call("kk_headnorm_rope_fwd", raw_q, H, q_n, H, B * Sq, h, Sq, 1, gains[0], None, None, 1, c, s, 1)
kk.call("kk_headnorm_rope_fwd", raw_kv, 2 * H, kv_n, 2 * H, B * Sk, h, Sk, 2, gains[1], gains[2], None, 1, c, s, 1)
k_n, v_n = kv_n, kv_n[:, H:]
do = dev(torch.randn(B * Sq, H, generator=g)).to(dt)
seed = torch.tensor([77], dtype=torch.int32, device="cuda")
first = None
for r in range(reps):
    junk = torch.randn(1 << 22, device="cuda") * (1e30 if r % 2 else 1.0)      # stir the allocator's memory
    del junk
    o, lse = torch.full((B * Sq, H), float("nan"), device="cuda", dtype=dt), torch.full((B, h, Sq), float("nan"), device="cuda")
    kk.call("kk_attn_fwd", q_n, k_n, v_n, o, lse, B, h, Sq, Sk, H, 2 * H, 2 * H, H, None, causal, 0.125, seed, 5, p, 1, 1)
    delta = torch.empty(B, h, Sq, device="cuda")
    dq_n, dkv_n = torch.empty_like(raw_q), torch.empty_like(raw_kv)
    kk.call("kk_attn_bwd_dq", q_n, k_n, v_n, do, lse, delta, dq_n, B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, None, causal, 0.125, seed, 5, p, 1, 1, o, H, None)
    kk.call("kk_attn_bwd_dkv", q_n, k_n, v_n, do, lse, delta, dkv_n, dkv_n[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, 2 * H, 2 * H, None,
            causal, 0.125, seed, 5, p, 1, 1, None)
    dkv_a = torch.empty_like(raw_kv)
    dg = [torch.zeros(64, device="cuda") for _ in range(3)]
    kk.call("kk_headnorm_rope_bwd", dkv_n, 2 * H, raw_kv, 2 * H, dkv_a, 2 * H, B * Sk, h, Sk, 2, gains[1], gains[2], None, dg[1], dg[2], None, None, 1, c, s, 1)
    nbk = kk.load().kk_attn_bwd_blocks(B, h, Sk)
    pkv = torch.full((2, nbk, 64), 5.0, device="cuda")
    dkv_b = torch.full_like(raw_kv, 7.0)
    delta_b = torch.full_like(delta, float("nan"))
    nbq = kk.load().kk_attn_bwd_blocks(B, h, Sq)
    pq = torch.full((1, nbq, 64), 5.0, device="cuda")
    dq_b = torch.full_like(raw_q, 7.0)
    kk.call("kk_attn_bwd_dq", q_n, k_n, v_n, do, lse, delta_b, dq_b, B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, None, causal, 0.125, seed, 5, p,
            1, 1, o, H, kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)]))
    kk.call("kk_attn_bwd_dkv", q_n, k_n, v_n, do, lse, delta_b, dkv_b, dkv_b[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, 2 * H, 2 * H, None,
            causal, 0.125, seed, 5, p, 1, 1, kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)]))
    torch.cuda.synchronize()
    cur = dict(delta=delta.clone(), delta_b=delta_b.clone(), o=o.clone(), lse=lse.clone(), dkv_n=dkv_n.clone(), dkv_a=dkv_a.clone(), dkv_b=dkv_b.clone())
    if first is None:
        first = cur
        print("nan in o/lse:", bool(o.isnan().any()), bool(lse.isnan().any()))
    else:
        d = {k: int((cur[k].float() != first[k].float()).sum()) for k in cur}
        if any(d.values()) or r % 100 == 0:
            print(r, d, flush=True)
            for k in cur:
                if d[k]:
                    idx = (cur[k].float() != first[k].float()).nonzero()[:4].tolist()
                    print("     ", k, idx, [float(cur[k][tuple(i)]) for i in idx], [float(first[k][tuple(i)]) for i in idx])
    if int((delta != delta_b).sum()):
        print("   delta vs delta_b differ at", int((delta != delta_b).sum()), "max", float((delta - delta_b).abs().max()))
    err = (dkv_b.float() - dkv_a.float()).abs()
    bad = (err > 0.02 + 0.02 * dkv_a.float().abs()).nonzero()
    for i, j in bad.tolist()[:6]:
        print(f"   off at row {i} (batch {i // Sk} pos {i % Sk}) col {j} ({'k' if j < H else 'v'} head {(j % H) // 64} d {j % 64}): "
              f"fused {float(dkv_b[i, j]):.4f} unfused {float(dkv_a[i, j]):.4f}; |d norm| row max {float(dkv_n[i, (j // 64) * 64:(j // 64) * 64 + 64].abs().max()):.2f}")
