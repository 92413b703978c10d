"""Fixed cost of the attention backward pair launch: one round of 512 resident workgroups at S = 256 / 512 / 1024 (B = 16 / 8 / 4, 8 heads), with
and without the head-norm epilogues — launch time = a + b * (units per wave).     python tools/probes/attn_bwd_fixed_cost.py [causal]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
from kokoro_ruslan_amd.spec import rope_tables
causal = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if os.environ.get("KK_LIBV"):
    kk.use_library(os.environ["KK_LIBV"])       # a --variant build (A/B of a compile-time switch)
h, H, P = 8, 512, 0.2
bf, dev = torch.bfloat16, "cuda"
res = {}
for S, B in ((256, 16), (512, 8), (1024, 4)):
    q, kv = torch.randn(B * S, H, device=dev).to(bf), torch.randn(B * S, 2 * H, device=dev).to(bf)
    raw_q, raw_kv, do = torch.randn_like(q), torch.randn_like(kv), torch.randn(B * S, H, device=dev).to(bf)
    o, lse = torch.empty_like(q), torch.empty(B, h, S, device=dev)
    seed = torch.tensor([7], dtype=torch.int32, device=dev)
    keep = torch.empty(kk.load().kk_attn_keep_bytes(B, h, S, S), dtype=torch.uint8, device=dev)
    kk.call("kk_attn_fwd_kb", q, kv, kv[:, H:], o, lse, B, h, S, S, H, 2 * H, 2 * H, H, None, causal, 0.125, seed, 5, P, 1, 1, keep)
    delta = torch.empty(B, h, S, device=dev)
    kk.call("kk_attn_delta", o, do, delta, B, h, S, H, H, 1)
    nb = kk.load().kk_attn_bwd_blocks(B, h, S)
    gains = [torch.ones(64, device=dev) for _ in range(3)]
    c, s = (t.cuda() for t in rope_tables(S, 64))
    pq, pkv = torch.zeros(1, nb, 64, device=dev), torch.zeros(2, nb, 64, device=dev)
    hq = kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)])
    hkv = kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)])
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    for hn in (1, 0):
        a = (q, kv, kv[:, H:], do, lse, delta, dq, dkv, dkv[:, H:], B, h, S, S, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, None, causal, 0.125, seed, 5, P, 1, 1,
             hq if hn else None, hkv if hn else None)
        run = lambda: kk.call("kk_attn_bwd_kb", *a, keep)
        for _ in range(5):
            run()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50):
                run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 20)
        res[(S, hn)] = sorted(ts)[2]
        print(f"S={S:5d} B={B:2d} causal={causal} head-norm epilogues={hn}: {res[(S, hn)]:.2f} us  ({kk.last_kernel()}, {S // 32} key units per dQ wave)")
for hn in (1, 0):
    b1 = (res[(512, hn)] - res[(256, hn)]) / 8
    b2 = (res[(1024, hn)] - res[(512, hn)]) / 16
    print(f"epilogues={hn}: per unit {b1:.3f} us (256 -> 512), {b2:.3f} us (512 -> 1024); fixed part {res[(512, hn)] - 16 * b2:.2f} us of {res[(512, hn)]:.2f} at S = 512")
