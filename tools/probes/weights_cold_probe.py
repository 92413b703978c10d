#!/usr/bin/env python3
"""How much of a decoder sub-layer's time is the weights (and activations) being COLD?  The self-attention sub-layer forward as its four
launches, 24 dependent sub-layers in one replayed graph, per sub-layer time with
  (a) the same weights for all 24 (hot in every XCD's L2),
  (b) a different weight set per sub-layer (Infinity-Cache resident at best),
  (c) as (b) with the Infinity Cache flushed before every replay (a 1 GB streaming fill: the weights come from HBM, as in a real step,
      whose ~7 GB of traffic per step evict them),
  (d) as (c) with a small PREFETCH launch per sub-layer, one sub-layer ahead on a second stream, that touches the next weight set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd import lib as kk
from kokoro_ruslan_amd import spec

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B, H, h, p, dpr, REPS = 8, 512, 8, 0.2, 0.05, 24
N = B * T
dev = "cuda"
g = torch.Generator().manual_seed(3)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
bf = torch.bfloat16
n1 = rnd(N, H).to(bf)
W = [(rnd(3 * H, H, sc=H ** -0.5).to(bf), rnd(H, H, sc=H ** -0.5).to(bf)) for _ in range(REPS)]
bo = rnd(H, sc=0.1)
gq, gk, gv = (1 + rnd(64, sc=0.1) for _ in range(3))
lng, lnb = 1 + rnd(H, sc=0.1), rnd(H, sc=0.1)
x_res = rnd(N, H)
cos, sin = (t.to(dev) for t in spec.rope_tables(4000, 64))
cos, sin = cos[:T], sin[:T]
seed = torch.tensor([1234], dtype=torch.int32, device=dev)
ptrs = kk.pointer_table([gq, gk, gv])
keep_bytes = kk.load().kk_attn_keep_bytes(B, h, T, T)
z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
# one activation set PER sub-layer as in the step (every layer has its own saved tensors)
O = [dict(raw=z(N, 3 * H, dt=bf), nrm=z(N, 3 * H, dt=bf), ctx=z(N, H, dt=bf), lse=z(B, h, T), keep=z(max(keep_bytes, 16), dt=torch.uint8),
          proj=z(N, H, dt=bf), x_out=z(N, H), n=z(N, H, dt=bf), mean=z(N), rstd=z(N)) for _ in range(REPS)]
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)       # 1 GB


def sublayer(o, xin, Wqkv, Wo):
    kk.call("kk_gemm_qkv_headnorm", N, 3, h, H, xin, H, Wqkv, None, o["raw"], 3 * H, o["nrm"], 3 * H, T, ptrs, 3, cos, sin)
    q, k, v = o["nrm"], o["nrm"][:, H:], o["nrm"][:, 2 * H:]
    kk.call("kk_attn_fwd_kb", q, k, v, o["ctx"], o["lse"], B, h, T, T, 3 * H, 3 * H, 3 * H, H, None, 1, 0.125, seed, 2003, p, kk.KK_MATH_BF16, 1,
            o["keep"] if keep_bytes else None)
    kk.call("kk_gemm", 0, 0, N, H, H, 1.0, o["ctx"], H, Wo, H, 0.0, o["proj"], H, bo, None, 0, 0, 0, kk.KK_MATH_BF16, 1 | 2 | 4)
    kk.call("kk_sublayer_out_fwd", o["proj"], 1, None, None, x_res, o["x_out"], lng, lnb, o["n"], 1, o["mean"], o["rstd"], N, H, T, seed,
            2000, p, 2001, 0.0, 2002, dpr)


side = torch.cuda.Stream()
sink = z(REPS, 64)


def build(distinct, prefetch, own_acts=True):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        x = n1
        for i in range(REPS):
            wq, wo = W[i] if distinct else W[0]
            o = O[i] if own_acts else O[0]
            if prefetch and i + 1 < REPS:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                      # touch the NEXT sub-layer's weights (2 MB): a 16-workgroup reduction
                    nq, no = W[i + 1]
                    torch.sum(nq.view(16, -1)[:, ::64].float(), dim=1, out=sink[i, :16])
                    torch.sum(no.view(16, -1)[:, ::64].float(), dim=1, out=sink[i, 16:32])
            sublayer(o, x, wq, wo)
            x = o["n"]
        if prefetch:
            torch.cuda.current_stream().wait_stream(side)
    return gr


def timed(gr, cold, iters=12):
    for _ in range(2):
        gr.replay()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        if cold:
            flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot * 1000.0 / (iters * REPS)


sublayer(O[0], n1, *W[0])
torch.cuda.synchronize()
for name, distinct, prefetch, cold, own in (("(a0) same weights, ONE activation set, hot", False, False, False, False),
                                            ("(a) same weights, own activations, hot", False, False, False, True),
                                            ("(b) distinct weights, Infinity-Cache warm", True, False, False, True),
                                            ("(c) distinct weights, cache flushed (HBM)", True, False, True, True),
                                            ("(d) as (c) + prefetch one sub-layer ahead", True, True, True, True)):
    gr = build(distinct, prefetch, own)
    r = [timed(gr, cold) for _ in range(3)]
    print(f"{name:48s} {sorted(r)[1]:7.2f} us per sub-layer   {['%.2f' % v for v in r]}", flush=True)
