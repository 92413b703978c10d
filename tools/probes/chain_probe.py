#!/usr/bin/env python3
"""Stage 0 of VERDICT r5 item 1: the decoder's self-attention sub-layer forward as ONE persistent launch with batch item b on XCD b
(csrc/kk_chain.hip) against the four launches it replaces — bit-identity of every stored tensor, time per sub-layer in a replayed
graph of 24 dependent sub-layers (the step's own situation: each reads what the previous wrote), and the per-phase clock stamps.

    python tools/probes/chain_probe.py [T] [flags]      T = 512 | 1024 frames per item (B = 8); flags bit 0 = agent-scope barrier atomics
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd import lib as kk
from kokoro_ruslan_amd import spec

if os.environ.get("KK_LIB"):
    kk.use_library(os.environ["KK_LIB"])
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B, H, h, p, dpr = 8, 512, 8, 0.2, 0.05
N = B * T
dev = "cuda"
g = torch.Generator().manual_seed(3)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
bf = torch.bfloat16
n1 = rnd(N, H).to(bf)
Wqkv = rnd(3 * H, H, sc=H ** -0.5).to(bf)
Wo = rnd(H, H, sc=H ** -0.5).to(bf)
bo = rnd(H, sc=0.1)
gq, gk, gv = (1 + rnd(64, sc=0.1) for _ in range(3))
lng, lnb = 1 + rnd(H, sc=0.1), rnd(H, sc=0.1)
x_res = rnd(N, H)
cos, sin = (t.to(dev) for t in spec.rope_tables(4000, 64))
cos, sin = cos[:T], sin[:T]
seed = torch.tensor([1234], dtype=torch.int32, device=dev)
ptrs = kk.pointer_table([gq, gk, gv])
keep_bytes = kk.load().kk_attn_keep_bytes(B, h, T, T)


def outputs():
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
    return dict(raw=z(N, 3 * H, dt=bf), nrm=z(N, 3 * H, dt=bf), ctx=z(N, H, dt=bf), lse=z(B, h, T), keep=z(max(keep_bytes, 16), dt=torch.uint8),
                proj=z(N, H, dt=bf), x_out=z(N, H), n=z(N, H, dt=bf), mean=z(N), rstd=z(N))


def sublayer(o, x_in=None):
    """the four launches exactly as engine._attn_fwd issues them (self-attention, bf16 storage, dropout on)"""
    xin = n1 if x_in is None else x_in
    kk.call("kk_gemm_qkv_headnorm", N, 3, h, H, xin, H, Wqkv, None, o["raw"], 3 * H, o["nrm"], 3 * H, T, ptrs, 3, cos, sin)
    q, k, v = o["nrm"], o["nrm"][:, H:], o["nrm"][:, 2 * H:]
    kk.call("kk_attn_fwd_kb", q, k, v, o["ctx"], o["lse"], B, h, T, T, 3 * H, 3 * H, 3 * H, H, None, 1, 0.125, seed, 2003, p, kk.KK_MATH_BF16, 1,
            o["keep"] if keep_bytes else None)
    kk.call("kk_gemm", 0, 0, N, H, H, 1.0, o["ctx"], H, Wo, H, 0.0, o["proj"], H, bo, None, 0, 0, 0, kk.KK_MATH_BF16, 1 | 2 | 4)
    kk.call("kk_sublayer_out_fwd", o["proj"], 1, None, None, x_res, o["x_out"], lng, lnb, o["n"], 1, o["mean"], o["rstd"], N, H, T, seed,
            2000, p, 2001, 0.0, 2002, dpr)


sync = torch.zeros(512, dtype=torch.int32, device=dev)
trace = torch.zeros(256 * 16, dtype=torch.int64, device=dev)


def chained(o, x_in=None, tr=None):
    kk.call("kk_chain_begin")
    try:
        sublayer(o, x_in)
    except BaseException:
        kk.call("kk_chain_abort")
        raise
    kk.call("kk_chain_launch", 0, sync, tr, FLAGS)


if len(sys.argv) > 3 and sys.argv[3] == "pmc":      # counter passes (rocprofv3 --pmc): 8 DEPENDENT sub-layers each way, eagerly, nothing else
    for fn in (sublayer, chained):
        o, x = outputs(), None
        for _ in range(8):
            fn(o, x)
            x = o["n"]
        torch.cuda.synchronize()
    sys.exit(0)
a, c = outputs(), outputs()
sublayer(a)
routes = []
chained(c, tr=trace)
torch.cuda.synchronize()
print("route:", kk.last_kernel(), " sync[0] (timeout) =", int(sync[0]), " placement mismatches =", int(sync[1]))
a2, c2 = outputs(), outputs()
sublayer(a2)
chained(c2)
torch.cuda.synchronize()
for k in a:
    same = torch.equal(a[k], c[k])
    nd = int((a[k] != c[k]).sum())
    print(f"  {k:6s} bit-identical: {same}" + ("" if same else f"   max|diff| = {float((a[k].float() - c[k].float()).abs().max()):.3e}  in {nd} of {a[k].numel()} elements")
          + f"   [four launches twice: {torch.equal(a[k], a2[k])}, chain twice: {torch.equal(c[k], c2[k])}]")
if not torch.equal(a["ctx"], c["ctx"]):
    d = (a["ctx"] != c["ctx"]).view(B, T, h, 64)
    idx = d.nonzero()
    print("  ctx differs at (b, t, head, d) e.g.", idx[:6].tolist(), " rows:", sorted(set(idx[:, 1].tolist()))[:12], " heads:", sorted(set(idx[:, 2].tolist())))
tr = trace.view(256, 16).cpu()
xcc = tr[:, 15]
print("XCC id == workgroup % 8 for", int((xcc == torch.arange(256) % 8).sum()), "of 256 workgroups")
st = (tr[:, :8] - tr[:, :1]).double() / 100.0
names = ["qkv+headnorm", "barrier", "attention (2 units)", "barrier", "w_o GEMM", "barrier", "tail rows"]
print("per-phase microseconds (median / max over the 256 workgroups):")
for i, nme in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print(f"  {nme:22s} {float(d.median()):7.2f} {float(d.max()):7.2f}")
print(f"  workgroup total        {float(st[:, 7].median()):7.2f} {float(st[:, 7].max()):7.2f}")


def timed(fn, reps=24, iters=20):
    """a graph of `reps` DEPENDENT sub-layers (each reads the previous one's LayerNorm output), replayed"""
    o = outputs()
    fn(o)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        x = None
        for _ in range(reps):
            fn(o, x)
            x = o["n"]
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / (iters * reps)


# placement inside replayed graphs: is every GROUP (workgroups w, w + 8, ...) still on one XCD when launches follow each other?
o = outputs()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(3):
        chained(o, tr=trace)
for rep in range(4):
    gr.replay()
    torch.cuda.synchronize()
    xc = trace.view(256, 16)[:, 15].cpu().view(32, 8)          # [member][group]
    one = bool((xc == xc[:1]).all())
    print(f"replay {rep}: XCC ids of groups 0..7 = {xc[0].tolist()}   every member of a group on that XCD: {one}")
res = []
for r in range(3):
    t4, t1 = timed(sublayer), timed(chained)
    res.append((t4, t1))
    print(f"round {r}: four launches {t4:7.2f} us per sub-layer   one persistent launch {t1:7.2f} us   ratio {t4 / t1:5.3f}")
print(f"B={B} T={T} flags={FLAGS}  median ratio {sorted(a_ / b_ for a_, b_ in res)[1]:.3f}   sync[0]={int(sync[0])} mismatches={int(sync[1])}")
