#!/usr/bin/env python3
"""What does the FORK cost the main chain, apart from sharing the chip?  Main chain: 24 dependent decoder self-attention sub-layers
(4 launches each).  Side work: 72 dependent 512-row GEMMs (the text encoder's backward is made of such launches).
  (1) main chain alone, one single-stream graph
  (2) ONE graph with the side work as a forked branch (what a captured step is today)
  (3) TWO single-stream graphs launched on two streams (events between the launches, outside the graphs)
  (4) as (3), side graph launched first
Per sub-layer time of the MAIN chain (events around the main graph only) and the wall time of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd import lib as kk
from kokoro_ruslan_amd import spec

T = 512
B, H, h, p, dpr, REPS, SIDE = 8, 512, 8, 0.2, 0.05, 24, int(sys.argv[1]) if len(sys.argv) > 1 else 72
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
O = [dict(raw=z(N, 3 * H, dt=bf), nrm=z(N, 3 * H, dt=bf), ctx=z(N, H, dt=bf), lse=z(B, h, T), keep=z(max(keep_bytes, 16), dt=torch.uint8),
          proj=z(N, H, dt=bf), x_out=z(N, H), n=z(N, H, dt=bf), mean=z(N), rstd=z(N)) for _ in range(REPS)]
sx = [rnd(512, 512).to(bf), z(512, 512, dt=bf)]
sw = [rnd(512, 512, sc=512 ** -0.5).to(bf) for _ in range(8)]


def sublayer(o, xin, Wqkv, Wo):
    kk.call("kk_gemm_qkv_headnorm", N, 3, h, H, xin, H, Wqkv, None, o["raw"], 3 * H, o["nrm"], 3 * H, T, ptrs, 3, cos, sin)
    q, k, v = o["nrm"], o["nrm"][:, H:], o["nrm"][:, 2 * H:]
    kk.call("kk_attn_fwd_kb", q, k, v, o["ctx"], o["lse"], B, h, T, T, 3 * H, 3 * H, 3 * H, H, None, 1, 0.125, seed, 2003, p, kk.KK_MATH_BF16, 1,
            o["keep"] if keep_bytes else None)
    kk.call("kk_gemm", 0, 0, N, H, H, 1.0, o["ctx"], H, Wo, H, 0.0, o["proj"], H, bo, None, 0, 0, 0, kk.KK_MATH_BF16, 1 | 2 | 4)
    kk.call("kk_sublayer_out_fwd", o["proj"], 1, None, None, x_res, o["x_out"], lng, lnb, o["n"], 1, o["mean"], o["rstd"], N, H, T, seed,
            2000, p, 2001, 0.0, 2002, dpr)


def main_chain():
    x = n1
    for i in range(REPS):
        sublayer(O[i], x, *W[i])
        x = O[i]["n"]


def side_chain():
    for i in range(SIDE):
        a, b = sx[i & 1], sx[(i + 1) & 1]
        kk.call("kk_gemm", 0, 0, 512, 512, 512, 1.0, a, 512, sw[i % 8], 512, 0.0, b, 512, None, None, 0, 0, 0, kk.KK_MATH_BF16, 1 | 2 | 4)


main_chain(); side_chain()
torch.cuda.synchronize()
s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()


def capture(fn, stream):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(gr, stream=stream):
            fn()
    return gr


def forked():
    main_s = torch.cuda.current_stream()
    s_side.wait_stream(main_s)
    with torch.cuda.stream(s_side):
        side_chain()
    main_chain()
    main_s.wait_stream(s_side)


g_main, g_side, g_fork = capture(main_chain, s_main), capture(side_chain, s_side), capture(forked, s_main)


def run(kind, iters=15):
    tm, tw = 0.0, 0.0
    for it in range(iters + 2):
        torch.cuda.synchronize()
        w0, w1, m0, m1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        with torch.cuda.stream(s_main):
            w0.record(s_main)
            if kind == "alone":
                m0.record(s_main); g_main.replay(); m1.record(s_main)
            elif kind == "forked":
                m0.record(s_main); g_fork.replay(); m1.record(s_main)
            else:
                s_side.wait_event(w0)
                if kind == "two, side first":
                    with torch.cuda.stream(s_side):
                        g_side.replay()
                m0.record(s_main); g_main.replay(); m1.record(s_main)
                if kind == "two":
                    with torch.cuda.stream(s_side):
                        g_side.replay()
                s_main.wait_stream(s_side)
            w1.record(s_main)
        torch.cuda.synchronize()
        if it >= 2:
            tm += m0.elapsed_time(m1)
            tw += w0.elapsed_time(w1)
    return tm * 1000 / (iters * REPS), tw * 1000 / iters


for kind in ("alone", "forked", "two", "two, side first", "alone"):
    m, w = run(kind)
    print(f"{kind:18s} main chain {m:7.2f} us per sub-layer    wall {w:8.1f} us", flush=True)
with torch.cuda.stream(s_side):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s_side); g_side.replay(); e1.record(s_side)
torch.cuda.synchronize()
print(f"side chain alone: {e0.elapsed_time(e1) * 1000:.1f} us for {SIDE} launches")

# ---- the price of cutting ONE stream's chain into several graphs (segment boundaries of a multi-graph step)
def chain_part(i0, i1):
    def fn():
        x = n1 if i0 == 0 else O[i0 - 1]["n"]
        for i in range(i0, i1):
            sublayer(O[i], x, *W[i])
            x = O[i]["n"]
    return fn


for parts in (1, 6, 24):
    per = REPS // parts
    gs = [capture(chain_part(k * per, (k + 1) * per), s_main) for k in range(parts)]
    tot = 0.0
    for it in range(12):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s_main):
            e0.record(s_main)
            for gg in gs:
                gg.replay()
            e1.record(s_main)
        torch.cuda.synchronize()
        if it >= 2:
            tot += e0.elapsed_time(e1)
    print(f"main chain as {parts:2d} graph(s) on one stream: {tot * 1000 / (10 * REPS):7.2f} us per sub-layer  ({tot * 100:.1f} us per pass)", flush=True)

# ---- two streams, each chain cut into 6 graphs, launched ALTERNATELY in host order (what a program of per-stream graphs does), with and
# without events between the launches
def side_part(i0, i1):
    def fn():
        for i in range(i0, i1):
            a, b = sx[i & 1], sx[(i + 1) & 1]
            kk.call("kk_gemm", 0, 0, 512, 512, 512, 1.0, a, 512, sw[i % 8], 512, 0.0, b, 512, None, None, 0, 0, 0, kk.KK_MATH_BF16, 1 | 2 | 4)
    return fn


mg = [capture(chain_part(k * 4, (k + 1) * 4), s_main) for k in range(6)]
sg = [capture(side_part(k * 12, (k + 1) * 12), s_side) for k in range(6)]
for variant in ("main only", "alternating, no events", "alternating, side waits for a main event each time", "all main then all side"):
    tm = tw = 0.0
    for it in range(12):
        torch.cuda.synchronize()
        w0, w1, m0, m1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        with torch.cuda.stream(s_main):
            w0.record(s_main)
            s_side.wait_event(w0)
            m0.record(s_main)
            if variant == "all main then all side":
                for k in range(6):
                    mg[k].replay()
                m1.record(s_main)
                with torch.cuda.stream(s_side):
                    for k in range(6):
                        sg[k].replay()
            else:
                for k in range(6):
                    mg[k].replay()
                    if variant != "main only":
                        if "event" in variant:
                            ev = torch.cuda.Event()
                            ev.record(s_main)
                            s_side.wait_event(ev)
                        with torch.cuda.stream(s_side):
                            sg[k].replay()
                m1.record(s_main)
            s_main.wait_stream(s_side)
            w1.record(s_main)
        torch.cuda.synchronize()
        if it >= 2:
            tm += m0.elapsed_time(m1)
            tw += w0.elapsed_time(w1)
    print(f"6 + 6 graphs, {variant:52s} main chain {tm * 1000 / (10 * REPS):7.2f} us per sub-layer   wall {tw * 100:8.1f} us", flush=True)
