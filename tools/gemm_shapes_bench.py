#!/usr/bin/env python3
"""Every GEMM-class launch of a decoder layer at the bench shapes, replayed from a hipGraph (R launches per graph on rotating
buffers, so the host is out of the picture), with the launch's L2 -> LDS floor beside it.

    python tools/gemm_shapes_bench.py [rows=4096] [--lib tuning]        (tuning flavour: KK_G16X=0 gives the old tiles)

floor_us = bytes through the busiest CU / 46 GB/s for the tile the policy picks is not known here; the table prints the
algorithmic FLOPs, the time and the rate, which is what the two flavours are compared on."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kokoro_ruslan_amd import lib as kk

args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--lib" in sys.argv:
    kk.use_library(sys.argv[sys.argv.index("--lib") + 1])
T = int(args[0]) if args else 4096
if len(args) > 1:                        # tools flavour: large-tile family switches (kk_gemm_tune16x: on bits | 256 no loaders | 512 8 + 4 waves)
    kk._tuning_hook("kk_gemm_tune16x")(int(args[1]), -1, 0)
H, F, L, R = 512, 1536, 6, 6
bf = torch.bfloat16
dev = "cuda"


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(bf)


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps / R * 1e3


rows = []


def case(name, flops, make):
    """make(i) -> a callable that issues launch i (its own buffers)"""
    fns = [make(i) for i in range(R)]

    def run():
        for f in fns:
            f()
    t = graph_time(run)
    rows.append((name, flops, t))
    print(f"{name:34s} {flops / 1e9:7.2f} GFLOP {t:8.2f} us {flops / t / 1e6:7.0f} TFLOP/s  ({flops / t / 1e6 / 25:5.1f} % of peak)", flush=True)


seed = torch.tensor([5], dtype=torch.int32, device=dev)
S = 512 if T % 512 == 0 else T
cos = torch.randn(S, 64, device=dev)
sin = torch.randn(S, 64, device=dev)


def fwd(N, K):
    def make(i):
        x, w, y = rnd(T, K), rnd(N, K, scale=0.05), torch.empty(T, N, device=dev, dtype=bf)
        return lambda: kk.call("kk_gemm", 0, 0, T, N, K, 1.0, x, K, w, K, 0.0, y, N, None, None, 0, 0, 0, 1, 7)
    return make


def dgrad(N, K):            # dX[T, N] = dY[T, K] . W[K, N]
    def make(i):
        dy, w, dx = rnd(T, K), rnd(K, N, scale=0.05), torch.empty(T, N, device=dev, dtype=bf)
        return lambda: kk.call("kk_gemm", 0, 1, T, N, K, 1.0, dy, K, w, N, 0.0, dx, N, None, None, 0, 0, 0, 1, 7)
    return make


def qkv(parts):
    N = parts * H
    def make(i):
        x, w = rnd(T, H), rnd(N, H, scale=0.05)
        raw, y = torch.empty(T, N, device=dev, dtype=bf), torch.empty(T, N, device=dev, dtype=bf)
        gains = [torch.ones(64, device=dev) for _ in range(parts)]
        tab = kk.pointer_table(gains)
        keep.append((gains, tab))
        return lambda: kk.call("kk_gemm_qkv_headnorm", T, parts, 8, H, x, H, w, None, raw, N, y, N, S, tab, 3 if parts == 3 else 0, cos, sin)
    return make


keep = []


def glu_fwd(i):
    x, w, b = rnd(T, H), rnd(2 * F, H, scale=0.05), torch.randn(2 * F, device=dev)
    h1, g = torch.empty(T, 2 * F, device=dev, dtype=bf), torch.empty(T, F, device=dev, dtype=bf)
    return lambda: kk.call("kk_gemm_linear_glu", T, F, H, x, H, w, b, h1, g, F, seed, 13, 0.2)


def glu_bwd(i):
    dy, w, h1 = rnd(T, H), rnd(H, F, scale=0.05), rnd(T, 2 * F)
    dh = torch.empty(T, 2 * F, device=dev, dtype=bf)
    nb = kk.load().kk_gemm_dgrad_glu_blocks(T)
    part = torch.empty(nb, 2 * F, device=dev)
    return lambda: kk.call("kk_gemm_dgrad_glu", T, F, H, dy, H, w, h1, dh, part, seed, 9, 0.2)


def delta(i):
    dy, w, o = rnd(T, H), rnd(H, H, scale=0.05), rnd(T, H)
    dx, d = torch.empty(T, H, device=dev, dtype=bf), torch.empty(T // S, 8, S, device=dev)
    return lambda: kk.call("kk_gemm_dgrad_delta", T, H, H, dy, H, w, H, dx, H, o, H, d, S, 8)


def group(shapes):
    def make(i):
        probs = [(rnd(T, M), rnd(T, N), torch.zeros(M, N, device=dev)) for M, N in shapes]
        tab = kk.wgrad_table(probs)
        keep.append((probs, tab))
        return lambda: kk.call("kk_gemm_wgrad_group", tab, len(probs), 0, 1, None, None, None)
    return make


fl = lambda n, k: 2.0 * T * n * k
case("q|k|v + head norm  N=1536 K=512", fl(3 * H, H), qkv(3))
case("cross q + head norm N=512 K=512", fl(H, H), qkv(1))
case("cross k|v x6 + head norm N=6144", fl(12 * H, H), qkv(12))
case("w_o forward        N=512 K=512", fl(H, H), fwd(H, H))
case("linear1 + GLU      N=3072 K=512", fl(2 * F, H), glu_fwd)
case("linear2 forward    N=512 K=1536", fl(H, F), fwd(H, F))
case("linear2 dgrad+GLU' N=1536 K=512", fl(F, H), glu_bwd)
case("linear1 dgrad      N=512 K=3072", fl(H, 2 * F), dgrad(H, 2 * F))
case("w_o dgrad + Delta  N=512 K=512", fl(H, H), delta)
case("q|k|v dgrad        N=512 K=1536", fl(H, 3 * H), dgrad(H, 3 * H))
case("cross k|v dgrad    N=512 K=6144", fl(H, 12 * H), dgrad(H, 12 * H))
dec = [(3 * H, H), (H, H), (H, H), (H, H), (2 * F, H), (H, F)]
case("grouped wgrad, decoder layer", sum(2.0 * T * m * n for m, n in dec), group(dec))
case("grouped wgrad, cross k|v x6", 2.0 * T * 12 * H * H, group([(12 * H, H)]))
print(f"sum {sum(t for _, _, t in rows):.1f} us over the {len(rows)} launch kinds at {T} rows")
