"""Where the time of the long-reduction launches goes (tools flavour): the 128x128 dgrads with K = 1536 / 3072 at 8192 rows, linear2
forward (K = 1536), and a decoder layer's grouped weight gradients (K = 8192 rows), each with the probes of g16x_body switched on one by
one: 0 full, 1 no epilogue, 5 + no MFMAs (DMA, barriers, LDS reads), 9 DMA + barriers only."""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
kk.use_library("tuning")
bf, dev, R = torch.bfloat16, "cuda", 6
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
H, F = 512, 1536
tune = kk._tuning_hook("kk_gemm_tune16x")

def gtime(fns, reps=20):
    def run():
        for f in fns: f()
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps / len(fns) * 1e3

def plain(tb, N, K):
    out = []
    for i in range(R):
        x = torch.randn(T, K, device=dev).to(bf)
        w = (torch.randn((K, N) if tb else (N, K), device=dev) * 0.05).to(bf)
        y = torch.empty(T, N, device=dev, dtype=bf)
        out.append(lambda x=x, w=w, y=y: kk.call("kk_gemm", 0, tb, T, N, K, 1.0, x, K, w, w.shape[1], 0.0, y, N, None, None, 0, 0, 1, 1, 7))
    return out

def group():
    out, keep = [], []
    shapes = [(3 * H, H), (H, H), (H, H), (H, H), (2 * F, H), (H, F)]          # dW[M, N] = dY[T, M]^T . X[T, N]: q|k|v, w_o, cross q, cross o, linear1, linear2
    for i in range(R):
        descs = (kk.KkWgradDesc * len(shapes))()
        ts = []
        for j, (M, N) in enumerate(shapes):
            dy, x, dw = torch.randn(T, M, device=dev).to(bf), torch.randn(T, N, device=dev).to(bf), torch.zeros(M, N, device=dev)
            ts.append((dy, x, dw))
            descs[j].dy, descs[j].lddy, descs[j].x, descs[j].ldx, descs[j].dw, descs[j].lddw = dy.data_ptr(), M, x.data_ptr(), N, dw.data_ptr(), N
            descs[j].M, descs[j].N, descs[j].T = M, N, T
        keep.append((descs, ts))
        out.append(lambda d=descs: kk.call("kk_gemm_wgrad_group", d, len(shapes), 0, 1))
    return out, keep

cases = [("dgrad q|k|v  8192x512x1536 (tb=1)", plain(1, H, 3 * H), 2.0 * T * H * 3 * H), ("dgrad linear1 8192x512x3072 (tb=1)", plain(1, H, 2 * F), 2.0 * T * H * 2 * F),
         ("linear2 fwd 8192x512x1536 (tb=0)", plain(0, H, F), 2.0 * T * H * F)]
gf, gkeep = group()
cases.append(("grouped wgrad, decoder layer", gf, 2.0 * T * sum(m * n for m, n in [(3 * H, H), (H, H), (H, H), (H, H), (2 * F, H), (H, F)])))
print("KK_G16X_NS4 =", os.environ.get("KK_G16X_NS4", "0"))
for name, fns, flops in cases:
    row = []
    for dbg in (0, 1, 5, 9):
        tune(15, -1, dbg | int(os.environ.get('KK_PROBE_OR', '0')))
        t = gtime(fns)
        row.append(f"dbg{dbg}: {t:7.2f}")
        if dbg == 0: full = t
    tune(15, -1, 0)
    print(f"T={T} {name:40s} " + "  ".join(row) + f"   {flops / full / 1e6:5.0f} TFLOP/s", flush=True)
