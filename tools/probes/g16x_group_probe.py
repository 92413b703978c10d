"""Grouped weight gradients of a decoder layer / the batched cross K|V gradient under the variants of the large-tile family."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
kk.use_library("tuning")
bf, dev, R = torch.bfloat16, "cuda", 4
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
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
keep = []
def group(shapes):
    fns = []
    for i in range(R):
        probs = [((torch.randn(T, M, device=dev)).to(bf), torch.randn(T, N, device=dev).to(bf), torch.zeros(M, N, device=dev)) for M, N in shapes]
        tab = kk.wgrad_table(probs)
        keep.append((probs, tab))
        fns.append(lambda tab=tab, n=len(probs): kk.call("kk_gemm_wgrad_group", tab, n, 0, 1))
    return fns
dec = [(3 * H, H), (H, H), (H, H), (H, H), (2 * F, H), (H, F)]
for name, shapes in (("decoder layer", dec), ("cross k|v x6", [(12 * H, H)])):
    fns = group(shapes)
    fl = sum(2.0 * T * m * n for m, n in shapes)
    row = []
    for label, on in (("old 128x64", 0), ("x 4+4 per-problem XCD runs", 15 | 1024), ("x 4+4 global XCD chunks", 15), ("x 8+4 chunks", 15 | 512)):
        tune(on, -1, 0)
        t = gtime(fns)
        row.append(f"{label}: {t:6.1f} us {fl / t / 1e6:4.0f} TF")
    print(f"T={T} grouped wgrad {name}: " + " | ".join(row))
