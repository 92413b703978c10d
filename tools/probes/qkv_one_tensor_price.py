"""What would writing ONE tensor instead of two from the q|k|v + head-norm epilogue buy?  Probe bit 128 of the
large-tile family skips the store of the raw projection (results unusable, timing valid); bit 64 skips both stores; bit 1 the epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
kk.use_library("tuning")
bf, dev, R, H, S = torch.bfloat16, "cuda", 6, 512, 512
tune = kk._tuning_hook("kk_gemm_tune16x")
cos, sin = torch.randn(S, 64, device=dev), torch.randn(S, 64, device=dev)
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
for T, parts in ((4096, 3), (8192, 3), (4096, 12), (8192, 12)):
    N, fns, keep = parts * H, [], []
    for i in range(R):
        x, w = torch.randn(T, H, device=dev).to(bf), (torch.randn(N, H, device=dev) * 0.05).to(bf)
        raw, y = torch.empty(T, N, device=dev, dtype=bf), torch.empty(T, N, device=dev, dtype=bf)
        gains = [torch.ones(64, device=dev) for _ in range(parts)]
        tab = kk.pointer_table(gains)
        keep.append((gains, tab))
        fns.append(lambda x=x, w=w, raw=raw, y=y, tab=tab, N=N: kk.call("kk_gemm_qkv_headnorm", T, parts, 8, H, x, H, w, None, raw, N, y, N, S, tab, 3 if parts == 3 else 0, cos, sin))
    row = []
    for dbg, nm in ((0, "as shipped"), (128, "raw not stored"), (64, "neither stored"), (1, "no epilogue")):
        tune(15, -1, dbg)
        row.append(f"{nm} {gtime(fns):7.2f}")
    tune(15, -1, 0)
    print(f"rows {T} parts {parts} ({'q|k|v' if parts == 3 else 'batched cross k|v'}): " + "  |  ".join(row) + "  us", flush=True)
