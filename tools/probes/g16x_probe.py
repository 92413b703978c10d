"""Where the time of a large-tile launch goes: q|k|v + head norm (and the plain GEMM of the same shape) under every tile of the
family with the k-loop / epilogue probes of g16x_body switched on one by one (tools flavour of the library)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
kk.use_library("tuning")
bf, dev, R = torch.bfloat16, "cuda", 6
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H, S = 512, 512
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

def qkv_fns(parts):
    N, out = parts * H, []
    for i in range(R):
        x, w = torch.randn(T, H, device=dev).to(bf), (torch.randn(N, H, device=dev) * 0.05).to(bf)
        raw, y = torch.empty(T, N, device=dev, dtype=bf), torch.empty(T, N, device=dev, dtype=bf)
        gains = [torch.ones(64, device=dev) for _ in range(parts)]
        tab = kk.pointer_table(gains)
        out.append((lambda x=x, w=w, raw=raw, y=y, tab=tab, N=N: kk.call("kk_gemm_qkv_headnorm", T, parts, 8, H, x, H, w, None, raw, N, y, N, S, tab, 3 if parts == 3 else 0, cos, sin), (gains, tab)))
    return [f for f, _ in out], out

def plain_fns(N, K):
    out = []
    for i in range(R):
        x, w, y = torch.randn(T, K, device=dev).to(bf), (torch.randn(N, K, device=dev) * 0.05).to(bf), torch.empty(T, N, device=dev, dtype=bf)
        out.append(lambda x=x, w=w, y=y: kk.call("kk_gemm", 0, 0, T, N, K, 1.0, x, K, w, K, 0.0, y, N, None, None, 0, 0, 0, 1, 7))
    return out

names = {-1: "cost", 0: "128x128", 1: "256x128", 2: "128x192", 3: "256x192"}
for parts in (3, 12):
    fns, keep = qkv_fns(parts)
    tune(0, -1, 0)
    print(f"T={T} head-norm parts={parts}: old tiles {gtime(fns):7.2f} us")
    for force in (0, 1, 2, 3):
        for lw in (2, 1, 0):
            if lw == 2 and force != 2: continue
            row = []
            for dbg in (0, 1, 5, 9):
                tune(15 | {0: 256, 1: 0, 2: 512}[lw], force, dbg)
                row.append(f"dbg{dbg}: {gtime(fns):7.2f}")
            print(f"   {names[force]:8s} loaders={lw} " + "  ".join(row) + "   (0 full, 1 no epilogue, 5 + no MFMA, 9 DMA only)")
fns = plain_fns(3 * H, H)
tune(0, -1, 0)
print(f"T={T} plain N=1536 K=512: old tiles {gtime(fns):7.2f} us")
for force in (0, 1):
    for lw in (1, 0):
        row = []
        for dbg in (0, 1, 5, 9):
            tune(15 | (0 if lw else 256), force, dbg)
            row.append(f"dbg{dbg}: {gtime(fns):7.2f}")
        print(f"   {names[force]:8s} loaders={lw} " + "  ".join(row))
