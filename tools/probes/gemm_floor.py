"""Where a K = 512 projection GEMM's time goes: the same 4096 x 512 output with K = 64 .. 1024 (1 .. 16 k-tiles), replayed
from a hipGraph (no host launch gaps) with a different weight matrix per launch (cold operands like in the step);
per-launch time = graph time / launches.  Run under rocprofv3 --kernel-trace for kernel-only durations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
bf = torch.bfloat16
T, N, R = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 24
for K in (64, 128, 256, 512, 1024):
    xs = [torch.randn(T, K, device="cuda").to(bf) for _ in range(R)]
    ws = [torch.randn(N, K, device="cuda").to(bf) for _ in range(R)]
    ys = [torch.empty(T, N, device="cuda", dtype=bf) for _ in range(R)]
    def run():
        for x, w, y in zip(xs, ws, ys):
            kk.call("kk_gemm", 0, 0, T, N, K, 1.0, x, K, w, K, 0.0, y, N, None, None, 0, 0, 0, 1, 7)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): g.replay()
    e.record(); torch.cuda.synchronize()
    print(f"K={K:5d}: {s.elapsed_time(e) / 10 / R * 1e3:6.2f} us per launch (graph, independent launches back to back)")
# dependent chain: y_{i+1} = gemm(y_i): the step's situation (each launch waits for the previous one's output)
K = 512
w = [torch.randn(K, K, device="cuda").to(bf) * 0.04 for _ in range(R)]
bufs = [torch.randn(T, K, device="cuda").to(bf) for _ in range(2)]
def chain():
    for i in range(R):
        kk.call("kk_gemm", 0, 0, T, K, K, 1.0, bufs[i & 1], K, w[i], K, 0.0, bufs[(i + 1) & 1], K, None, None, 0, 0, 0, 1, 7)
chain(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain()
for _ in range(3): g.replay()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): g.replay()
e.record(); torch.cuda.synchronize()
print(f"dependent chain K=512: {s.elapsed_time(e) / 10 / R * 1e3:6.2f} us per launch")
