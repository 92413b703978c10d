"""kk_linear_tail_fwd (one row-owner launch) against kk_gemm + kk_sublayer_out_fwd (the two launches it replaces): hipGraph replays of a
dependent chain over 6 operand sets (each link's residual input is the previous link's stream output), us per link."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
if os.environ.get("KK_LIB"): kk.use_library(os.environ["KK_LIB"])
bf, dev, R, H, S = torch.bfloat16, "cuda", 6, 512, 512
seed = torch.tensor([7], dtype=torch.int32, device=dev)

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

def chain(T, K, ffn, fused):
    out = []
    xs = [torch.randn(T, H, device=dev) for _ in range(R + 1)]
    for i in range(R):
        x = torch.randn(T, K, device=dev).to(bf)
        w = (torch.randn(H, K, device=dev) / K ** 0.5).to(bf)
        b, gain, gam, bet = (torch.randn(H, device=dev) for _ in range(4))
        y = torch.empty(T, H, device=dev, dtype=bf)
        n, mean, rstd, rsf = torch.empty(T, H, device=dev, dtype=bf), torch.empty(T, device=dev), torch.empty(T, device=dev), torch.empty(T, device=dev)
        res, xo = xs[i], xs[i + 1]
        g_, r_ = (gain, rsf) if ffn else (None, None)
        p2 = 0.1 if ffn else 0.0
        if fused:
            out.append(lambda x=x, w=w, b=b, y=y, n=n, mean=mean, rstd=rstd, res=res, xo=xo, g_=g_, r_=r_, gam=gam, bet=bet, p2=p2:
                       kk.call("kk_linear_tail_fwd", x, K, w, b, K, y if ffn else None, 1, g_, r_, res, xo, gam, bet, n, 1, mean, rstd, T, H, S, seed,
                               40, 0.1, 41, p2, 42, 0.05))
        else:
            def two(x=x, w=w, b=b, y=y, n=n, mean=mean, rstd=rstd, res=res, xo=xo, g_=g_, r_=r_, gam=gam, bet=bet, p2=p2):
                kk.call("kk_gemm", 0, 0, T, H, K, 1.0, x, K, w, K, 0.0, y, H, b, None, 0, 0, 0, 1, 7)
                kk.call("kk_sublayer_out_fwd", y, 1, g_, r_, res, xo, gam, bet, n, 1, mean, rstd, T, H, S, seed, 40, 0.1, 41, p2, 42, 0.05)
            out.append(two)
    return out

for T in (4096, 8192, 16384):
    for K, ffn in ((512, 0), (1536, 1), (2048, 1)):
        a, b = gtime(chain(T, K, ffn, False)), gtime(chain(T, K, ffn, True))
        print(f"rows {T:6d}  K {K:5d}  {'linear2 + FFN tail' if ffn else 'w_o + attention tail':22s}  two launches {a:7.2f} us   one launch {b:7.2f} us   ({b / a:5.2f} x)", flush=True)
