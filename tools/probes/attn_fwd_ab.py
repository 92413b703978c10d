"""attn_fwd at the decoder shapes, graph-replayed (R launches): KK_ATTN_FWD3 = 0 (second generation) / 1 (third) in the tools flavour."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
kk.use_library("tuning")
B, h, R = 8, 8, 6
H = h * 64
bf = torch.bfloat16
seed = torch.tensor([7], dtype=torch.int32, device="cuda")
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
for S in (512, 1024, 700):
    for causal in (0, 1):
        for p in (0.2, 0.0):
            fns = []
            for i in range(R):
                qkv = torch.randn(B * S, 3 * H, device="cuda").to(bf)
                o, lse = torch.empty(B * S, H, device="cuda", dtype=bf), torch.empty(B, h, S, device="cuda")
                fns.append(lambda qkv=qkv, o=o, lse=lse: kk.call("kk_attn_fwd", qkv, qkv[:, H:], qkv[:, 2 * H:], o, lse, B, h, S, S, 3 * H, 3 * H, 3 * H, H, None, causal, 0.125, seed, 3, p, 1, 1))
            t = gtime(fns)
            fl = 4.0 * B * h * S * S * 64 * (0.5 if causal else 1.0)
            print(f"FWD3={os.environ.get('KK_ATTN_FWD3', '1')} S={S} causal={causal} p={p}: {t:6.2f} us  {fl / t / 1e6:5.0f} TFLOP/s ({fl / t / 1e6 / 25:4.1f} %)")
