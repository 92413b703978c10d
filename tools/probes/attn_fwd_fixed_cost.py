"""Fixed cost of the attention forward reading keep bits (attn_fwd3_q64r): one round of 512 resident workgroups at S = 256 / 512 / 1024
(B = 16 / 8 / 4, 8 heads) — launch time = a + b * (units per wave).     python tools/probes/attn_fwd_fixed_cost.py [causal]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
causal = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if os.environ.get("KK_LIBV"):
    kk.use_library(os.environ["KK_LIBV"])
h, H, P = 8, 512, 0.2
bf, dev = torch.bfloat16, "cuda"
res = {}
for S, B in ((256, 16), (512, 8), (1024, 4)):
    q, kv = torch.randn(B * S, H, device=dev).to(bf), torch.randn(B * S, 2 * H, device=dev).to(bf)
    o, lse = torch.empty_like(q), torch.empty(B, h, S, device=dev)
    seed = torch.tensor([7], dtype=torch.int32, device=dev)
    keep = torch.empty(kk.load().kk_attn_keep_bytes(B, h, S, S), dtype=torch.uint8, device=dev)
    kk.call("kk_attn_fwd_kb", q, kv, kv[:, H:], o, lse, B, h, S, S, H, 2 * H, 2 * H, H, None, causal, 0.125, seed, 5, P, 1, 1, keep)
    for name, fn in (("rb", "kk_attn_fwd_rb"), ("kb", "kk_attn_fwd_kb")):
        run = lambda: kk.call(fn, q, kv, kv[:, H:], o, lse, B, h, S, S, H, 2 * H, 2 * H, H, None, causal, 0.125, seed, 5, P, 1, 1, keep)
        for _ in range(5):
            run()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50):
                run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 20)
        res[(S, name)] = sorted(ts)[2]
        print(f"S={S:5d} B={B:2d} causal={causal} {name}: {res[(S, name)]:.2f} us  ({kk.last_kernel()})")
for name in ("rb", "kb"):
    b1 = (res[(512, name)] - res[(256, name)]) / 8
    b2 = (res[(1024, name)] - res[(512, name)]) / 16
    print(f"{name}: per 32-key step {b1:.3f} us (256 -> 512), {b2:.3f} us (512 -> 1024); fixed part {res[(512, name)] - 16 * b2:.2f} us of {res[(512, name)]:.2f} at S = 512")
