"""Shader-clock stamps of workgroup 0 of attn_fwd2 (KK_ATTN_DBG=256, tools flavour): where a wave's time goes per 64-key tile."""
import os, sys, ctypes, torch
os.environ["KK_ATTN_DBG"] = "256"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
kk.use_library("tuning")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
causal = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
B, h, H = 8, 8, 512
bf, dev = torch.bfloat16, "cuda"
qkv = torch.randn(B * S, 3 * H, device=dev).to(bf)
ctx, lse = torch.empty(B * S, H, device=dev, dtype=bf), torch.empty(B, h, S, device=dev)
seed = torch.tensor([7], dtype=torch.int32, device=dev)
buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
kk._tuning_hook("kk_attn_trace")(ctypes.c_void_p(buf.data_ptr()))
run = lambda: kk.call("kk_attn_fwd", qkv, qkv[:, H:], qkv[:, 2 * H:], ctx, lse, B, h, S, S, 3 * H, 3 * H, 3 * H, H, None, causal, 0.125, seed, 3, p, 1, 1)
for _ in range(3): run()
torch.cuda.synchronize()
t = buf.cpu().view(8, 64)
t0 = int(t[:, 0].min())
e = lambda w, s: int(t[w, s]) - t0 if int(t[w, s]) != 0 else -1
print(f"S={S} causal={causal} p={p}: clocks since the first wave's entry")
print("wave: entry | DMA issued | first tiles landed+sync | q frags + first QK || per tile: top, waited, barrier, issued, unit0 done, unit1 done || loop end, drained+sync, merged, stored")
for w in range(8):
    tiles = " | ".join(" ".join(f"{e(w, 4 + 6 * k + j):6d}" for j in range(6)) for k in range(min(8, (S + 127) // 128)))
    print(f"w{w}: {e(w,0):5d} {e(w,1):6d} {e(w,2):6d} {e(w,3):6d} || {tiles} || {e(w,58):6d} {e(w,59):6d} {e(w,60):6d} {e(w,61):6d}")
