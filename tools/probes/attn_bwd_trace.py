"""Shader-clock stamps of workgroup (0, 0) of both halves of the attention backward pair launch (KK_ATTN_DBG=256, tools flavour): where a
wave's time goes — prologue, per 64-row tile step (top, DMA waited, barrier passed), loop end, head-norm epilogue.
    python tools/probes/attn_bwd_trace.py [S] [causal] [keep_bits]"""
import os, sys, ctypes, torch
os.environ["KK_ATTN_DBG"] = "256"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
from oracle import kokoro_oracle as O
kk.use_library("tuning")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
causal = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B, h, H, P = 8, 8, 512, 0.2
bf, dev = torch.bfloat16, "cuda"
q, kv = torch.randn(B * S, H, device=dev).to(bf), torch.randn(B * S, 2 * H, device=dev).to(bf)
raw_q, raw_kv, do = torch.randn_like(q), torch.randn_like(kv), torch.randn(B * S, H, device=dev).to(bf)
o, lse = torch.empty_like(q), torch.empty(B, h, S, device=dev)
seed = torch.tensor([7], dtype=torch.int32, device=dev)
keep = torch.empty(kk.load().kk_attn_keep_bytes(B, h, S, S), dtype=torch.uint8, device=dev)
buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
kk._tuning_hook("kk_attn_trace")(ctypes.c_void_p(buf.data_ptr()))
kk.call("kk_attn_fwd_kb", q, kv, kv[:, H:], o, lse, B, h, S, S, H, 2 * H, 2 * H, H, None, causal, 0.125, seed, 5, P, 1, 1, keep)
delta = torch.empty(B, h, S, device=dev)
kk.call("kk_attn_delta", o, do, delta, B, h, S, H, H, 1)
nb = kk.load().kk_attn_bwd_blocks(B, h, S)
gains = [torch.ones(64, device=dev) for _ in range(3)]
c, s = (t.cuda() for t in O.rope_tables(S, 64))
pq, pkv = torch.zeros(1, nb, 64, device=dev), torch.zeros(2, nb, 64, device=dev)
hq = kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)])
hkv = kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)])
dq, dkv = torch.empty_like(q), torch.empty_like(kv)
a = (q, kv, kv[:, H:], do, lse, delta, dq, dkv, dkv[:, H:], B, h, S, S, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, None, causal, 0.125, seed, 5, P, 1, 1, hq, hkv)
run = (lambda: kk.call("kk_attn_bwd_kb", *a, keep)) if kb else (lambda: kk.call("kk_attn_bwd", *a))
for _ in range(3):
    buf.zero_(); run()
torch.cuda.synchronize()
print(kk.last_kernel())
t = buf.cpu().view(8, 64)
nt = (S + 63) // 64
for half, rows in (("dQ half (workgroup 0,0: the LAST 128 queries of a causal launch)", range(0, 4)), ("dK/dV half", range(4, 8))):
    t0 = min(int(t[w, 0]) for w in rows if int(t[w, 0]))
    e = lambda w, sl: (int(t[w, sl]) - t0) if int(t[w, sl]) else -1
    print(f"S={S} causal={causal} keep_bits={kb} {half}: shader clocks since the first wave's entry")
    print("wave: entry, row fragments ready || per tile: top, DMA waited, barrier passed || loop end, ring free, epilogue images landed, done")
    for w in rows:
        tiles = " | ".join(" ".join(f"{e(w, 4 + 3 * k + j):6d}" for j in range(3)) for k in range(min(nt, 18)) if e(w, 4 + 3 * k) >= 0)
        print(f"w{w}: {e(w,0):5d} {e(w,1):6d} || {tiles} || {e(w,58):6d} {e(w,59):6d} {e(w,60):6d} {e(w,61):6d}")
