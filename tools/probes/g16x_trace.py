"""Shader-clock stamps of workgroup 0 of a large-tile head-norm launch (probe bit 32): per wave and k-step
[loop top, after the vmcnt wait, after the barrier, after the DMA issue block] + the end of the loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
kk.use_library("tuning")
bf, dev = torch.bfloat16, "cuda"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
force = int(sys.argv[2]) if len(sys.argv) > 2 else 2
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
H, S, parts = 512, 512, 3
N = parts * H
tune, trace = kk._tuning_hook("kk_gemm_tune16x"), kk._tuning_hook("kk_gemm_trace16x")
cos, sin = torch.randn(S, 64, device=dev), torch.randn(S, 64, device=dev)
x, w = torch.randn(T, H, device=dev).to(bf), (torch.randn(N, H, device=dev) * 0.05).to(bf)
raw, y = torch.empty(T, N, device=dev, dtype=bf), torch.empty(T, N, device=dev, dtype=bf)
gains = [torch.ones(64, device=dev) for _ in range(parts)]
tab = kk.pointer_table(gains)
buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
import ctypes
kind = sys.argv[4] if len(sys.argv) > 4 else "hn"
F = 1536
seed = torch.tensor([5], dtype=torch.int32, device=dev)
if kind == "hn":
    run = lambda: kk.call("kk_gemm_qkv_headnorm", T, parts, 8, H, x, H, w, None, raw, N, y, N, S, tab, 3, cos, sin)
elif kind == "glu_fwd":
    w1, b1 = (torch.randn(2 * F, H, device=dev) * 0.05).to(bf), torch.randn(2 * F, device=dev)
    h1, g = torch.empty(T, 2 * F, device=dev, dtype=bf), torch.empty(T, F, device=dev, dtype=bf)
    run = lambda: kk.call("kk_gemm_linear_glu", T, F, H, x, H, w1, b1, h1, g, F, seed, 13, 0.2)
else:
    w2, h1 = (torch.randn(H, F, device=dev) * 0.05).to(bf), torch.randn(T, 2 * F, device=dev).to(bf)
    dh = torch.empty(T, 2 * F, device=dev, dtype=bf)
    part = torch.empty(kk.load().kk_gemm_dgrad_glu_blocks(T), 2 * F, device=dev)
    run = lambda: kk.call("kk_gemm_dgrad_glu", T, F, H, x, H, w2, h1, dh, part, seed, 9, 0.2)
lwbits = int(sys.argv[5]) if len(sys.argv) > 5 else 0
tune(15 | lwbits, force, 0)
trace(ctypes.c_void_p(buf.data_ptr()))
for _ in range(3): run()
tune(15 | lwbits, force, 32 | extra)
trace(ctypes.c_void_p(buf.data_ptr()))
run(); torch.cuda.synchronize()
t = buf.cpu().view(8, 16, 4)
t0 = int(t[:, 0, 0].min())
print(f"T={T} tile {force} dbg {32 | extra}: clocks relative to the first wave's loop entry; per k-step: top / waited / barrier / issued")
for wv in range(8):
    print(f"wave {wv}: " + " | ".join(" ".join(f"{int(t[wv, k, j]) - t0:6d}" for j in range(4)) for k in range(9)))
print("epilogue: after the barrier / tile written / after the sync | head block 0 / 1 / 2 start | loop end / stores retired")
for wv in range(8):
    e = lambda k, j: int(t[wv, k, j]) - t0
    print(f"wave {wv}: {e(9,0):6d} {e(9,1):6d} {e(9,2):6d} | {e(10,0):6d} {e(11,0):6d} {e(12,0):6d} | {e(14,0):6d} {e(14,1):6d}")
