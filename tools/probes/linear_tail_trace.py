"""Wall-clock stamps (100 MHz) of workgroup 0 of kk_linear_tail_fwd: start | prologue loads landed | barrier | k-loop done | tile in LDS | rows done | stores acknowledged."""
import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
kk.use_library("tuning")
bf, dev, H, S = torch.bfloat16, "cuda", 512, 512
seed = torch.tensor([7], dtype=torch.int32, device=dev)
buf = torch.zeros(64, dtype=torch.int64, device=dev)
for T, K in ((4096, 512), (8192, 512), (4096, 1536)):
    x = torch.randn(T, K, device=dev).to(bf); w = (torch.randn(H, K, device=dev) / K ** 0.5).to(bf)
    b, gam, bet = (torch.randn(H, device=dev) for _ in range(3))
    res, xo = torch.randn(T, H, device=dev), torch.empty(T, H, device=dev)
    n, mean, rstd = torch.empty(T, H, device=dev, dtype=bf), torch.empty(T, device=dev), torch.empty(T, device=dev)
    kk._tuning_hook("kk_linear_tail_trace")(ctypes.c_void_p(buf.data_ptr()))
    for it in range(3):
        buf.zero_()
        kk.call("kk_linear_tail_fwd", x, K, w, b, K, None, 1, None, None, res, xo, gam, bet, n, 1, mean, rstd, T, H, S, seed, 40, 0.1, 41, 0.0, 42, 0.05)
        torch.cuda.synchronize()
    kk._tuning_hook("kk_linear_tail_trace")(ctypes.c_void_p(0))
    st = buf.cpu().view(8, 8)
    t0 = int(st[:, 0].min())
    print(f"rows {T} K {K}: stamps in us since the first wave's start (waves 0..7)")
    for w_ in range(8):
        print("   wave", w_, "  ".join(f"{(int(v) - t0) / 100:6.2f}" for v in st[w_, :7]))
