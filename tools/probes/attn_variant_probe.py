"""Forward attention of two library flavours side by side: same inputs, outputs compared, time per launch from hipGraph replays.
    PYTHONPATH=. python tools/probes/attn_variant_probe.py <variant> [S ...]     (libkokoro_hip.so against libkokoro_hip_<variant>.so)"""
import ctypes as C
import os
import sys
import torch

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "kokoro_ruslan_amd")
variant = sys.argv[1]
sizes = [int(x) for x in sys.argv[2:]] or [512, 1024]
libs = {"product": C.CDLL(os.path.join(HERE, "libkokoro_hip.so")), variant: C.CDLL(os.path.join(HERE, f"libkokoro_hip_{variant}.so"))}
P, I, L, F, U = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
for l in libs.values():
    l.kk_attn_fwd.argtypes = [P, P, P, P, P, I, I, I, I, L, L, L, L, P, I, F, P, U, F, I, I, P]
    l.kk_attn_fwd.restype = I
    l.kk_last_error.restype = C.c_char_p
B, h = 8, 8
H = h * 64
bf = torch.bfloat16
for S in sizes:
    g = torch.Generator().manual_seed(S)
    qkv = torch.randn(B * S, 3 * H, generator=g).cuda().to(bf)
    q, k, v = qkv, qkv[:, H:], qkv[:, 2 * H:]
    seed = torch.tensor([7], dtype=torch.int32, device="cuda")
    km = torch.zeros(B, S, dtype=torch.uint8, device="cuda")
    km[:, S - 37:] = 1
    for causal, mask in ((1, None), (0, km)):
        outs = {}
        for name, lib in libs.items():
            o, lse = torch.zeros(B * S, H, device="cuda", dtype=bf), torch.zeros(B, h, S, device="cuda")

            def call():
                rc = lib.kk_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, h, S, S, 3 * H, 3 * H, 3 * H, H,
                                     mask.data_ptr() if mask is not None else None, causal, 0.125, seed.data_ptr(), 3, 0.2, 1, 1,
                                     torch.cuda.current_stream().cuda_stream)
                assert rc == 0, lib.kk_last_error()
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(50):
                    call()
            gr.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                gr.replay()
            b.record()
            torch.cuda.synchronize()
            outs[name] = (o.float().clone(), lse.clone(), a.elapsed_time(b) / 500 * 1e3)
        (o0, l0, t0), (o1, l1, t1) = outs["product"], outs[variant]
        fin = torch.isfinite(l0)
        print(f"S={S} causal={causal} mask={'y' if mask is not None else 'n'}: product {t0:6.2f} us, {variant} {t1:6.2f} us ({t0 / t1:.2f}x); "
              f"max|dO| {float((o0 - o1).abs().max()):.2e}, max|dLSE| {float((l0 - l1)[fin].abs().max()):.2e}", flush=True)
