"""Where the time of the fused encoder launch (kk_encoder_stack_fwd) goes: one workgroup stamps the 100 MHz clock after every
phase and after every group barrier; printed per phase kind, averaged over the layers (work = phase body of THAT workgroup,
wait = its barrier: arrival + the slowest member's work)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

B, T, P = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 64
eng = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
eng.train_dropout = True
batch = {k: v.cuda() for k, v in synthetic_batch(B, T, P, seed=1).items()}
names = ["qkv", "attn", "w_o", "tail1", "lin1", "lin2", "tail2"]
L = eng.dims.enc_layers
for wg in [int(x) for x in os.environ.get("WGS", "0,8,64,255").split(",")]:
    eng.enc_trace = torch.zeros(8192, dtype=torch.int64, device="cuda")
    eng.enc_trace_wg = wg
    eng._invalidate()
    for _ in range(3):
        eng.forward_backward(batch, backward=False)
    torch.cuda.synchronize()
    raw = eng.enc_trace.cpu().tolist()
    # sub-stamps per layer (S <= 64): qkv, w_o, lin1 one pass of (DMA issued, DMA landed, MFMAs done), lin2 two 32-row passes
    sub = [(x - raw[0]) / 100.0 for x in raw[2048:2048 + 15 * L]]
    main = [(x - raw[0]) / 100.0 for x in raw[:1 + 2 * 7 * L]]
    for nm, ph, o in [("qkv", 0, 0), ("w_o", 2, 3), ("lin1", 4, 6), ("lin2", 5, 9)]:
        rows = []
        for l in range(1, L):
            st = sub[15 * l + o: 15 * l + o + (6 if nm == "lin2" else 3)]
            t0, t1 = main[2 * (7 * l + ph)], main[1 + 2 * (7 * l + ph)]
            rows.append([st[0] - t0] + [st[i + 1] - st[i] for i in range(len(st) - 1)] + [t1 - st[-1]])
        m = [sum(r[i] for r in rows) / len(rows) for i in range(len(rows[0]))]
        print(f"   {nm:5s}: issue X DMA, wait, MFMA + barrier, (lin2: epilogue, issue, wait, MFMA,) prefetch + epilogue + stores: " + " | ".join(f"{v:5.2f}" for v in m))
    t = raw
    n = 1 + 2 * 7 * L
    t = [(x - t[0]) / 100.0 for x in t[:n]]
    print(f"workgroup {wg} (group {wg % 8}, member {wg // 8}): whole launch {t[-1]:.1f} us")
    for k, nm in enumerate(names):
        work = [t[1 + 2 * (7 * l + k)] - t[2 * (7 * l + k)] for l in range(L)]
        wait = [t[2 + 2 * (7 * l + k)] - t[1 + 2 * (7 * l + k)] for l in range(L)]
        print(f"   {nm:6s} work {sum(work) / L:6.2f} us (min {min(work):5.2f} max {max(work):5.2f})   barrier {sum(wait) / L:6.2f} us (min {min(wait):5.2f} max {max(wait):5.2f})")
