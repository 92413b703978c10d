#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s of the full Kokoro train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic padded batch already resident in HBM: forward, 6 losses,
full backward, (N>1: gradient SUM all-reduce over RCCL), pre-clip + global clip, fused AdamW + EMA, weight-norm
projection.  Workload at every N: BASELINE.json configs[1]'s shape per GPU (8 x 512 mel frames, 64 phonemes,
default 49.4 M-parameter model, bf16 MFMA arithmetic with fp32 accumulate/master weights) — weak scaling.
Rank 0 prints ONE JSON line (contract in the task description) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def train_flops(B: int, T: int, P: int, H=512, F=1536, Le=6, Ld=6, M=80, Fv=256) -> float:
    """Algorithmic FLOPs of one train step on a padded B x T x P batch — SURVEY §8d's convention, verbatim: 1 MAC = 2 FLOP, matmul-
    class work only, causal self-attention = lower triangle, padded positions count, training = 3 x forward, NO credit for
    recomputation.  (870.1 GFLOP at 8x512x64, 1974.4 GFLOP at 8x1024x128 — asserted in tests/test_bench_table.py.)"""
    Cv = 3 * H * Fv + 3 * Fv * Fv + Fv
    macs = (B * P * (Le * (4 * H * H + 3 * H * F) + Le * 2 * P * H + Cv)
            + B * T * (Ld * (8 * H * H + 3 * H * F) + Ld * (2 * T * H + (T + 1) * H) + 2 * Cv + 2 * M * H + H))
    return 6.0 * macs

GEMM_ROLE = {(0, 0): "X.W^T fwd", (0, 1): "dY.W dgrad", (1, 1): "dY^T.X wgrad", (1, 0): "X^T.W"}
PMC_ROUNDS = ("r06", "r05", "r04", "r03", "r02h", "r02g")       # profiles/<round>_pmc_hbm_traffic_BxTxP.json, newest first (tools/rocprof_pmc.sh: separate --pmc passes)


CUS = 256                                  # compute units; set_device_cus() replaces it with the device's count (the library's tile policy reads the same attribute)
L2_LDS_BPC = 54.0                          # bytes per clock and CU the L2 -> LDS DMA delivers with 64 KB in flight (profiles/r04_stream_rate_probe.txt)
SCLK_HZ = 2.4e9


def set_device_cus(n: int) -> None:
    global CUS
    if n > 0:
        CUS = int(n)


def _cd(x, y):
    return -(-x // y)


def _cost(tiles, bm, bn):
    return _cd(tiles, CUS) * (bm + bn)


def x_tile(kind, M, N, K):
    """(BM, BN, name) of the large-tile family's kernel for this launch, or None when kk_gemm16.hip keeps its 64x64 / 128x64 tiles
    (mirrors g16_cost / the thresholds of kk_gemm16.hip: plain and Delta launches need K >= 1024, head-norm ones N >= 1024)."""
    if kind in ("plain0", "plain1", "delta"):                    # X.W^T / dY.W; Delta = dY.W on 128x128 only
        if K < 1024:
            return None
        bm0, bn0 = (128, 64) if _cd(M, 128) * _cd(N, 64) >= 128 else (64, 64)
        best, bc = None, _cost(_cd(M, bm0) * _cd(N, bn0), bm0, bn0)
        for bm, bn in ((128, 128),) if kind == "delta" else ((128, 128), (256, 128)):
            c = _cost(_cd(M, bm) * _cd(N, bn), bm, bn)
            if c < bc:
                best, bc = (bm, bn), c
        if best is None:
            return None
        tb = "false" if kind == "plain0" else "true"
        return best + (f"g16x_kernel<false,{tb},{best[0]},{best[1]},3,0,2,2,4>",)
    if kind == "hn":
        if N < 1024 or K < 192:
            return None
        bm0, bn0 = (128, 64) if _cd(M, 128) * _cd(N, 64) >= 128 else (64, 64)
        best, bc = None, _cost(_cd(M, bm0) * _cd(N, bn0), bm0, bn0)
        for bm, bn in ((128, 128), (128, 192), (256, 128), (256, 192)):
            c = _cost(_cd(M, bm) * _cd(N, bn), bm, bn)
            if c < bc:
                best, bc = (bm, bn), c
        if best is None:
            return None
        cfg = "2,3,4,2,0" if best == (256, 192) else "3,3,2,2,4"
        return best + (f"g16x_kernel<false,false,{best[0]},{best[1]},{cfg}>",)
    if kind == "glu_fwd":                                        # N = F: 256 rows x (96 + 96) columns against 64 x (64 + 64)
        if K < 192 or _cost(_cd(M, 256) * _cd(N, 96), 256, 192) >= _cost(_cd(M, 64) * _cd(N, 64), 64, 128):
            return None
        return (256, 192, "g16x_kernel<false,false,256,192,2,2,8,1,0>")
    if kind == "glu_bwd":
        if K < 192 or _cost(_cd(M, 128) * _cd(N, 192), 128, 192) >= _cost(_cd(M, 128) * _cd(N, 64), 128, 64):
            return None
        return (128, 192, "g16x_kernel<false,true,128,192,3,1,2,2,4>")
    return None


def lds_floor_us(macs, bm, bn):
    """The launch's L2 -> LDS floor: 2 B x MACs x (1 / BM + 1 / BN) through every CU at the rate the DMA path DELIVERS when it is all a
    kernel does (54 B/clk/CU at 2.4 GHz = 130 GB/s per CU: the round-4 probe; rounds 3-4 priced 46 GB/s, which the k-loops achieve —
    VERDICT r4: a floor must be what the path can do, not what the kernel does)."""
    return 2.0 * macs * (1.0 / bm + 1.0 / bn) / (CUS * L2_LDS_BPC * SCLK_HZ) * 1e6


def gemm_symbol(ta, tb, M, N, K, math_bf16, dtypes):
    """(kernel instantiation kk_gemm launches for this call, BM, BN) — mirrors the dispatch in kk_gemm.hip / kk_gemm16.hip."""
    b = lambda v: "true" if v else "false"
    if math_bf16 and (dtypes & 3) == 3 and (K % 64 == 0 or (ta and tb)):      # DMA-staged bf16 x bf16 core
        cd = _cd
        if not ta:
            x = x_tile("plain1" if tb else "plain0", M, N, K)
            if x is not None:
                return f"{x[2]} ({GEMM_ROLE[(ta, tb)]}, {x[0]}x{x[1]} tiles, loader waves)", x[0], x[1]
        if cd(M, 128) * cd(N, 64) >= 128 and os.environ.get("KK_G16_W8", "3") != "0":      # eight waves on 128x64 tiles
            return f"gemm16_kernel_w8<{b(ta)},{b(tb)},{3 if cd(K, 64) >= 3 else 2}> ({GEMM_ROLE[(ta, tb)]}, 128x64 tiles)", 128, 64
        tiles, ktiles = cd(M, 64) * cd(N, 64), cd(K, 64)
        splits = 1
        if tiles * 2 <= 128 and not (dtypes & 4):                                          # (g16_split_target, kk_gemm16.hip)
            splits = max(1, min(cd(128, tiles), max(ktiles // 2, 1)))
        ns = 2 if ktiles // splits < 3 else 3
        return f"gemm16_kernel<{b(ta)},{b(tb)},64,64,{ns}> ({GEMM_ROLE[(ta, tb)]})", 64, 64
    tile = 128 if -(-M // 128) * -(-N // 128) >= 512 else 64
    return f"gemm_kernel<{b(ta)},{b(tb)},{b(math_bf16)},{tile}> ({GEMM_ROLE[(ta, tb)]})", tile, tile


def kernel_table(records, math_bf16: bool):
    """Aggregate per-launch event timings into {kernel: {launches, ms, flops, bytes}}."""
    agg = {}
    for name, sc, ms in records:
        flops = byts = floor = 0.0
        key = name
        if name == "kk_gemm":
            ta, tb, M, N, K = (int(x) for x in sc[:5])
            dt = int(sc[-1])               # storage bits: A, B, C bf16
            key, bm_, bn_ = gemm_symbol(ta, tb, M, N, K, math_bf16, dt)
            flops = 2.0 * M * N * K
            floor = lds_floor_us(M * N * K, bm_, bn_)
            byts = (2.0 if dt & 1 else 4.0) * M * K + (2.0 if dt & 2 else 4.0) * N * K + (2.0 if dt & 4 else 4.0) * M * N
        elif name == "kk_gemm_wgrad_group":             # (n, split_k, overwrite, then M, N, T of every problem); 128x64 tiles, see kk_gemm16.hip
            dims = [int(x) for x in sc[3:]]
            overwrite = int(sc[2])
            probs = [dims[i:i + 3] for i in range(0, len(dims), 3)]
            key, bm_, bn_ = "gemm16_group_kernel<true,true,128,64,2,8> (a layer's dY^T.X wgrads, one launch)", 128, 64
            # (mirrors kk_gemm16_wgrad_group: 128x128 tiles of the large-tile family for long reductions that fill the chip without k-slices)
            t_old = sum(_cd(M, 128) * _cd(N, 64) for M, N, _ in probs)
            t_new = sum(_cd(M, 128) * _cd(N, 128) for M, N, _ in probs)
            if min(T_ for _, _, T_ in probs) >= 1024 and (overwrite or t_old * 2 > 384) and _cost(t_new, 128, 128) < _cost(t_old, 128, 64):
                key, bm_, bn_ = "g16x_group_kernel<true,true,128,128,3,2,2,4> (a layer's dY^T.X wgrads, one launch, 128x128 tiles, loader waves)", 128, 128
            flops = sum(2.0 * M * N * T_ for M, N, T_ in probs)
            floor = lds_floor_us(sum(M * N * T_ for M, N, T_ in probs), bm_, bn_)
            byts = sum(2.0 * T_ * (M + N) + (4.0 if overwrite else 8.0) * M * N for M, N, T_ in probs)   # bf16 operands, fp32 dW (read +) written
        elif name == "kk_gemm_linear_glu":              # (T, F, K, ...): h1 = x.W1^T (2F columns) + gate
            T_, F_, K_ = (int(x) for x in sc[:3])
            key, bm_, bn_ = "gemm16_kernel<false,false,64,64,2,2> (X.W1^T + GLU gate epilogue)", 64, 128
            x = x_tile("glu_fwd", T_, F_, K_)
            if x is not None:
                key, bm_, bn_ = f"{x[2]} (X.W1^T + GLU gate epilogue, 256 x (96 + 96) tiles)", x[0], x[1]
            flops, byts = 2.0 * T_ * 2 * F_ * K_, 2.0 * (T_ * K_ + 2 * F_ * K_ + 3 * T_ * F_)
            floor = lds_floor_us(T_ * 2 * F_ * K_, bm_, bn_)
        elif name == "kk_gemm_dgrad_glu":               # (T, F, H, ...): dG = dY.W2 + gate backward
            T_, F_, H_ = (int(x) for x in sc[:3])
            key = f"gemm16_kernel<false,true,64,64,{2 if H_ // 64 < 3 else 3},1> (dY.W2 + GLU backward epilogue)"
            if H_ // 64 >= 3 and -(-T_ // 128) * -(-F_ // 64) >= 128 and int(os.environ.get("KK_G16_W8_GLU", "1")) & 1:
                key = "gemm16_kernel_w8_glu (dY.W2 + GLU backward epilogue, 128x64 tiles)"
            bm_, bn_ = (128, 64) if "w8" in key else (64, 64)
            x = x_tile("glu_bwd", T_, F_, H_)
            if x is not None:
                key, bm_, bn_ = f"{x[2]} (dY.W2 + GLU backward epilogue, 128x192 tiles, loader waves)", x[0], x[1]
            flops, byts = 2.0 * T_ * F_ * H_, 2.0 * (T_ * H_ + F_ * H_ + 4 * T_ * F_)
            floor = lds_floor_us(T_ * F_ * H_, bm_, bn_)
        elif name == "kk_gemm_qkv_headnorm":            # (T, parts, heads, K, ...)
            T_, parts, heads, K_ = (int(x) for x in sc[:4])
            N_ = parts * heads * 64
            key = f"gemm16_kernel<false,false,64,64,{2 if K_ // 64 < 3 else 3},3> (q|k|v projection + head-norm epilogue)"
            if K_ // 64 >= 3 and -(-T_ // 128) * -(-N_ // 64) >= 128 and os.environ.get("KK_G16_W8_HN", "1") != "0":
                key = "gemm16_kernel_w8_hn (q|k|v projection + head-norm epilogue, 128x64 tiles)"
            bm_, bn_ = (128, 64) if "w8" in key else (64, 64)
            x = x_tile("hn", T_, N_, K_)
            if x is not None:
                key, bm_, bn_ = f"{x[2]} (q|k|v projection + head-norm epilogue, {x[0]}x{x[1]} tiles)", x[0], x[1]
            flops, byts = 2.0 * T_ * N_ * K_, 2.0 * (T_ * K_ + N_ * K_ + 2 * T_ * N_)
            floor = lds_floor_us(T_ * N_ * K_, bm_, bn_)
        elif name == "kk_gemm_dgrad_delta":             # (M, N, K, ...): the w_o dgrad with the Delta epilogue, same instantiation as kk_gemm's
            M, N, K = (int(x) for x in sc[:3])
            ta, tb = 0, 1
            key, bm_, bn_ = gemm_symbol(0, 1, M, N, K, math_bf16, 7)
            x = x_tile("delta", M, N, K)
            if x is not None:
                key, bm_, bn_ = f"{x[2]} (dY.W dgrad + Delta epilogue, 128x128 tiles, loader waves)", x[0], x[1]
            flops, byts = 2.0 * M * N * K, 2.0 * (M * K + N * K + 2 * M * N)
            floor = lds_floor_us(M * N * K, bm_, bn_)
        elif name == "kk_linear_tail_fwd":              # (ldx, K, y_round, n_bf16, rows, H, S, ...): w_o projection + sub-layer tail, one row-owner launch
            K, rows_, H_ = int(sc[1]), int(sc[4]), int(sc[5])
            key = "linear_tail_fwd_kernel (attention output projection + dropout / residual / LayerNorm tail, 32 rows x 512 columns per workgroup)"
            flops = 2.0 * rows_ * H_ * K
            byts = 2.0 * rows_ * K + 2.0 * H_ * K + rows_ * H_ * (4.0 + 4.0 + 2.0)      # x, W once; residual in, stream out, LayerNorm out
        elif name in ("kk_attn_bwd", "kk_attn_bwd_kb"):  # (B, h, Sq, Sk, 7 row strides, causal, scale, site, p_drop, math, io_bf16); _kb: + the keep-bit buffer (a tensor)
            B, h, Sq, Sk = (int(x) for x in sc[:4])
            causal = int(sc[-6])
            kb = name == "kk_attn_bwd_kb" and Sk > 128 and float(sc[-3]) > 0.0      # (mirrors kk_attn_bwd_kb: the launches whose forward stored bits)
            # (the text encoder's one-tile launches, S <= 64, are a different regime from the decoder's: listed apart)
            key = ("attn_bwd_pair3k_kernel (dQ | dK, dV in one launch, two workgroups per CU, stored dropout keep bits)" if kb else
                   "attn_bwd_pair3_kernel (dQ | dK, dV in one launch, two workgroups per CU)" + (", one-tile sequences (text encoder)" if max(Sq, Sk) <= 64 else ""))
            # §8d: training = 3 x forward, no credit for recomputation — the backward of the forward's 2 matmuls is 4 (dV, dP, dQ, dK);
            # the launch EXECUTES 7 (S and dP are computed by both halves): executed work is not algorithmic work
            flops = 4 * 2.0 * B * h * Sq * Sk * 64 * (0.5 if causal else 1.0)
            byts = (2.0 if int(sc[-1]) else 4.0) * B * h * 64 * (4 * Sq + 4 * Sk) + (2 * B * h * Sq * Sk / 8.0 * (0.5 if causal else 1.0) if kb else 0.0)   # (both halves read the bits)
        elif name == "kk_attn_bwd_ws":                  # (... as kk_attn_bwd ..., ws_bytes): two kernels behind one entry point, event-timed together
            B, h, Sq, Sk = (int(x) for x in sc[:4])
            causal = int(sc[-7])
            key = "attn_bwd_dkv3s_kernel + attn_bwd_dqpass_kernel (dK, dV and the dS tiles, then dQ = dS.K)"
            flops = 4 * 2.0 * B * h * Sq * Sk * 64 * (0.5 if causal else 1.0)        # executes 5: S is the only recomputation left
            byts = 2.0 * B * h * 64 * (4 * Sq + 4 * Sk) + 2 * 2.0 * B * h * Sq * Sk * (0.5 if causal else 1.0)   # + dS written and read once
        elif name in ("kk_attn_fwd", "kk_attn_fwd_kb", "kk_attn_fwd_rb", "kk_attn_bwd_dq", "kk_attn_bwd_dkv"):
            kb_fwd = name in ("kk_attn_fwd_kb", "kk_attn_fwd_rb")   # (the same kernels, also storing — or, _rb, reading — one bit per score)
            rb_fwd = name == "kk_attn_fwd_rb"
            name = "kk_attn_fwd" if kb_fwd else name
            B, h, Sq, Sk = (int(x) for x in sc[:4])
            off = 1 if name == "kk_attn_bwd_dq" else 0       # (..., causal, scale, site, p_drop, math, io_bf16[, ldo])
            causal = int(sc[-6 - off])
            mm = {"kk_attn_fwd": 2, "kk_attn_bwd_dq": 1, "kk_attn_bwd_dkv": 3}[name]   # ALGORITHMIC matmuls of Sq x Sk x 64 (dQ | dV, dP, dK)
            v2 = int(sc[-1 - off]) and Sk > (64 if name == "kk_attn_fwd" else 32)       # (mirrors the dispatch in kk_attn.hip)
            fwd_name = "attn_fwd_kernel (flash forward, first generation)"
            if v2:                                               # (mirrors kk_attn_fwd: the third generation above 128 keys, 128- or 64-query blocks)
                fwd_name = ("attn_fwd2_kernel (flash forward, DMA-staged)" if Sk <= 128 else
                            "attn_fwd3_q128_kernel (flash forward, 2 workgroups per CU, 128-query blocks x 2 key slots)" if _cd(Sq, 128) * B * h >= 512 else
                            "attn_fwd3_q64_kernel (flash forward, 2 workgroups per CU, 64-query blocks x 4 key slots)")
            if rb_fwd:
                fwd_name = fwd_name.replace("_kernel (flash forward", "r_kernel (flash forward READING the keep bits of kk_attn_keep_gen")
            key = {"kk_attn_fwd": fwd_name,
                   "kk_attn_bwd_dq": "attn_bwd_dq3_kernel (dQ)" if v2 else "attn_bwd_dq_kernel (dQ, first generation)",
                   "kk_attn_bwd_dkv": "attn_bwd_dkv3_kernel (dK, dV)" if v2 else "attn_bwd_dkv_kernel (dK, dV, first generation)"}[name]
            flops = mm * 2.0 * B * h * Sq * Sk * 64 * (0.5 if causal else 1.0)
            byts = (2.0 if int(sc[-1 - off]) else 4.0) * B * h * 64 * (2 * Sq + 2 * Sk) + (B * h * Sq * Sk / 8.0 * (0.5 if causal else 1.0) if kb_fwd else 0.0)
        keys = [key]
        if name in ("kk_gemm", "kk_gemm_dgrad_delta"):
            keys.append(f"  shape ta={ta} tb={tb} M={M} N={N} K={K}")
        elif name.startswith("kk_attn_") and name not in ("kk_attn_delta", "kk_attn_keep_gen"):
            keys.append(f"  shape {name} B={B} h={h} Sq={Sq} Sk={Sk} causal={causal}")
        for kq in keys:
            a = agg.setdefault(kq, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "floor_us": 0.0})
            a["launches"] += 1
            a["ms"] += ms
            a["flops"] += flops
            a["bytes"] += byts
            a["floor_us"] += floor
    return agg


def oracle_steps(B, T, P, device, dropout, steps, warmup=1):
    """[seconds per step] of the oracle's full train step (forward + losses + backward + pre-clip/clip/AdamW/EMA, fp32, no
    recompute) on `device`, after `warmup` untimed steps."""
    from oracle import kokoro_oracle as O
    d, hp = O.ModelDims(), O.StepHyper()
    dev = torch.device(device)
    Pm = {n: p.to(dev) for n, p in O.init_params(d, 0).items()}
    Bf = {n: b.to(dev) for n, b in O.make_buffers(d).items()}
    ema = {n: p.clone() for n, p in Pm.items()}
    st = O.OptState()
    batch = {k: v.to(dev) for k, v in O.synthetic_batch(B, T, P, d, seed=1234).items()}
    drop = O.DropCfg(on=True) if dropout else None
    times = []
    for i in range(warmup + steps):
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        G, _, _ = O.grads_of(Pm, Bf, batch, d, hp, drop=drop)
        O.optimizer_step(Pm, G, st, hp, hp.learning_rate, hp.max_grad_norm, ema, None)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if i >= warmup:
            times.append(time.perf_counter() - t0)
    return times


def cpu_baseline(B, T, P, steps=3, full=False):
    """The oracle (CPU restatement of the reference, proven equal to it by tests/golden/make_golden.py) timed on this box's
    host cores: a bounded sample — 1 warm-up + `steps` timed steps of the bench batch with dropout off, then ONE timed step with
    the reference's training-time dropout on (SURVEY §8d asks for both: half of the reference's CPU step is dropout-mask
    generation).  `full` adds the 8x1024x128 shape.  Reported baseline, not the target."""
    def entry(b, t, p, dropout, n, warm):
        ts = oracle_steps(b, t, p, "cpu", dropout, n, warmup=warm)
        mean = sum(ts) / len(ts)
        return {"value": round(b * t / mean, 1), "s_per_step": [round(x, 2) for x in ts], "dropout": "on" if dropout else "off",
                "shape": f"{b}x{t}x{p}"}
    base = entry(B, T, P, False, steps, 1)
    drop = entry(B, T, P, True, 1, 0)                    # (the process is warm: same graph of ops plus the mask draws)
    out = {"value": base["value"], "unit": "mel-frames/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"1 warm-up + {steps} timed full train steps of the {B}x{T} batch (P={P}), fp32, dropout off, mean: "
                     f"{base['s_per_step']} s/step; then 1 timed step with the reference's dropout / stochastic depth on: "
                     f"{drop['s_per_step']} s/step",
           "value_dropout_on": drop["value"], "variants": [base, drop]}
    if full and (B, T, P) != (8, 1024, 128):
        out["variants"] += [entry(8, 1024, 128, False, 1, 1), entry(8, 1024, 128, True, 1, 0)]
    return out


def torch_rocm_baseline(B, T, P, steps=3):
    """The honest same-hardware comparator: the same restatement run by stock PyTorch-ROCm ops on this GPU
    (eager, fp32, dropout on like the timed engine step; the reference's own CUDA path would add ~1000 host syncs per step)."""
    ts = oracle_steps(B, T, P, "cuda", True, steps, warmup=2)
    mean = sum(ts) / len(ts)
    return {"value": round(B * T / mean, 1), "unit": "mel-frames/s", "kind": "port on cuda (stock PyTorch-ROCm ops, eager, fp32, dropout on)",
            "ms_per_step": [round(x * 1e3, 1) for x in ts], "torch": torch.__version__}


def pmc_entry(B, T, P, math, kernel):
    """(HBM bytes per launch, MFMA-busy fraction, method note, file) of `kernel` from the newest committed PMC summary of this workload
    shape (profiles/<round>_pmc_hbm_traffic_BxTxP.json, written by tools/rocprof_pmc.sh: separate --pmc passes)."""
    for rnd in PMC_ROUNDS:
        f = os.path.join(ROOT, "profiles", f"{rnd}_pmc_hbm_traffic_{B}x{T}x{P}.json")
        if not os.path.exists(f):
            continue
        pmc = json.load(open(f))
        parts = [pmc.get("kernels", {}).get(sym) for sym in kernel.split(" (")[0].split(" + ")]      # (an entry point of two kernels: both)
        if all(parts) and pmc.get("workload") == [B, T, P, math]:
            busy = parts[0].get("mfma_busy") if len(parts) == 1 else None
            return (round(sum(e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"] for e in parts)), busy, pmc.get("method"),
                    os.path.basename(f))
    return None, None, None, None


def attn_ffn_flops(B: int, T: int, P: int, H=512, F=1536, Le=6, Ld=6) -> float:
    """SURVEY 8d's attention + FFN subset of train_flops(): the Cv (variance predictors) and 2MH + H (mel projections, stop head)
    terms dropped — the FLOPs of the encoder and decoder FFT-block stacks, which is what `north_star`'s 40 % target is quoted on."""
    macs = (B * P * (Le * (4 * H * H + 3 * H * F) + Le * 2 * P * H)
            + B * T * (Ld * (8 * H * H + 3 * H * F) + Ld * (2 * T * H + (T + 1) * H)))
    return 6.0 * macs


def roofline_leg(eng, kk, batches, math, pmc_shape):
    """The dominant-kernel roofline of a workload: its step(s) run eagerly with every launch bracketed by events on the launch stream
    (kk.profile_start), algorithmic FLOPs per SURVEY 8d.  batches: the workload's batches (one eager step each, two passes)."""
    kk.profile_start()
    for _ in range(2):
        for b in batches:
            eng.zero_grad()
            eng._first_micro = True              # like the timed step: the first micro-batch of a cycle (weight gradients overwrite)
            eng.forward_backward(b, loss_scale=eng.dp_loss_scale, adaptive=True)
            eng._first_micro = False
            eng.optimizer_step(b["mel_specs"].shape[1])
    table = kernel_table(kk.profile_stop(), math == "bf16")
    shapes = {k: v for k, v in table.items() if k.startswith("  shape")}
    table = {k: v for k, v in table.items() if not k.startswith("  shape")}
    mfma = {k: v for k, v in table.items() if v["flops"] > 0}
    peak = PEAK_BF16_TFLOPS if math == "bf16" else PEAK_F32_TFLOPS
    nsteps = 2 * len(batches)

    def entry(name):
        v = mfma[name]
        ach = v["flops"] / (v["ms"] * 1e-3) / 1e12
        traffic, busy, note, src = pmc_entry(*pmc_shape, math, name) if pmc_shape else (None, None, None, None)
        e = {"kernel": name, "achieved": round(ach, 2), "frac": round(ach / peak, 4), "launches_per_step": round(v["launches"] / nsteps, 2),
             "avg_launch_us": round(v["ms"] * 1e3 / v["launches"], 2),
             "algorithmic_gflop_per_launch": round(v["flops"] / v["launches"] / 1e9, 3),
             "algorithmic_bytes_per_launch": round(v["bytes"] / v["launches"]), "traffic": traffic, "mfma_busy": busy,
             "pmc_source": src, "pmc_method": note}
        if v["floor_us"] > 0:                    # GEMMs: the tile's L2 -> LDS floor (2 B x MACs x (1/BM + 1/BN) at CUs x 54 B/clk x 2.4 GHz), per launch
            e["l2_to_lds_floor_us"] = round(v["floor_us"] / v["launches"], 2)
        return e
    ranked = sorted(mfma, key=lambda k: -mfma[k]["ms"])
    # dominant = the kernel SYMBOL with the most time (what a rocprofv3 summary ranks by: the rows of one symbol — the pair launch's
    # decoder and one-tile text-encoder launches — count together), reported through that symbol's largest row
    sym_ms = {}
    for k in ranked:
        sym_ms[k.split(" (")[0]] = sym_ms.get(k.split(" (")[0], 0.0) + mfma[k]["ms"]
    top_sym = max(sym_ms, key=sym_ms.get)
    dom = entry(next(k for k in ranked if k.split(" (")[0] == top_sym))
    roof = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": dom["frac"],
            "traffic": dom["traffic"], "traffic_unit": "bytes/launch", "traffic_source": dom["pmc_method"],
            "traffic_scope": "mean over ALL launches of the kernel symbol in the PMC pass (for the attention pair launch that includes the "
                             "text encoder's six one-tile launches per step, which the rate above lists apart)",
            "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "launches_per_step": dom["launches_per_step"],
            "avg_launch_us": dom["avg_launch_us"], "algorithmic_gflop_per_launch": dom["algorithmic_gflop_per_launch"],
            "mfma_busy": dom["mfma_busy"],
            "flop_convention": "SURVEY 8d: 2 FLOP/MAC, causal = lower triangle, attention backward = 4 matmuls (no credit for the "
                               "recomputed S / dP), event-timed on the launch stream in 2 eager steps",
            # the other MFMA kernels by time, same accounting (mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs), from
            # the committed PMC pass of this shape; null where the pass has no entry)
            "top_kernels": [{k: e[k] for k in ("kernel", "achieved", "frac", "launches_per_step", "avg_launch_us", "traffic", "mfma_busy",
                                               "l2_to_lds_floor_us") if k in e} for e in (entry(n) for n in ranked[:8])]}
    return roof, table, shapes


def stack_fraction(eng, batch):
    """`north_star`'s own number: the attention + FFN FLOP subset over the time the encoder and decoder stacks occupy in a
    replayed step, from device time stamps captured into the step's graphs (one-thread kk_timestamp launches at the marks of the step:
    they cost the traced step ~1 %).  Stack time = step start .. last decoder layer's forward, minus the length regulator / variance
    adaptor segment, plus loss gradients .. backward joined (decoder backward with the text encoder's backward beside it)."""
    B, T, P = batch["mel_specs"].shape[0], batch["mel_specs"].shape[1], batch["phoneme_indices"].shape[1]
    was = eng.trace
    eng.trace, eng._marks = True, {}
    eng._mark_buf = torch.zeros(512, dtype=torch.int64, device=eng.device)
    eng._graphs.clear()
    try:
        for _ in range(6):
            eng.train_step_graphed(batch)
        torch.cuda.synchronize()
        rows = {n: t for t, n in eng.timeline()}
    finally:
        eng.trace = was
        eng._graphs.clear()
    last_dec = max(int(n[3:n.index(" ")]) for n in rows if n.startswith("dec") and n.endswith("fwd done"))
    last_enc = max(int(n[3:n.index(" ")]) for n in rows if n.startswith("enc") and n.endswith("fwd done"))
    fwd = rows[f"dec{last_dec} fwd done"] - rows["step.start"] - (rows["memory ready"] - rows[f"enc{last_enc} fwd done"])
    bwd = rows["backward joined, partials reduced"] - rows["losses + loss gradients done"]
    us = fwd + bwd
    fl = attn_ffn_flops(B, T, P)
    return {"attn_ffn_gflop": round(fl / 1e9, 1), "stack_us": round(us, 1), "forward_us": round(fwd, 1), "backward_us": round(bwd, 1),
            "step_us": round(rows["optimizer done"] - rows["step.start"], 1),
            "attn_ffn_tflops": round(fl / us / 1e6, 1), "attn_ffn_mfma_frac": round(fl / us / 1e6 / PEAK_BF16_TFLOPS, 4),
            "definition": "SURVEY 8d attention + FFN FLOPs (train_flops without the Cv and 2MH + H terms) / device time between the step's "
                          "marks: [start .. last decoder layer forward] - [length regulator + variance adaptor] + [loss gradients .. backward joined]"}


def median(xs):
    ys = sorted(xs)
    n = len(ys)
    return ys[n // 2] if n % 2 else 0.5 * (ys[n // 2 - 1] + ys[n // 2])


def ragged_workload(max_frames=16384, n_utts=600, seed=7):
    """BASELINE configs[2] as a resident synthetic workload: utterance lengths U[90, 1500) frames (phonemes ~ frames / 12), packed by
    the reference-identical frame-budget sampler (B * longest <= max_frames, B in [4, 32]); every batch ragged like a collated one.
    Returns [batch dict on the GPU]."""
    import random
    from kokoro.data.cached import FrameBudgetBatchSampler
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    rnd = random.Random(seed)
    lens = [(t, max(12, min(60, t // 12 + rnd.randint(-3, 3)))) for t in (rnd.randrange(90, 1500) for _ in range(n_utts))]

    class DS:
        samples = [{"audio_length": t, "phoneme_length": p} for t, p in lens]
    out = []
    for k, idxs in enumerate(FrameBudgetBatchSampler(DS, max_frames, 4, 32, True, 0, 1, seed=seed, drop_last=True).global_batches()):
        mel = [lens[i][0] for i in idxs]
        ph = [lens[i][1] for i in idxs]
        b = synthetic_batch(len(idxs), max(mel), max(ph), seed=1000 + k, lengths=(mel, ph))
        out.append({kk_: v.cuda() for kk_, v in b.items()})
    return out


def extra_shapes(eng, kk, math, steps_1024=30, passes=3, n_batches=24):
    """The two other shapes the verdict asks the driver-run record to carry, measured in this process after the headline region:
    8x1024x128 (BASELINE configs[3]'s per-GPU shape: the north-star shape) replayed from hipGraphs, and configs[2] (dynamic
    batching, B*T <= 16384) over resident ragged batches through train_step_auto (graphs from the second sight of a shape).
    Every step is a full optimizer step (G = 1)."""
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    out = []
    b = {k: v.cuda() for k, v in synthetic_batch(8, 1024, 128, seed=1234).items()}
    for _ in range(6):
        eng.train_step_graphed(b)
    reps = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps_1024):
            eng.train_step_graphed(b)
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / steps_1024)
    ms = median(reps) * 1e3
    fl = train_flops(8, 1024, 128)
    out.append({"workload": "8x1024 mel frames x 128 phonemes (configs[3] per-GPU shape), hipGraph replay", "ms_per_step": round(ms, 3),
                "value": round(8 * 1024 / ms * 1e3, 1), "unit": "mel-frames/s", "steps": steps_1024, "repeats_ms": [round(x * 1e3, 3) for x in reps],
                "model_tflops": round(fl / ms / 1e9, 2), "model_mfma_frac": round(fl / ms / 1e9 / PEAK_BF16_TFLOPS, 4)})
    out[-1]["stacks"] = stack_fraction(eng, b)
    out[-1]["roofline"] = roofline_leg(eng, kk, [b], math, (8, 1024, 128))[0]
    batches = ragged_workload()[:n_batches]
    for _ in range(3):                                 # a shape's 1st sight runs eagerly, the 2nd is train_step_graphed's own eager pass
        for b in batches:                              # (it sizes the workspace), the 3rd captures: the 4th pass replays graphs
            eng.train_step_auto(b)
    reps = []
    for _ in range(passes):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            eng.train_step_auto(b)
        torch.cuda.synchronize()
        reps.append(time.perf_counter() - t0)
    dt = median(reps)
    valid = sum(int(b["mel_lengths"].sum()) for b in batches)
    padded = sum(b["mel_specs"].shape[0] * b["mel_specs"].shape[1] for b in batches)
    fl = sum(train_flops(b["mel_specs"].shape[0], b["mel_specs"].shape[1], b["phoneme_indices"].shape[1]) for b in batches)
    out.append({"workload": f"dyn16384: configs[2], dynamic batching B*T <= 16384 (B 4..32, T 90..1500), {len(batches)} resident ragged batches "
                            f"({len({tuple(b['mel_specs'].shape[:2]) for b in batches})} shapes), train_step_auto",
                "ms_per_step": round(dt / len(batches) * 1e3, 3), "value": round(valid / dt, 1), "unit": "valid mel-frames/s",
                "padded_frames_per_s": round(padded / dt, 1), "padded_tokens_per_step": round(padded / len(batches)),
                "passes_s": [round(x, 4) for x in reps],
                "model_tflops": round(fl / dt / 1e12, 2), "model_mfma_frac": round(fl / dt / 1e12 / PEAK_BF16_TFLOPS, 4)})
    out[-1]["roofline"] = roofline_leg(eng, kk, batches, math, None)[0]        # (one eager pass over the 24 shapes, twice)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="the timed region of --steps steps is repeated this many times; the line "
                    "reports the MEDIAN region (ms_per_step, value) and every region's ms_per_step")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--phonemes", type=int, default=64)
    ap.add_argument("--math", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--storage", choices=["auto", "f32", "bf16", "bf16-dec"], default="auto",
                    help="GEMM/attention operand storage in HBM (auto: bf16 in the bf16 mode)")
    ap.add_argument("--workload", choices=["fixed", "dyn16384"], default="fixed", help="dyn16384: BASELINE configs[2] (dynamic batching, "
                    "B*T <= 16384, resident ragged batches) as the timed workload of a 1-GPU run instead of the fixed --batch x --frames shape "
                    "(profiling / A-B runs; the default line carries it under extra_shapes)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="also time the CPU baseline at 8x1024x128 (minutes of CPU time)")
    ap.add_argument("--no-dropout", action="store_true", help="parity configuration (p = 0 everywhere)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-shapes", action="store_true", help="skip the 8x1024x128 and dyn16384 measurements after the headline region")
    ap.add_argument("--no-roofline", action="store_true", help="skip the eager event-timed roofline leg (A/B runs)")
    ap.add_argument("--lib", default="product", help="tools: 'tuning' = the flavour of the library that reads the KK_* A/B switches from "
                    "the environment (python -m kokoro_ruslan_amd.build --tuning), or the name of a --variant build")
    ap.add_argument("--set", action="append", default=[], metavar="ATTR=VALUE", help="tools: set an engine attribute (a fusion switch such "
                    "as attn_bwd_pair=0) for an A/B run; the line records it under config.engine_overrides")
    ap.add_argument("--kernel-table", default="")
    ap.add_argument("--gemm16", default="", help="tuning sweep hook: enable,thr128,thr12864,split_target for the bf16 GEMM core")
    args = ap.parse_args()

    from kokoro_ruslan_amd import dp, lib as kk
    from kokoro_ruslan_amd.engine import KokoroEngine
    from kokoro_ruslan_amd.spec import ModelDims, StepHyper
    from kokoro_ruslan_amd.synthetic import synthetic_batch

    if args.lib != "product":
        kk.use_library(args.lib)
    if args.gemm16:                    # A/B of the GEMM tile policy: read once when the library loads (no tuning calls in the ABI)
        os.environ["KK_GEMM16_TUNE"] = ",".join(args.gemm16.split(",")[-3:])
    rank, world, local = dp.init()
    if world != max(1, args.gpus) and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    torch.cuda.set_device(local)
    set_device_cus(torch.cuda.get_device_properties(local).multi_processor_count)      # (what the library's tile policy reads: hipDeviceAttributeMultiprocessorCount)
    B, T, P = args.batch, args.frames, args.phonemes
    hp = StepHyper(gradient_accumulation_steps=1)
    eng = KokoroEngine(ModelDims(), hp, math_mode=args.math, total_steps=20000, seed=0,
                       storage=args.storage)                           # same seed ⇒ identical replicas
    eng.train_dropout = not args.no_dropout          # reference-faithful: dropout, stochastic depth, SpecAugment on
    for kv in args.set:
        name, val = kv.split("=", 1)
        if not hasattr(eng, name):
            raise SystemExit(f"--set {kv}: KokoroEngine has no attribute {name}")
        setattr(eng, name, type(getattr(eng, name))(int(val)) if isinstance(getattr(eng, name), (bool, int)) else float(val))
    force = os.environ.get("KK_DP_FORCE") == "1"       # run the data-parallel code path (1-rank group) on a single GPU
    sync = dp.GradSync(world, force=force)
    eng.dp_loss_scale = sync.loss_scale
    batch = {k: v.cuda() for k, v in synthetic_batch(B, T, P, seed=1234 + rank).items()}
    use_sync = world > 1 or force
    # default data-parallel exchange: bucket by bucket INSIDE the step graph over the C ABI's kk_comm_* (RCCL);
    # KK_DP_LEGACY=1 = one torch.distributed all-reduce between the backward graph and the optimizer graph
    dp_mode = "none"
    if use_sync:
        dp_mode = "legacy" if os.environ.get("KK_DP_LEGACY") == "1" else "in-graph"
    if dp_mode == "in-graph":
        eng.dp_comm = dp.BucketedExchange.create(eng.dims, rank, world, eng.device)
        dp_mode += f" ({eng.dp_comm.backend}, {eng.dp_comm.payload} payload, {len(eng.dp_comm.plan)} buckets)"
        use_sync = False
    step = (lambda: eng.train_step(batch)) if args.no_graph else (lambda: eng.train_step_graphed(batch, sync if use_sync else None))
    dyn = None
    if args.workload == "dyn16384":
        if world != 1:
            raise SystemExit("--workload dyn16384 is a 1-GPU measurement")
        dyn = {"batches": ragged_workload()[:24], "i": 0}

        def step():   # noqa: F811
            b = dyn["batches"][dyn["i"] % len(dyn["batches"])]
            dyn["i"] += 1
            (eng.train_step if args.no_graph else eng.train_step_auto)(b)
        args.warmup = max(args.warmup, 3 * len(dyn["batches"]))        # eager, eager (graphed's own), capture: then replays
        args.steps = max(len(dyn["batches"]), args.steps // len(dyn["batches"]) * len(dyn["batches"]))     # whole passes
    if args.no_graph and world > 1:
        def step():   # noqa: F811
            eng.zero_grad()
            eng.forward_backward(batch, loss_scale=eng.dp_loss_scale, adaptive=True)
            sync(eng.arena.g)
            eng.optimizer_step(T)

    def warm():
        for _ in range(max(args.warmup, 2)):   # >= 2: first call allocates workspaces, second captures the graphs
            step()

    replicas_ok = None
    try:
        warm()
        if world > 1 or force:
            torch.cuda.synchronize()            # (the step's own RCCL communicator is idle before torch.distributed's is used)
            replicas_ok = dp.replicas_in_step(eng.arena.p)
    except Exception as e:                     # (a hang cannot be caught; an error of the in-graph exchange can)
        if not dp_mode.startswith("in-graph"):
            raise
        print(f"[bench] rank {rank}: in-graph exchange failed ({e}); falling back to the after-the-backward all-reduce", file=sys.stderr)
        replicas_ok = False
    if dp_mode.startswith("in-graph") and (world > 1 or force):
        # every rank must take the same branch: any rank that saw an error or diverged replicas sends all of them to the
        # round-1 form (one torch.distributed all-reduce between the backward graph and the optimizer graph)
        # (KK_BENCH_TEST_FALLBACK=1 takes this path on purpose: it is rehearsed on one GPU with KK_DP_FORCE=1)
        bad = torch.tensor([0.0 if (replicas_ok and os.environ.get("KK_BENCH_TEST_FALLBACK") != "1") else 1.0], device="cuda")
        dp.all_max(bad)
        if float(bad) > 0:
            if rank == 0:
                print("[bench] replicas out of step after the in-graph exchange; using the torch.distributed all-reduce", file=sys.stderr)
            eng.dp_comm = None
            eng._graphs.clear()
            dist_p = eng.arena.p.clone()
            torch.distributed.broadcast(dist_p, src=0)          # re-align the replicas on rank 0's weights and state
            eng.arena.p.copy_(dist_p)
            for slab in (eng.arena.m, eng.arena.v, eng.arena.ema, eng.opt_state):
                if slab is not None:
                    torch.distributed.broadcast(slab, src=0)
            eng.sync_shadow()
            dp_mode = "legacy (fallback)"
            use_sync = True
            step = lambda: eng.train_step_graphed(batch, sync)   # noqa: E731
            warm()
            torch.cuda.synchronize()
            replicas_ok = dp.replicas_in_step(eng.arena.p)
    # ---- the timed region: EXACTLY --steps steps between barrier + synchronize on both sides, MAX over ranks; repeated --repeats
    # times (a 20-step region is 0.08 s: one region is thin), the line reports the median region and every region's figure
    regions = []
    for _ in range(max(1, args.repeats)):
        torch.cuda.synchronize()
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dp.barrier()
        dt_r = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dp.all_max(dt_r)
        regions.append(float(dt_r))
    dt = median(regions)
    losses = eng.losses.cpu().tolist()
    stats = eng.opt_stats()
    if eng.encoder_stack_error():
        raise RuntimeError("a group barrier of the fused encoder launch timed out during the timed region: the number is void")

    roof, table = None, {}
    # From here on rank 0 works ALONE (roofline leg, extra shapes, baselines) while the other ranks wait at the barrier below: the
    # in-step exchange must be detached first — an eager forward_backward with eng.dp_comm set would issue RCCL collectives that no
    # other rank matches (a hang, not an error).  The timed numbers are already taken.
    eng.dp_comm = None
    if rank == 0 and not args.no_roofline and dyn is None:
        # Roofline leg: the same step, eager, every launch bracketed by events on the launch stream.
        roof, table, shapes = roofline_leg(eng, kk, [batch], args.math, (B, T, P))
        if args.kernel_table:
            tot = sum(x["ms"] for x in table.values())
            rows = sorted(table.items(), key=lambda kv: -kv[1]["ms"])
            with open(args.kernel_table, "w") as f:
                srows = sorted(shapes.items(), key=lambda kv: -kv[1]["ms"])
                json.dump({"total_ms_2_steps": tot, "kernels": {k: v for k, v in rows}, "shapes": {k: v for k, v in srows}}, f, indent=1)
    extra = stacks = None
    if rank == 0 and not args.no_roofline and not args.no_graph and dyn is None:
        stacks = stack_fraction(eng, batch)
    if rank == 0 and world == 1 and not args.no_extra_shapes and not args.no_graph and args.math == "bf16" and dyn is None:
        extra = extra_shapes(eng, kk, args.math)
        if eng.encoder_stack_error():
            raise RuntimeError("a group barrier of the fused encoder launch timed out during the extra shapes")
    dp.barrier()
    if rank != 0:
        dp.shutdown()
        return
    frames = world * B * T * args.steps
    fl = train_flops(B, T, P)
    if dyn is not None:                                  # valid frames and algorithmic FLOPs of the passes that were timed
        nb, passes = len(dyn["batches"]), args.steps // len(dyn["batches"])
        frames = passes * sum(int(b["mel_lengths"].sum()) for b in dyn["batches"])
        fl_pass = sum(train_flops(b["mel_specs"].shape[0], b["mel_specs"].shape[1], b["phoneme_indices"].shape[1]) for b in dyn["batches"])
        B, T = 1, frames // args.steps                   # (so that model_tflops below = FLOPs of the passes / time)
        fl = fl_pass * passes / args.steps
    ms_regions = [round(x / args.steps * 1e3, 3) for x in regions]
    out = {"metric": "mel-frames/sec (full train step)", "value": round(frames / dt, 1), "unit": "mel-frames/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.math, "data": "synthetic",
           "timed_regions": {"repeats": len(regions), "steps_each": args.steps, "reported": "median", "ms_per_step": ms_regions,
                             "min": min(ms_regions), "max": max(ms_regions)},
           "storage": eng.storage,
           "per_gpu": round(frames / dt / world, 1),
           "config": {"workload": ("dyn16384: BASELINE configs[2], dynamic batching B*T <= 16384 over 24 resident ragged batches, value = VALID mel frames/s, "
                                   "49.4M params, fwd+loss+bwd+clip+AdamW+EMA every step") if dyn is not None else
                                  f"kokoro acoustic-model train step, {B}x{T} mel frames x {P} phonemes per GPU "
                                  f"({'BASELINE configs[1]' if (B, T, P) == (8, 512, 64) else 'configs[3] per-GPU shape' if (B, T, P) == (8, 1024, 128) else 'non-baseline shape'}), "
                                  f"49.4M params, fwd+loss+bwd+clip+AdamW+EMA every step",
                      "global_batch": world * B, "frames": T, "phonemes": P, "parallelism": f"dp{world}",
                      "grad_allreduce": (dp_mode if dp_mode.startswith("in-graph") else
                                         (("after the backward (torch.distributed)" + (", fallback from the in-graph exchange" if dp_mode.endswith("(fallback)") else ""))
                                          if use_sync else "none (1 GPU)")),
                      # a tripwire, not a proof: two fp64 checksums (sum, sum of squares) of every rank's parameter arena, MIN- and
                      # MAX-reduced — equal on all ranks after the warm-up (None: 1 GPU); not a bitwise compare
                      "replicas_in_step": replicas_ok,
                      "replicas_in_step_check": "fp64 sum + sum of squares of the parameter arena, MIN == MAX over ranks (checksum tripwire)",
                      "grad_accumulation": 1,
                      "dropout": ("off (p=0 parity configuration)" if args.no_dropout else
                                  "on: enc 0.15 / dec 0.20 / dec-input 0.15 / variance 0.10, stochastic depth 0.1, SpecAugment (config.py defaults)"),
                      "hipgraph": not args.no_graph, "engine_overrides": args.set or None},
           "final_losses": [round(x, 5) for x in losses], "optimizer_steps": stats["attempt"], "skipped": stats["skipped"],
           "roofline": roof,
           "stacks": stacks,
           "model_tflops": round(frames / dt * fl / (B * T) / 1e12, 2),
           "model_mfma_frac": round(frames / dt * fl / (B * T) / 1e12 / world / (PEAK_BF16_TFLOPS if args.math == "bf16" else PEAK_F32_TFLOPS), 4)}
    if extra is not None:
        out["extra_shapes"] = extra
    if world == 1 and not args.no_cpu_baseline:
        out["torch_rocm_baseline"] = torch_rocm_baseline(B, T, P)
        out["cpu_baseline"] = cpu_baseline(B, T, P, full=args.cpu_baseline_full)
    print(json.dumps(out), flush=True)
    dp.shutdown()


if __name__ == "__main__":
    main()
