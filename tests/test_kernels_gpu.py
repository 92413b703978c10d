"""GPU suite, kernel level: every C-ABI entry point against a plain fp32 PyTorch/NumPy restatement on the same
seeded inputs (the oracle's pieces), called through the C ABI (kokoro_ruslan_amd.lib.call)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import kokoro_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kk():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from kokoro_ruslan_amd import lib
    lib.load()
    return lib


def dev(t):
    return t.cuda().contiguous()


def close(got, ref, atol, rtol, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{bad.numel()} off; max|err|={float(err.max()):.3e} "
                                 f"max|ref|={float(ref.abs().max()):.3e}")


TOL = {0: (2e-5, 2e-5), 1: (3e-2, 3e-2)}   # math mode -> (atol, rtol) for O(1) data


def test_mfma_fragment_layout(kk):
    o32, o16 = torch.zeros(32, 32, device="cuda"), torch.zeros(32, 32, device="cuda")
    kk.call("kk_mfma_probe", o32, o16)
    i = torch.arange(32.0)[:, None]
    k = torch.arange(16.0)
    A = i + 0.25 * k[None, :] - 3.0
    Bm = 0.5 * k[:, None] - 0.125 * torch.arange(32.0)[None, :] + 1.0
    ref = A @ Bm
    close(o32, ref, 1e-4, 1e-6, "f32 mfma 32x32x2")
    close(o16, A.bfloat16().float() @ Bm.bfloat16().float(), 1e-3, 1e-6, "bf16 mfma 32x32x16")


@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 80, 80), (517, 1536, 512), (64, 512, 1544),
                                   (4096, 2048, 136)])   # the last one takes the 128x128-tile path
def test_gemm_layouts(kk, math_mode, ta, tb, M, N, K):
    if ta and M % 4:
        M += 4 - M % 4
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    Bm = torch.randn((K, N) if tb else (N, K), generator=g)
    bias = torch.randn(N, generator=g)
    res = torch.randn(50, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    Aop = A.t() if ta else A
    Bop = Bm if tb else Bm.t()
    if math_mode:
        Aop, Bop = Aop.bfloat16().float(), Bop.bfloat16().float()
    ref = 0.5 * (Aop.double() @ Bop.double()).float() + bias + res[torch.arange(M) % 50] + 2.0 * C0
    Cd = dev(C0)
    kk.call("kk_gemm", ta, tb, M, N, K, 0.5, dev(A), A.shape[1], dev(Bm), Bm.shape[1], 2.0, Cd, N, dev(bias), dev(res),
            N, 50, 1, math_mode, 0)
    atol, rtol = (2e-4, 2e-5) if math_mode == 0 else (5e-2, 1e-3)
    close(Cd, ref, atol * math.sqrt(K / 64), rtol, f"gemm ta={ta} tb={tb} {M}x{N}x{K} math={math_mode}")


@pytest.mark.parametrize("math_mode", [0, 1])
def test_gemm_splitk_wgrad(kk, math_mode):
    g = torch.Generator().manual_seed(5)
    M, N, K = 512, 256, 4096              # dW[M=512 out rows, N=256] = dY^T[K=4096 rows] . X
    dY, X = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    acc0 = torch.randn(M, N, generator=g)
    a, b = (dY.bfloat16().float(), X.bfloat16().float()) if math_mode else (dY, X)
    ref = (a.t().double() @ b.double()).float()
    for beta, split in ((1.0, 0), (0.0, 0), (1.0, 8), (0.0, 1)):
        Cd = dev(acc0)
        kk.call("kk_gemm", 1, 1, M, N, K, 1.0, dev(dY), M, dev(X), N, beta, Cd, N, None, None, 0, 0, split, math_mode, 0)
        close(Cd, ref + beta * acc0, 2e-3 if math_mode == 0 else 0.3, 1e-3, f"split-k beta={beta} split={split}")


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 64, 64), (200, 136, 192), (520, 1536, 512), (4096, 512, 1536), (1000, 3072, 320),
                                   (72, 264, 4096), (8200, 512, 512), (8192, 1000, 192)])
def test_gemm_bf16_dma_core(kk, ta, tb, M, N, K):
    """bf16 x bf16 operands (the DMA-staged core, kk_gemm16.hip): strided operands and outputs, bias, residual with
    a row period, alpha/beta, bf16 and fp32 outputs, ragged tiles; for k-strided operands also a K that is not a
    multiple of the 64-deep stage (zero-filled by the buffer bounds check)."""
    g = torch.Generator().manual_seed(M * 3 + N * 5 + K + 2 * ta + tb)
    if ta and tb:
        K += 40                                   # K tail: only legal when both operands are k-strided
    pad_a, pad_b, pad_c = 16, 8, 24                # leading dimensions larger than the logical widths
    A = torch.randn((K, M + pad_a) if ta else (M, K + pad_a), generator=g).bfloat16()
    Bm = torch.randn((K, N + pad_b) if tb else (N, K + pad_b), generator=g).bfloat16()
    Al = (A[:, :M].float().t() if ta else A[:, :K].float()).double()
    Bl = (Bm[:, :N].float() if tb else Bm[:, :K].float().t()).double()
    bias, res = torch.randn(N, generator=g), torch.randn(50, N + 8, generator=g)
    rows = torch.arange(M) % 50
    ref = (0.5 * (Al @ Bl)).float() + bias + res[rows][:, :N]
    Ad, Bd = dev(A), dev(Bm)
    for c16 in (0, 1):
        Cd = torch.full((M, N + pad_c), 7.0, device="cuda", dtype=torch.bfloat16 if c16 else torch.float32)
        kk.call("kk_gemm", ta, tb, M, N, K, 0.5, Ad, A.shape[1], Bd, Bm.shape[1], 0.0, Cd, N + pad_c, dev(bias), dev(res), N + 8, 50,
                1, 1, 3 | (c16 << 2))
        close(Cd[:, :N], ref, 2e-3 * math.sqrt(K / 64) + (0.1 if c16 else 0.0), 1e-2 if c16 else 1e-4, f"dma core c16={c16}")
        assert bool((Cd[:, N:].float() == 7.0).all()), "columns beyond N must not be written"
    acc = torch.randn(M, N, generator=g)
    for split in (1, 0, 3):                        # beta = 1 accumulate: plain, auto split-K, forced split-K (fp32 atomics)
        Cd = dev(acc)
        kk.call("kk_gemm", ta, tb, M, N, K, 1.0, Ad, A.shape[1], Bd, Bm.shape[1], 1.0, Cd, N, None, None, 0, 0, split, 1, 3)
        close(Cd, (Al @ Bl).float() + acc, 3e-3 * math.sqrt(K / 64), 1e-4, f"dma core accumulate split={split}")
    Cd = torch.empty(M, N, device="cuda")           # beta = 0 with split-K (memset + atomics)
    kk.call("kk_gemm", ta, tb, M, N, K, 1.0, Ad, A.shape[1], Bd, Bm.shape[1], 0.0, Cd, N, None, None, 0, 0, 2, 1, 3)
    close(Cd, (Al @ Bl).float(), 3e-3 * math.sqrt(K / 64), 1e-4, "dma core beta=0 split")


# The plain kernels of the large-tile family (kk_gemm16x.hip): kk_gemm takes them for k-contiguous A, no k-slices, K >= 1024 when the
# bytes through the busiest CU say so — q|k|v and linear1 dgrads, linear2 forward at 8192+ rows (transformers.py:131-136,90-91) and
# every such launch under dynamic batching (ragged row counts).  The route is asserted, so a policy change cannot un-test them.
@pytest.mark.parametrize("tb", [0, 1])
@pytest.mark.parametrize("M,N,K,tile", [(8192, 512, 1536, "128,128"), (8192, 512, 3072, "128,128"), (15996, 512, 1536, "256,128"),
                                        (8192, 1000, 1088, "256,128"), (4090, 1000, 1088, "128,128"), (16384, 1536, 1024, "256,128")])
def test_gemm_bf16_large_tile_plain(kk, tb, M, N, K, tile):
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + tb)
    pad_a, pad_b, pad_c = 16, 8, 24
    A = torch.randn(M, K + pad_a, generator=g).bfloat16()
    Bm = torch.randn((K, N + pad_b) if tb else (N, K + pad_b), generator=g).bfloat16()
    Ad, Bd = dev(A), dev(Bm)
    Al, Bl = Ad[:, :K].double(), (Bd[:, :N].double() if tb else Bd[:, :K].double().t())
    bias = torch.randn(N, generator=g)
    prod = Al @ Bl                                                  # float64 on the device (the host takes minutes at these sizes)
    ref = (0.5 * prod).float() + dev(bias)
    want = f"g16x<0,{tb},{tile},3,0,"
    for c16 in (0, 1):
        Cd = torch.full((M, N + pad_c), 7.0, device="cuda", dtype=torch.bfloat16 if c16 else torch.float32)
        kk.call("kk_gemm", 0, tb, M, N, K, 0.5, Ad, A.shape[1], Bd, Bm.shape[1], 0.0, Cd, N + pad_c, dev(bias), None, 0, 0, 1, 1,
                3 | (c16 << 2))
        assert kk.last_kernel().startswith(want), (kk.last_kernel(), want)
        close(Cd[:, :N], ref, 2e-3 * math.sqrt(K / 64) + (0.1 if c16 else 0.0), 1e-2 if c16 else 1e-4, f"g16x plain c16={c16}")
        assert bool((Cd[:, N:].float() == 7.0).all()), "columns beyond N must not be written"
    # the general epilogue (residual with a row period, beta = 1 accumulation) straight from the accumulators
    res = torch.randn(50, N + 8, generator=g)
    acc = torch.randn(M, N, generator=g)
    Cd = dev(acc)
    kk.call("kk_gemm", 0, tb, M, N, K, 1.0, Ad, A.shape[1], Bd, Bm.shape[1], 1.0, Cd, N, None, dev(res), N + 8, 50, 1, 1, 3)
    assert kk.last_kernel().startswith(want), (kk.last_kernel(), want)
    close(Cd, prod.float() + dev(acc) + dev(res)[torch.arange(M, device="cuda") % 50][:, :N], 3e-3 * math.sqrt(K / 64), 1e-4, "g16x plain accumulate + residual")


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("dtypes", [1, 2, 3, 7, 6])
@pytest.mark.parametrize("M,N,K", [(136, 72, 200), (4096, 2048, 136), (520, 1536, 512)])
def test_gemm_bf16_storage(kk, ta, tb, dtypes, M, N, K):
    """A / B / C held as bf16 in HBM (bit0 / bit1 / bit2 of `dtypes`): same result as fp32 storage of the rounded values."""
    g = torch.Generator().manual_seed(M + N + K + dtypes)
    A = torch.randn((K, M) if ta else (M, K), generator=g).bfloat16()
    Bm = torch.randn((K, N) if tb else (N, K), generator=g).bfloat16()
    bias = torch.randn(N, generator=g)
    ref = ((A.float().t() if ta else A.float()).double() @ (Bm.float() if tb else Bm.float().t()).double()).float() + bias
    a16, b16, c16 = dtypes & 1, (dtypes >> 1) & 1, (dtypes >> 2) & 1
    Ad = dev(A if a16 else A.float())
    Bd = dev(Bm if b16 else Bm.float())
    Cd = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if c16 else torch.float32)
    kk.call("kk_gemm", ta, tb, M, N, K, 1.0, Ad, A.shape[1], Bd, Bm.shape[1], 0.0, Cd, N, dev(bias), None, 0, 0, 1, 1, dtypes)
    close(Cd, ref, 2e-3 * math.sqrt(K / 64) + (0.1 if c16 else 0.0), 1e-2 if c16 else 1e-4, f"bf16-storage gemm dtypes={dtypes}")
    if not c16 and ta and tb:      # weight-gradient form: split-K atomics accumulate into fp32
        acc = torch.randn(M, N, generator=g)
        Cd = dev(acc)
        kk.call("kk_gemm", ta, tb, M, N, K, 1.0, Ad, A.shape[1], Bd, Bm.shape[1], 1.0, Cd, N, None, None, 0, 0, 4, 1, dtypes)
        close(Cd, ref - bias + acc, 3e-3 * math.sqrt(K / 64), 1e-4, "bf16-storage split-k")


def test_colsum(kk):
    X = torch.randn(777, 200)
    out = dev(torch.ones(200))
    kk.call("kk_colsum_acc", dev(X), 200, 777, 200, out, 0)
    close(out, 1 + X.sum(0), 1e-3, 1e-5, "colsum")


def _attn_ref(q, k, v, causal, key_mask, scale):
    # q,k,v [B,h,S,d] double
    s = q @ k.transpose(-1, -2) * scale
    if causal:
        s = s + torch.triu(torch.full(s.shape[-2:], float("-inf"), dtype=s.dtype), 1)
    if key_mask is not None:
        s = s.masked_fill(key_mask.bool()[:, None, None, :], float("-inf"))
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("B,h,Sq,Sk,causal,masked", [(2, 2, 64, 64, 0, 1), (1, 2, 200, 200, 1, 0),
                                                      (2, 1, 37, 150, 0, 1), (1, 8, 300, 300, 1, 0),
                                                      (2, 2, 130, 70, 0, 0)])
def test_attention_fwd_bwd(kk, math_mode, B, h, Sq, Sk, causal, masked):
    g = torch.Generator().manual_seed(Sq * 3 + Sk + causal)
    H = h * 64
    Q, K, V = (torch.randn(B, S, H, generator=g) for S in (Sq, Sk, Sk))
    dO = torch.randn(B, Sq, H, generator=g)
    km = None
    if masked:
        km = torch.rand(B, Sk, generator=g) < 0.3
        km[:, 0] = False
    scale = 1 / 8.0

    def heads(x, S):
        return x.view(B, S, h, 64).transpose(1, 2)
    Qr, Kr, Vr = (t.clone().double().requires_grad_(True) for t in (Q, K, V))
    if math_mode:
        Qr, Kr, Vr = (t.detach().bfloat16().double().requires_grad_(True) for t in (Q, K, V))
    ref = _attn_ref(heads(Qr, Sq), heads(Kr, Sk), heads(Vr, Sk), causal, km, scale).transpose(1, 2).reshape(B, Sq, H)
    ref.backward(dO.double())
    Qd, Kd, Vd, dOd = dev(Q), dev(K), dev(V), dev(dO)
    Od = torch.zeros(B, Sq, H, device="cuda")
    lse = torch.zeros(B, h, Sq, device="cuda")
    kmd = dev(km.to(torch.uint8)) if km is not None else None
    kk.call("kk_attn_fwd", Qd, Kd, Vd, Od, lse, B, h, Sq, Sk, H, H, H, H, kmd, causal, scale, None, 0, 0.0, math_mode, 0)
    atol, rtol = (2e-5, 1e-4) if math_mode == 0 else (3e-2, 3e-2)
    close(Od, ref, atol, rtol, "attn fwd")
    delta = torch.zeros(B, h, Sq, device="cuda")
    kk.call("kk_attn_delta", Od, dOd, delta, B, h, Sq, H, H, 0)
    dQ, dK, dV = (torch.zeros_like(t) for t in (Qd, Kd, Vd))
    kk.call("kk_attn_bwd_dq", Qd, Kd, Vd, dOd, lse, delta, dQ, B, h, Sq, Sk, H, H, H, H, H, kmd, causal, scale, None, 0, 0.0, math_mode, 0, None, 0, None)
    kk.call("kk_attn_bwd_dkv", Qd, Kd, Vd, dOd, lse, delta, dK, dV, B, h, Sq, Sk, H, H, H, H, H, H, kmd, causal, scale,
            None, 0, 0.0, math_mode, 0, None)
    atol, rtol = (1e-4, 1e-3) if math_mode == 0 else (8e-2, 5e-2)
    close(dQ, Qr.grad, atol, rtol, "attn dQ")
    close(dK, Kr.grad, atol, rtol, "attn dK")
    close(dV, Vr.grad, atol, rtol, "attn dV")


def test_attention_strided_fused_qkv(kk):
    """Operands living inside a fused [rows, 3H] projection buffer (row stride 3H)."""
    B, h, S = 2, 2, 96
    H = h * 64
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B, S, 3 * H, generator=g)
    ref = _attn_ref(*(qkv[..., i * H:(i + 1) * H].double().view(B, S, h, 64).transpose(1, 2) for i in range(3)), 1, None,
                    0.125).transpose(1, 2).reshape(B, S, H)
    d = dev(qkv)
    O_ = torch.zeros(B, S, H, device="cuda")
    lse = torch.zeros(B, h, S, device="cuda")
    kk.call("kk_attn_fwd", d, d[..., H:], d[..., 2 * H:], O_, lse, B, h, S, S, 3 * H, 3 * H, 3 * H, H, None, 1, 0.125, None, 0, 0.0, 0, 0)
    close(O_, ref, 2e-5, 1e-4, "attn fused-qkv strides")


@pytest.mark.parametrize("rows,H", [(37, 64), (1000, 512), (5, 2048), (129, 128)])
def test_layernorm(kk, rows, H):
    g = torch.Generator().manual_seed(rows + H)
    x = torch.randn(rows, H, generator=g) * 2 + 0.5
    gam, bet = torch.randn(H, generator=g), torch.randn(H, generator=g)
    dy = torch.randn(rows, H, generator=g)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gam, bet))
    y = F.layer_norm(xr, (H,), gr, br, 1e-5)
    y.backward(dy)
    yd, mean, rstd = torch.empty(rows, H, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    kk.call("kk_layernorm_fwd", dev(x), dev(gam), dev(bet), yd, mean, rstd, rows, H, 0)
    close(yd, y, 2e-5, 2e-5, "ln fwd")
    dx0 = torch.randn(rows, H, generator=g)
    dx, dg, db = dev(dx0), torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    kk.call("kk_layernorm_bwd", dev(dy), dev(x), dev(gam), mean, rstd, dx, 1, dg, db, None, rows, H, 0)
    close(dx, xr.grad + dx0, 1e-4, 1e-4, "ln dx (accumulate)")
    close(dg, gr.grad, 1e-3, 1e-4, "ln dgamma")
    close(db, br.grad, 1e-3, 1e-4, "ln dbeta")


@pytest.mark.parametrize("rows,H", [(33, 64), (700, 512)])
def test_rmsnorm_residual(kk, rows, H):
    g = torch.Generator().manual_seed(rows)
    x, res = torch.randn(rows, H, generator=g), torch.randn(rows, H, generator=g)
    gain, dy = torch.randn(H, generator=g), torch.randn(rows, H, generator=g)
    xr, gr = x.clone().requires_grad_(True), gain.clone().requires_grad_(True)
    y = res + O._rms_norm(xr, gr)
    y.backward(dy)
    yd, rstd = torch.empty(rows, H, device="cuda"), torch.empty(rows, device="cuda")
    kk.call("kk_rmsnorm_fwd", dev(x), dev(gain), dev(res), yd, rstd, rows, H, 0)
    close(yd, y, 2e-5, 2e-5, "rms fwd")
    dx, dg = torch.empty(rows, H, device="cuda"), torch.zeros(H, device="cuda")
    kk.call("kk_rmsnorm_bwd", dev(dy), dev(x), dev(gain), rstd, dx, dg, None, rows, H, 0)
    close(dx, xr.grad, 1e-4, 1e-4, "rms dx")
    close(dg, gr.grad, 1e-3, 1e-4, "rms dgain")


@pytest.mark.parametrize("rope", [0, 1])
def test_headnorm_rope_fused_qkv(kk, rope):
    """q|k|v thirds of a fused [rows, 3H] projection in one launch: own gains, RoPE on q and k only."""
    B, S, h = 2, 50, 3
    H = h * 64
    g = torch.Generator().manual_seed(9 + rope)
    x = torch.randn(B * S, 3 * H, generator=g)
    gains = [torch.rand(64, generator=g) + 0.5 for _ in range(3)]
    dy = torch.randn(B * S, 3 * H, generator=g)
    cos, sin = O.rope_tables(S, 64)
    xr = x.clone().requires_grad_(True)
    gr = [t.clone().requires_grad_(True) for t in gains]
    outs = []
    for j in range(3):
        n = O._rms_norm(xr[:, j * H:(j + 1) * H].view(B, S, h, 64), gr[j])
        if rope and j < 2:
            n = n * cos[None, :, None, :] + O._rotate_half(n) * sin[None, :, None, :]
        outs.append(n.reshape(B * S, H))
    ref = torch.cat(outs, 1)
    ref.backward(dy)
    xd, y = dev(x), torch.empty(B * S, 3 * H, device="cuda")
    gd = [dev(t) for t in gains]
    ct, st_ = dev(cos), dev(sin)
    kk.call("kk_headnorm_rope_fwd", xd, 3 * H, y, 3 * H, B * S, h, S, 3, gd[0], gd[1], gd[2], 3 if rope else 0, ct, st_, 0)
    close(y, ref, 2e-5, 2e-5, "headnorm fwd")
    dx, dg = torch.zeros(B * S, 3 * H, device="cuda"), [torch.zeros(64, device="cuda") for _ in range(3)]
    kk.call("kk_headnorm_rope_bwd", dev(dy), 3 * H, xd, 3 * H, dx, 3 * H, B * S, h, S, 3, gd[0], gd[1], gd[2], dg[0], dg[1], dg[2],
            None, 3 if rope else 0, ct, st_, 0)
    close(dx, xr.grad, 1e-4, 1e-4, "headnorm dx")
    for j in range(3):
        close(dg[j], gr[j].grad, 1e-3, 1e-4, f"headnorm dgain{j}")
    # single part with a row stride (cross-attention q) and the 2-part k|v form
    y1 = torch.empty(B * S, H, device="cuda")
    kk.call("kk_headnorm_rope_fwd", xd[:, H:], 3 * H, y1, H, B * S, h, S, 1, gd[1], None, None, 1 if rope else 0, ct, st_, 0)
    close(y1, ref[:, H:2 * H], 2e-5, 2e-5, "headnorm single part, strided")


def test_glu(kk):
    rows, Fd = 123, 96
    h = torch.randn(rows, 2 * Fd) * 2
    dg = torch.randn(rows, Fd)
    hr = h.clone().requires_grad_(True)
    gate, lin = hr.chunk(2, -1)
    y = F.gelu(gate) * lin
    y.backward(dg)
    out = torch.empty(rows, Fd, device="cuda")
    kk.call("kk_glu_fwd", dev(h), out, rows, Fd, None, 0, 0.0, 0)
    close(out, y, 1e-5, 1e-5, "glu fwd")
    dh = torch.empty(rows, 2 * Fd, device="cuda")
    kk.call("kk_glu_bwd", dev(dg), dev(h), dh, rows, Fd, None, 0, 0.0, 0)
    close(dh, hr.grad, 1e-5, 1e-5, "glu bwd")


def test_embed(kk):
    B, P, H, V = 3, 17, 64, 59
    g = torch.Generator().manual_seed(2)
    ids, stress = torch.randint(0, V, (B, P), generator=g), torch.randint(0, 3, (B, P), generator=g)
    emb, semb, pe = torch.randn(V, H, generator=g), torch.randn(3, H, generator=g), torch.randn(40, H, generator=g)
    dout = torch.randn(B, P, H, generator=g)
    er, sr = emb.clone().requires_grad_(True), semb.clone().requires_grad_(True)
    y = F.embedding(ids, er) * 8.0 + F.embedding(stress, sr, padding_idx=0) + pe[:P]
    y.backward(dout)
    out = torch.empty(B, P, H, device="cuda")
    kk.call("kk_embed_fwd", dev(ids), dev(stress), dev(emb), dev(semb), dev(pe), out, B, P, H, 8.0, None, 0, 0.0)
    close(out, y, 1e-6, 1e-6, "embed fwd")
    de, ds = torch.zeros(V, H, device="cuda"), torch.zeros(3, H, device="cuda")
    kk.call("kk_embed_bwd", dev(ids), dev(stress), dev(dout), de, ds, B, P, H, 8.0, None, 0, 0.0)
    close(de, er.grad, 1e-4, 1e-5, "embed demb")
    close(ds, sr.grad, 1e-4, 1e-5, "embed dstress")


@pytest.mark.parametrize("H,bf,p", [(512, 1, 0.15), (512, 0, 0.0), (64, 0, 0.15), (768, 1, 0.1)])
def test_embed_ln_fwd_equals_three_launches(kk, H, bf, p):
    """kk_embed_ln_fwd == kk_ids_eq_zero + kk_embed_fwd + kk_layernorm_fwd, bit for bit (same seed and call site: same dropout masks),
    and against the float64 LayerNorm of the embedding sum without dropout."""
    B, P, V = 5, 67, 59
    g = torch.Generator().manual_seed(H + bf)
    ids, stress = dev(torch.randint(0, V, (B, P), generator=g)), dev(torch.randint(0, 3, (B, P), generator=g))
    ids[1, 40:] = 0
    emb, semb, pe = dev(torch.randn(V, H, generator=g)), dev(torch.randn(3, H, generator=g)), dev(torch.randn(80, H, generator=g))
    gam, bet = dev(1 + 0.1 * torch.randn(H, generator=g)), dev(0.1 * torch.randn(H, generator=g))
    ydt = torch.bfloat16 if bf else torch.float32
    seed = _seed(77) if p > 0 else None
    xa, ma = torch.empty(B * P, H, device="cuda"), torch.empty(B, P, dtype=torch.uint8, device="cuda")
    ya, mea, rsa = torch.empty(B * P, H, device="cuda", dtype=ydt), torch.empty(B * P, device="cuda"), torch.empty(B * P, device="cuda")
    kk.call("kk_ids_eq_zero", ids, ma, B * P)
    kk.call("kk_embed_fwd", ids, stress, emb, semb, pe, xa, B, P, H, float(H ** 0.5), seed, 1, p)
    kk.call("kk_layernorm_fwd", xa, gam, bet, ya, mea, rsa, B * P, H, bf)
    xb, mb = torch.zeros_like(xa), torch.full_like(ma, 7)
    yb, meb, rsb = torch.zeros_like(ya), torch.zeros_like(mea), torch.zeros_like(rsa)
    kk.call("kk_embed_ln_fwd", ids, stress, emb, semb, pe, xb, B, P, H, float(H ** 0.5), seed, 1, p, mb, gam, bet, yb, bf, meb, rsb)
    torch.cuda.synchronize()
    assert torch.equal(xb, xa) and torch.equal(mb, ma) and int(mb.sum()) >= 27
    assert torch.equal(meb, mea) and torch.equal(rsb, rsa) and torch.equal(yb, ya)
    if p == 0.0:
        x64 = emb.double()[ids.view(-1)] * float(H ** 0.5) + semb.double()[stress.view(-1)] + pe.double()[:P].repeat(B, 1)
        y64 = (x64 - x64.mean(1, keepdim=True)) / torch.sqrt(x64.var(1, unbiased=False, keepdim=True) + 1e-5) * gam.double() + bet.double()
        close(yb, y64, 2e-5, 2e-5, "embed + LayerNorm vs float64")


def test_length_regulator_golden_bit_exact(kk, golden_dir):
    fx = np.load(os.path.join(golden_dir, "length_regulator.npz"))
    for i in range(int(fx["n"])):
        c = {k: fx[f"{i}/{k}"] for k in ("tokens", "dur", "max_len", "out", "idx", "lens")}
        dur = c["dur"]
        dur = np.trunc(dur).astype(np.int64) if dur.dtype.kind == "f" else dur.astype(np.int64)   # lengths.py:31 .long()
        B, P = dur.shape
        L = c["idx"].shape[1]
        idx, lens, tot = (torch.full((B, L), -9, dtype=torch.int64, device="cuda"),
                          torch.zeros(B, dtype=torch.int64, device="cuda"), torch.zeros(B, dtype=torch.int64, device="cuda"))
        kk.call("kk_length_regulate_index", dev(torch.from_numpy(dur)), idx, lens, tot, B, P, L)
        assert np.array_equal(idx.cpu().numpy(), c["idx"]), f"case {i}: idx differs"
        assert np.array_equal(lens.cpu().numpy(), c["lens"]), f"case {i}: lens differs"
        tok = torch.from_numpy(c["tokens"]).float()
        if tok.dim() == 3 and tok.shape[2] % 4 == 0:
            out = torch.empty(B, L, tok.shape[2], device="cuda")
            kk.call("kk_length_regulate_gather", dev(tok), idx, out, B, P, L, tok.shape[2])
            assert np.array_equal(out.cpu().numpy(), c["out"].astype(np.float32)), f"case {i}: payload differs"


def test_length_regulator_full_size_properties(kk):
    """BASELINE sizes (8x1024 frames, P=128, H=512): bit-exact vs the oracle + size-independent properties."""
    d = O.ModelDims()
    b = O.synthetic_batch(8, 1024, 128, d, ragged=True)
    B, P, L, H = 8, 128, 1024, 512
    dur = b["phoneme_durations"]
    idx, lens, tot = (torch.empty(B, L, dtype=torch.int64, device="cuda"), torch.empty(B, dtype=torch.int64, device="cuda"),
                      torch.empty(B, dtype=torch.int64, device="cuda"))
    kk.call("kk_length_regulate_index", dev(dur), idx, lens, tot, B, P, L)
    ref_idx, ref_lens, _ = O.length_regulate_index(dur.numpy(), L)
    assert np.array_equal(idx.cpu().numpy(), ref_idx) and np.array_equal(lens.cpu().numpy(), ref_lens)
    i = idx.cpu()
    assert (tot.cpu() == dur.clamp(min=0).sum(1)).all()
    valid = i >= 0
    assert (valid.sum(1) == lens.cpu()).all()                                   # exactly len_b expanded frames
    assert ((i[:, 1:] >= i[:, :-1]) | ~valid[:, 1:]).all()                      # sortedness
    counts = torch.stack([torch.bincount(i[bb][valid[bb]], minlength=P) for bb in range(B)])
    assert (counts == dur.clamp(min=0)).all()                                   # each token repeated dur times
    x = torch.randn(B, P, H)
    out = torch.empty(B, L, H, device="cuda")
    kk.call("kk_length_regulate_gather", dev(x), idx, out, B, P, L, H)
    assert torch.equal(out.cpu(), O.length_regulate(x, dur, L))
    mx = torch.zeros(1, dtype=torch.int64, device="cuda")
    kk.call("kk_max_i64", dev(dur), dur.numel(), mx)
    assert int(mx) == int(dur.max())


@pytest.mark.parametrize("L", [40, 600, 513])
def test_variance_predictor_chain(kk, L):
    """im2col+GEMM conv, chunked GroupNorm(1,C)+ReLU, Linear(C->1)+mask: forward and backward vs the oracle."""
    B, H, Fv = 2, 64, 32
    g = torch.Generator().manual_seed(L)
    names = O._varpred_names("vp", H, Fv, 3)
    P = {n: torch.randn(s, generator=g) * (0.2 if len(s) > 1 else 0.5) for n, s in names}
    x = torch.randn(B, L, H, generator=g)
    mask = torch.zeros(B, L, dtype=torch.bool)
    mask[1, L - 7:] = True
    dout = torch.randn(B, L, generator=g)
    Pr = {n: p.clone().requires_grad_(True) for n, p in P.items()}
    xr = x.clone().requires_grad_(True)
    ref = O.variance_predictor(Pr, "vp", xr, mask)
    ref.backward(dout)
    Pd = {n: dev(p) for n, p in P.items()}
    Gd = {n: torch.zeros_like(p) for n, p in Pd.items()}
    xd, rows, nch = dev(x), B * L, (L + 511) // 512
    scratch = torch.zeros(2 * B * nch, dtype=torch.float64, device="cuda")
    acts, cin, inp = [], H, xd
    for li in range(2):
        col = torch.empty(rows, 3 * cin, device="cuda")
        kk.call("kk_im2col3_fwd", inp, col, B, L, cin, 512, 0)
        c = torch.empty(rows, Fv, device="cuda")
        kk.call("kk_gemm", 0, 0, rows, Fv, 3 * cin, 1.0, col, 3 * cin, Pd[f"vp.conv_layers.{li}.weight"], 3 * cin, 0.0, c, Fv,
                Pd[f"vp.conv_layers.{li}.bias"], None, 0, 0, 1, 0, 0)
        y, stats = torch.empty(rows, Fv, device="cuda"), torch.empty(B * nch, 2, device="cuda")
        kk.call("kk_groupnorm_relu_fwd", c, Pd[f"vp.norms.{li}.weight"], Pd[f"vp.norms.{li}.bias"], y, stats, scratch, B, L, Fv, 512, None, 0, 0.0)
        acts.append((col, c, y, stats, cin))
        inp, cin = y, Fv
    md = dev(mask.to(torch.uint8))
    out = torch.empty(rows, device="cuda")
    kk.call("kk_rowdot_fwd", inp, Pd["vp.linear.weight"], Pd["vp.linear.bias"], md, out, rows, Fv, L, 512, 0)
    close(out.view(B, L), ref, 2e-4, 2e-4, "varpred fwd")
    dy = torch.empty(rows, Fv, device="cuda")
    kk.call("kk_rowdot_bwd", dev(dout), inp, Pd["vp.linear.weight"], md, dy, Gd["vp.linear.weight"], Gd["vp.linear.bias"],
            rows, Fv, L, 512, 0, None)
    for li in (1, 0):
        col, c, y, stats, cin = acts[li]
        dc = torch.empty(rows, Fv, device="cuda")
        kk.call("kk_groupnorm_relu_bwd", dy, c, y, Pd[f"vp.norms.{li}.weight"], stats, dc, Gd[f"vp.norms.{li}.weight"],
                Gd[f"vp.norms.{li}.bias"], scratch, B, L, Fv, 512, 0.0, 0)
        kk.call("kk_gemm", 1, 1, Fv, 3 * cin, rows, 1.0, dc, Fv, col, 3 * cin, 1.0, Gd[f"vp.conv_layers.{li}.weight"], 3 * cin,
                None, None, 0, 0, 0, 0, 0)
        kk.call("kk_colsum_acc", dc, Fv, rows, Fv, Gd[f"vp.conv_layers.{li}.bias"], 0)
        dcol = torch.empty(rows, 3 * cin, device="cuda")
        kk.call("kk_gemm", 0, 1, rows, 3 * cin, Fv, 1.0, dc, Fv, Pd[f"vp.conv_layers.{li}.weight"], 3 * cin, 0.0, dcol, 3 * cin,
                None, None, 0, 0, 1, 0, 0)
        dy = torch.empty(rows, cin, device="cuda")
        kk.call("kk_im2col3_bwd", dcol, dy, B, L, cin, 512, 0)
    close(dy.view(B, L, H), xr.grad, 2e-4, 1e-3, "varpred dx")
    for n in P:
        close(Gd[n].view(P[n].shape), Pr[n].grad, 5e-4, 2e-3, f"varpred grad {n}")


def test_bucket_embed_add(kk):
    B, T, H, nb = 2, 33, 64, 256
    g = torch.Generator().manual_seed(4)
    x, pemb, eemb = torch.randn(B, T, H, generator=g), torch.randn(nb, H, generator=g), torch.randn(nb, H, generator=g)
    pitch, energy = torch.rand(B, T, generator=g), torch.rand(B, T, generator=g)
    bins = torch.linspace(0, 1, nb - 1)
    pitch[0, :5] = torch.tensor([0.0, 1.0, float(bins[7]), float(bins[100]), 0.5])     # boundary values
    lens = torch.tensor([T, 20])
    dout = torch.randn(B, T, H, generator=g)
    pr, er = pemb.clone().requires_grad_(True), eemb.clone().requires_grad_(True)
    fm = torch.arange(T)[None] >= lens[:, None]
    ref = (x + F.embedding(torch.bucketize(pitch, bins), pr) + F.embedding(torch.bucketize(energy, bins), er)
           ).masked_fill(fm[..., None], 0.0)
    ref.backward(dout)
    out = torch.empty(B, T, H, device="cuda")
    pi, ei = torch.empty(B, T, dtype=torch.int32, device="cuda"), torch.empty(B, T, dtype=torch.int32, device="cuda")
    fmd = torch.empty(B, T, dtype=torch.uint8, device="cuda")
    kk.call("kk_bucket_embed_add_fwd", dev(x), dev(pitch), dev(energy), dev(bins), dev(bins), dev(pemb), dev(eemb), dev(lens),
            out, pi, ei, fmd, B, T, H, nb, 0)
    assert torch.equal(pi.cpu().long(), torch.bucketize(pitch, bins)), "bucketize must be bit-exact"
    assert torch.equal(fmd.cpu().bool(), fm)
    close(out, ref, 1e-6, 1e-6, "bucket-embed fwd")
    dp, de = torch.zeros(nb, H, device="cuda"), torch.zeros(nb, H, device="cuda")
    kk.call("kk_bucket_embed_add_bwd", dev(dout), pi, ei, fmd, dp, de, B, T, H, nb)
    close(dp, pr.grad, 1e-4, 1e-5, "bucket-embed dpitch_emb")
    close(de, er.grad, 1e-4, 1e-5, "bucket-embed denergy_emb")


@pytest.mark.parametrize("B,T,H,nb,heavy", [(8, 512, 512, 256, 0.3), (8, 1024, 512, 256, 0.0), (3, 77, 64, 16, 0.9), (1, 5, 1028, 1024, 0.0),
                                            (2, 40, 512, 256, 1.0)])
def test_bucket_embed_bwd_segmented_sum(kk, B, T, H, nb, heavy):
    """kk_bucket_sort + kk_bucket_embed_add_bwd_sorted against float64 index_add and against the LDS form: the sort's order holds every
    unmasked frame exactly once, bin by bin; pieces cover each bin's run in order with at most 16 frames each; a bin holding `heavy` of
    all frames (unvoiced frames share pitch bin 0) is cut into many pieces; an all-masked batch and accumulation onto existing gradients."""
    g = torch.Generator().manual_seed(B * 1000 + T)
    rows = B * T
    pi, ei = torch.randint(0, nb, (rows,), generator=g, dtype=torch.int32), torch.randint(0, nb, (rows,), generator=g, dtype=torch.int32)
    pi[torch.rand(rows, generator=g) < heavy] = 0
    fm = (torch.rand(rows, generator=g) < (1.0 if heavy == 1.0 else 0.15)).to(torch.uint8)
    dout = torch.randn(rows, H, generator=g)
    keep = fm == 0
    n_items = kk.load().kk_bucket_sort_items(rows, nb)
    order = torch.full((2, rows), -1, dtype=torch.int32, device="cuda")
    items = torch.full((2, n_items, 4), -7, dtype=torch.int32, device="cuda")
    kk.call("kk_bucket_sort", dev(pi), dev(ei), dev(fm), rows, nb, order, items)
    for tb, idx in enumerate((pi, ei)):
        nk = int(keep.sum())
        o = order[tb, :nk].cpu().long()
        assert sorted(o.tolist()) == torch.nonzero(keep).flatten().tolist(), "every unmasked frame exactly once"
        assert bool((idx[o][1:] >= idx[o][:-1]).all()) if nk > 1 else True, "sorted by bin"
        it = items[tb].cpu()
        used = it[it[:, 2] > it[:, 1]]
        assert bool(((used[:, 2] - used[:, 1]) <= 16).all()) and int((used[:, 2] - used[:, 1]).sum()) == nk
        for b_, beg, end, _ in used.tolist():
            assert bool((idx[o[beg:end]] == b_).all())
        assert bool((it[it[:, 2] <= it[:, 1]][:, 1:3] == 0).all()), "unused entries are empty"
    base_p, base_e = torch.randn(nb, H, generator=g), torch.randn(nb, H, generator=g)
    ref_p, ref_e = base_p.double().clone(), base_e.double().clone()
    ref_p.index_add_(0, pi[keep].long(), dout[keep].double())
    ref_e.index_add_(0, ei[keep].long(), dout[keep].double())
    dp, de = dev(base_p), dev(base_e)
    kk.call("kk_bucket_embed_add_bwd_sorted", dev(dout), order, items, dp, de, rows, H, nb)
    close(dp, ref_p.float(), 1e-5, 2e-5, "segmented dpitch_emb")
    close(de, ref_e.float(), 1e-5, 2e-5, "segmented denergy_emb")
    if nb * 16 * 8 <= 64 * 1024:
        lp, le = dev(base_p), dev(base_e)
        kk.call("kk_bucket_embed_add_bwd", dev(dout), dev(pi), dev(ei), dev(fm), lp, le, B, T, H, nb)
        close(dp, lp, 2e-4, 2e-5, "segmented vs LDS form")            # (two fp32 summation orders of up to ~1 K rows against each other)


@pytest.mark.parametrize("B,L,C,chunk,xbf", [(8, 512, 256, 0, 0), (8, 512, 256, 0, 1), (8, 64, 256, 0, 0), (3, 437, 256, 128, 0),
                                              (5, 99, 512, 0, 1), (2, 33, 100, 0, 0)])
def test_side_branch_backward_kernels_model_shapes(kk, B, L, C, chunk, xbf):
    """The predictors' three backward kernels of the side branch at the model's shapes against float64: kk_rowdot_bwd (wave-per-row
    16-byte form; C = 100 takes... the scalar form's column ownership stays for C % 4 != 0 only), kk_groupnorm_relu_bwd (16-byte partial
    sums) and kk_bucket_embed_add_bwd (independent loads)."""
    g = torch.Generator().manual_seed(B * L + C)
    rows = B * L
    # ---- rowdot backward: dx = d w, dw = sum_r d x, db = sum_r d with masked / dead rows
    x = dev(_r16(torch.randn(rows, C, generator=g)))
    w, dout = dev(torch.randn(C, generator=g)), dev(torch.randn(rows, generator=g))
    mask = dev((torch.rand(rows, generator=g) < 0.2).to(torch.uint8))
    dx, dw, db = torch.full((rows, C), 9.0, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(1, device="cuda")
    kk.call("kk_rowdot_bwd", dout, x.bfloat16() if xbf else x, w, mask, dx, dw, db, rows, C, L, chunk, xbf, None)
    d64 = dout.double() * (mask == 0).double()
    if chunk > 0:
        l = torch.arange(rows, device="cuda") % L
        cb = (l // chunk) * chunk
        d64 = d64 * (torch.minimum(torch.tensor(chunk, device="cuda"), L - cb) >= 2).double()
    close(dx, d64[:, None] * w.double()[None], 1e-6, 1e-6, "rowdot dx")
    close(dw, (d64[:, None] * x.double()).sum(0), 2e-4, 2e-5, "rowdot dw")
    close(db, d64.sum().view(1), 2e-4, 2e-5, "rowdot db")
    if C % 4 == 0:      # ... and with one partial row per workgroup + kk_partials_reduce instead of the atomics: added onto what is there
        nb = kk.load().kk_rowdot_bwd_blocks(rows)
        part = torch.full((nb, C + 4), 7.0, device="cuda")
        dw2, db2, dx2 = torch.full((C,), 0.5, device="cuda"), torch.full((1,), 0.25, device="cuda"), torch.empty(rows, C, device="cuda")
        kk.call("kk_rowdot_bwd", dout, x.bfloat16() if xbf else x, w, mask, dx2, dw2, db2, rows, C, L, chunk, xbf, part)
        assert float(dw2[0]) == 0.5 and float(db2[0]) == 0.25, "with partial rows the launch must not touch the gradient vectors"
        kk.call("kk_partials_reduce", kk.reduce_table([(part, dw2, db2, nb, C + 1, C, C + 4)], "cuda"), 1, C + 1)
        assert torch.equal(dx2, dx)
        close(dw2 - 0.5, (d64[:, None] * x.double()).sum(0), 2e-4, 2e-5, "rowdot dw through partial rows")
        close(db2 - 0.25, d64.sum().view(1), 2e-4, 2e-5, "rowdot db through partial rows")
    # ---- GroupNorm + ReLU backward (one group per (batch item, chunk of frames))
    if C <= 256 and 256 % C == 0:
        ck = chunk if chunk > 0 else 512
        xg, gam, bet = dev(torch.randn(rows, C, generator=g)), dev(torch.rand(C, generator=g) + 0.5), dev(torch.randn(C, generator=g) * 0.1)
        nch = (L + ck - 1) // ck
        y, st = torch.empty(rows, C, device="cuda"), torch.empty(B * nch, 2, device="cuda")
        scr = torch.zeros(2 * B * nch, dtype=torch.float64, device="cuda")
        kk.call("kk_groupnorm_relu_fwd", xg, gam, bet, y, st, scr, B, L, C, ck, None, 0, 0.0)
        dy = dev(torch.randn(rows, C, generator=g))
        dxg, dg, dbt = torch.empty(rows, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        kk.call("kk_groupnorm_relu_bwd", dy, xg, y, gam, st, dxg, dg, dbt, scr, B, L, C, ck, 0.0, 0)
        x64 = xg.double().view(B, L, C).requires_grad_(True)
        g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
        outs = []
        for c0 in range(0, L, ck):
            xc = x64[:, c0:c0 + ck]
            if xc.shape[1] < 2:
                outs.append(torch.zeros_like(xc))
                continue
            mu, var = xc.mean((1, 2), keepdim=True), xc.var((1, 2), unbiased=False, keepdim=True)
            outs.append(torch.relu((xc - mu) / torch.sqrt(var + 1e-5) * g64 + b64))
        torch.cat(outs, 1).backward(dy.double().view(B, L, C))
        close(dxg, x64.grad.view(rows, C), 2e-4, 2e-4, "groupnorm dx")
        close(dg, g64.grad, 2e-3, 2e-4, "groupnorm dgamma")
        close(dbt, b64.grad, 2e-3, 2e-4, "groupnorm dbeta")
    # ---- bucket embedding backward: scatter-add of the unmasked frames' gradient rows into the pitch / energy tables
    H, nb = 512, 256
    dout2 = dev(torch.randn(rows, H, generator=g))
    pi, ei = dev(torch.randint(0, nb, (rows,), generator=g).int()), dev(torch.randint(0, nb, (rows,), generator=g).int())
    fm = dev((torch.rand(rows, generator=g) < 0.25).to(torch.uint8))
    dp, de = torch.zeros(nb, H, device="cuda"), torch.zeros(nb, H, device="cuda")
    kk.call("kk_bucket_embed_add_bwd", dout2, pi, ei, fm, dp, de, B, L, H, nb)
    d2 = dout2.double() * (fm == 0).double()[:, None]
    rp, re = torch.zeros(nb, H, dtype=torch.float64, device="cuda"), torch.zeros(nb, H, dtype=torch.float64, device="cuda")
    rp.index_add_(0, pi.long(), d2)
    re.index_add_(0, ei.long(), d2)
    close(dp, rp, 2e-4, 2e-5, "bucket-embed dpitch_emb")
    close(de, re, 2e-4, 2e-5, "bucket-embed denergy_emb")


@pytest.mark.parametrize("B,T,P,bf,aug", [(8, 512, 64, 1, 1), (8, 1024, 128, 1, 1), (3, 437, 53, 0, 1), (2, 33, 7, 1, 0)])
def test_regulate_embed_fwd_equals_three_launches(kk, B, T, P, bf, aug):
    """kk_regulate_embed_fwd == kk_length_regulate_gather + kk_bucket_embed_add_fwd + kk_specaug, bit for bit (ragged lengths, frames
    past the expansion, boundary pitch values; same seed and site: same SpecAugment masks)."""
    H, nb = 512, 256
    g = torch.Generator().manual_seed(B * T + P)
    enc = dev(torch.randn(B * P, H, generator=g))
    dur = torch.randint(0, 2 * T // P + 2, (B, P), generator=g)
    dur[0, :] = 0
    dur[0, 0] = T // 2                                   # a short sample: half of its frames are padding
    idx, lens, tot = (torch.full((B, T), -9, dtype=torch.int64, device="cuda"), torch.zeros(B, dtype=torch.int64, device="cuda"),
                      torch.zeros(B, dtype=torch.int64, device="cuda"))
    kk.call("kk_length_regulate_index", dev(dur), idx, lens, tot, B, P, T)
    pitch, energy = dev(torch.rand(B, T, generator=g)), dev(torch.rand(B, T, generator=g))
    bins = dev(torch.linspace(0, 1, nb - 1))
    pitch[0, :3] = torch.tensor([0.0, 1.0, float(bins[7])], device="cuda")
    pemb, eemb = dev(torch.randn(nb, H, generator=g)), dev(torch.randn(nb, H, generator=g))
    mdt = torch.bfloat16 if bf else torch.float32
    seed = _seed(5)
    xa, ma = torch.empty(B * T, H, device="cuda"), torch.empty(B * T, H, device="cuda", dtype=mdt)
    pa, ea, fa = (torch.empty(B, T, dtype=torch.int32, device="cuda"), torch.empty(B, T, dtype=torch.int32, device="cuda"),
                  torch.empty(B, T, dtype=torch.uint8, device="cuda"))
    kk.call("kk_length_regulate_gather", enc, idx, xa, B, P, T, H)
    kk.call("kk_bucket_embed_add_fwd", xa, pitch, energy, bins, bins, pemb, eemb, lens, ma, pa, ea, fa, B, T, H, nb, bf)
    if aug:
        kk.call("kk_specaug", ma, B, T, H, seed, 20, 30, 40, 2, 2, bf)
    xb, mb = torch.full_like(xa, 3.0), torch.full_like(ma, 3.0)
    pb, eb, fb = torch.full_like(pa, -1), torch.full_like(ea, -1), torch.full_like(fa, 9)
    kk.call("kk_regulate_embed_fwd", enc, idx, pitch, energy, bins, bins, pemb, eemb, lens, xb, mb, pb, eb, fb, B, P, T, H, nb, bf,
            seed if aug else None, 20, 30, 40, 2, 2)
    torch.cuda.synchronize()
    assert torch.equal(xb, xa) and torch.equal(mb, ma) and torch.equal(pb, pa) and torch.equal(eb, ea) and torch.equal(fb, fa)
    if aug:      # the masks really fired: some unmasked frames have zeroed columns / whole zero rows
        live = (fa.view(-1) == 0)
        assert float((mb[live].float() == 0).float().mean()) > 0.02


def test_small_helpers(kk):
    ids = torch.tensor([[0, 3, 0, 5]])
    m = torch.empty(1, 4, dtype=torch.uint8, device="cuda")
    kk.call("kk_ids_eq_zero", dev(ids), m, 4)
    assert m.cpu().tolist() == [[1, 0, 1, 0]]
    mel = torch.randn(2, 5, 8)
    out = torch.empty(2, 5, 8, device="cuda")
    kk.call("kk_shift_right", dev(mel), out, 2, 5, 8)
    assert torch.equal(out.cpu(), F.pad(mel[:, :-1], (0, 0, 1, 0)))


@pytest.mark.parametrize("ragged", [False, True])
def test_losses_fwd_bwd(kk, ragged):
    from kokoro_ruslan_amd.lib import KkLossCfg
    d = O.ModelDims()
    B, T, Pn = 3, 50, 9
    b = O.synthetic_batch(B, T, Pn, d, seed=5, ragged=ragged)
    g = torch.Generator().manual_seed(1)
    out = {"mel": torch.randn(B, T, 80, generator=g) - 5, "log_dur": torch.randn(B, Pn, generator=g) + 1.5,
           "stop": torch.randn(B, T, generator=g) * 3, "pitch": torch.rand(B, T, generator=g),
           "energy": torch.rand(B, T, generator=g)}
    if ragged:
        out["mel"][0, 3, 4] = float("nan")            # non-finite element inside a valid frame is filtered
        out["mel"][1, T - 1, 0] = float("inf")        # padded frame
    outr = {k: v.clone().requires_grad_(True) for k, v in out.items()}
    hp = O.StepHyper()
    ls = O.losses(outr, b, hp)
    (ls[0] * 0.5).backward()
    cfg = KkLossCfg(hp.duration_loss_weight, hp.stop_token_loss_weight, hp.pitch_loss_weight, hp.energy_loss_weight,
                    hp.duration_huber_delta, hp.pitch_huber_delta, hp.energy_huber_delta, hp.stop_token_pos_weight, 0.5, 0)
    acc = torch.zeros(12, dtype=torch.float64, device="cuda")
    losses, coef = torch.zeros(6, device="cuda"), torch.zeros(5, device="cuda")
    dv = {k: dev(v) for k, v in out.items()}
    bd = {k: dev(v) for k, v in b.items()}
    args = (dv["mel"], bd["mel_specs"], dv["log_dur"], bd["phoneme_durations"], dv["stop"], bd["stop_token_targets"], dv["pitch"],
            bd["pitches"], dv["energy"], bd["energies"], bd["mel_lengths"], bd["phoneme_lengths"], B, T, Pn, 80, cfg)
    kk.call("kk_losses_fwd", *args, None, acc, losses, coef, None, 0)
    close(losses, torch.stack([x.detach() for x in ls]), 1e-5, 1e-5, "losses")
    grads = [torch.empty_like(dv[k]) for k in ("mel", "log_dur", "stop", "pitch", "energy")]
    kk.call("kk_losses_bwd", *args, coef, *grads)
    for k, gd in zip(("mel", "log_dur", "stop", "pitch", "energy"), grads):
        ref = torch.nan_to_num(outr[k].grad, nan=0.0, posinf=0.0, neginf=0.0)
        close(gd, ref, 1e-7, 1e-4, f"loss grad {k}")
    # the accumulator handed round zero (KK_LOSS_ACC_ZEROED | KK_LOSS_ACC_CLEAR): same scalars, acc zero again afterwards; and the
    # data-parallel order: forward leaves acc for the collective, the explicit finalize clears it
    assert float(acc.abs().sum()) > 0
    l0, c0 = losses.clone(), coef.clone()
    acc.zero_(); losses.zero_(); coef.zero_()
    for _ in range(2):
        kk.call("kk_losses_fwd", *args, None, acc, losses, coef, None, 3)
        assert torch.equal(losses, l0) and torch.equal(coef, c0) and float(acc.abs().sum()) == 0.0
    kk.call("kk_losses_fwd", *args, None, acc, losses, coef, None, 1)
    assert float(acc.abs().sum()) > 0
    losses.zero_()
    kk.call("kk_losses_finalize", acc, cfg, None, T, losses, coef, None, 1)
    assert torch.equal(losses, l0) and torch.equal(coef, c0) and float(acc.abs().sum()) == 0.0


def test_optimizer_accumulators_handed_round_zero(kk):
    """kk_seg_sumsq (order-deterministic since round 6) / kk_opt_prepare(clear_a, clear_b) / kk_adamw_ema(zeroed = 1): the per-segment accumulators
    kept zero by the one-workgroup launch between their writers give the same numbers as the zero-fill launches they replace."""
    g = torch.Generator().manual_seed(5)
    BLK, nblocks, nseg = 1024, 96, 7
    seg_of = torch.tensor(sorted([int(v) for v in torch.randint(0, nseg, (nblocks,), generator=g)]), dtype=torch.int32, device="cuda")
    n = nblocks * BLK
    grad = dev(torch.randn(n, generator=g) * 0.01)
    ref = torch.zeros(nseg, dtype=torch.float64, device="cuda")
    ref.index_add_(0, seg_of.long().repeat_interleave(BLK), grad.double() ** 2)
    ws = torch.empty(kk.load().kk_seg_sumsq_ws_bytes(nblocks), dtype=torch.uint8, device="cuda")
    ws.fill_(0xAB)                                                        # (the record workspace needs no initial state)
    ss = torch.full((nseg,), 7.0, dtype=torch.float64, device="cuda")
    kk.call("kk_seg_sumsq", grad, seg_of, nblocks, ss, nseg, ws, None, 0)          # every segment is STORED: garbage in the output is harmless
    close(ss, ref, 1e-6, 1e-9, "seg_sumsq (stores every segment, no zero-fill)")
    # round 6: a pure function of the buffer — no atomics, the partials merged in arena order by a fixed tree.
    # A big arena (every workgroup of the launch busy, segments of 1 block up to thousands, boundaries at every position inside a
    # workgroup's range): 30 calls give the SAME BITS, and they are the fp64 reference's value to rounding.
    gb = torch.Generator().manual_seed(11)
    lens = [1, 1, 3, 700, 1, 2, 1500, 1, 1, 64, 5000, 1, 23, 24, 25, 1, 1, 3000, 2, 9000, 1, 1, 1, 777, 48, 1]
    lens = lens * 3
    nb2, ns2 = sum(lens), len(lens)
    seg2 = torch.repeat_interleave(torch.arange(ns2, dtype=torch.int32), torch.tensor(lens)).to("cuda")
    big = (torch.randn(nb2 * BLK, generator=gb) * torch.rand(nb2 * BLK, generator=gb) * 3.0).to("cuda")
    ref2 = torch.zeros(ns2, dtype=torch.float64, device="cuda")
    ref2.index_add_(0, seg2.long().repeat_interleave(BLK), big.double() ** 2)
    ws2 = torch.empty(kk.load().kk_seg_sumsq_ws_bytes(nb2), dtype=torch.uint8, device="cuda")
    outs = []
    for i in range(30):
        o = torch.full((ns2,), float("nan"), dtype=torch.float64, device="cuda")
        ws2.random_(0, 255)
        kk.call("kk_seg_sumsq", big, seg2, nb2, o, ns2, ws2, None, 0)
        outs.append(o)
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "kk_seg_sumsq must give the same bits on every call (order-deterministic reduction)"
    assert float(((outs[0] - ref2).abs() / ref2).max()) < 1e-7, "seg_sumsq, 78 segments over ~60 K blocks, against the fp64 reference (squares are fp32 products)"
    # an inf / NaN gradient poisons exactly its own segment
    bad = big.clone()
    bad[(sum(lens[:6]) + 700) * BLK + 5] = float("inf")
    o = torch.zeros(ns2, dtype=torch.float64, device="cuda")
    kk.call("kk_seg_sumsq", bad, seg2, nb2, o, ns2, ws2, None, 0)
    fin = torch.isfinite(o)
    assert int((~fin).sum()) == 1 and not bool(fin[6]) and torch.equal(o[fin], outs[0][fin])
    f = lambda v: torch.full((nseg,), v, device="cuda")
    cfg = kk.KkOptCfg()
    cfg.learning_rate, cfg.max_lr, cfg.warmup_start_lr, cfg.warmup_target_lr, cfg.use_warmup = 1e-3, 1e-3, 1e-4, 1e-3, 0
    cfg.pct_start, cfg.div_factor, cfg.final_div_factor, cfg.warmup_steps, cfg.onecycle_steps = 0.3, 25.0, 1e4, 10, 1000
    cfg.max_grad_norm, cfg.beta1, cfg.beta2, cfg.eps, cfg.mel_length = 1.0, 0.9, 0.999, 1e-8, 64
    cfg.expl_abs_floor, cfg.expl_warmup_floor, cfg.expl_multiplier, cfg.expl_alpha = 1e9, 1e9, 10.0, 0.9
    cfg.expl_warmup_steps, cfg.expl_min_ema_steps, cfg.ema_decay, cfg.max_weight_norm = 0, 5, 0.999, 0.0
    st = torch.zeros(16, dtype=torch.float64, device="cuda")
    gs, dec, stp, consts = f(0.0), f(0.0), f(0.0), torch.zeros(4, device="cuda")
    psq = torch.full((nseg,), 3, dtype=torch.int64, device="cuda")         # (Q34.30 fixed point since round 6)
    kk.call("kk_opt_prepare", ss, f(0.0), f(1.0), f(0.01), nseg, None, cfg, st, gs, dec, stp, consts, ss, psq)
    assert float(ss.abs().sum()) == 0.0 and int(psq.abs().sum()) == 0, "kk_opt_prepare leaves both accumulators zero"
    total = float(ref.sum().sqrt())
    close(torch.tensor(float(st[kk.OS["LAST_GRAD_NORM"]])), torch.tensor(total), 1e-6, 1e-9, "total norm read before the clear")
    p, m, v, ema = dev(torch.randn(n, generator=g)), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    flags = torch.full((nseg,), 1 | 4, dtype=torch.int32, device="cuda")
    pa, pb = p.clone(), p.clone()
    psa = torch.full((nseg,), 9, dtype=torch.int64, device="cuda")
    kk.call("kk_adamw_ema", pa, grad, m.clone(), v.clone(), ema.clone(), seg_of, nblocks, gs, dec, stp, flags, consts, 0.9, 0.999, 0.999, psa, nseg, None, 0)
    kk.call("kk_adamw_ema", pb, grad, m.clone(), v.clone(), ema.clone(), seg_of, nblocks, gs, dec, stp, flags, consts, 0.9, 0.999, 0.999, psq, nseg, None, 1)
    assert torch.equal(pa, pb) and not torch.equal(pa, p)
    assert torch.equal(psq, psa), "p_sumsq (zero on entry == zero-fill inside): integer sums, bit for bit"
    want = torch.zeros(nseg, dtype=torch.float64, device="cuda")
    want.index_add_(0, seg_of.long().repeat_interleave(BLK), pa.double() ** 2)
    close(psq.double() / 2.0 ** 30, want, 1e-6, 1e-6, "p_sumsq = Q34.30 sum of the post-step squares")
    for _ in range(5):                                                     # order-independent: repeated passes agree bit for bit
        pc, psc = p.clone(), torch.zeros(nseg, dtype=torch.int64, device="cuda")
        kk.call("kk_adamw_ema", pc, grad, m.clone(), v.clone(), ema.clone(), seg_of, nblocks, gs, dec, stp, flags, consts, 0.9, 0.999, 0.999, psc, nseg, None, 1)
        assert torch.equal(psc, psq)


# ----------------------------------------------------------------------------------------------------------------
# Dropout / DropPath / SpecAugment: bit-wise parity with the reference's CPU RNG stream is impossible,
# so these check what matters: mask values in {0, 1/(1-p)}, keep rate, per-sample DropPath, and above all that the
# BACKWARD kernels regenerate exactly the forward's mask.
def _seed(v=1234):
    return torch.tensor([v], dtype=torch.int32, device="cuda")


def test_dropout_residual_droppath(kk):
    rows, H, S = 64 * 40, 128, 40                      # 64 samples of 40 rows
    g = torch.Generator().manual_seed(0)
    x, res = torch.rand(rows, H, generator=g) + 1.0, torch.randn(rows, H, generator=g)     # |x| >= 1: clean mask recovery
    xd, rd, out = dev(x), dev(res), torch.empty(rows, H, device="cuda")
    p1, p2, dp = 0.2, 0.1, 0.25
    kk.call("kk_dropout_fwd", xd, rd, 0, out, rows, H, S, _seed(), 5, p1, 6, p2, 7, dp)
    m = ((out.cpu() - res) / x)
    full = 1.0 / ((1 - p1) * (1 - p2) * (1 - dp))
    assert bool(((m.abs() < 1e-5) | ((m - full).abs() < 1e-3)).all()), "mask values must be 0 or 1/keep"
    per_sample = (m.view(64, S * H) != 0).any(1)
    assert 30 <= int(per_sample.sum()) <= 60, "DropPath drops whole samples at ~25 %"
    kept = m.view(64, -1)[per_sample]
    rate = float((kept > 0).float().mean())
    assert abs(rate - (1 - p1) * (1 - p2)) < 0.01, rate
    dx = torch.empty(rows, H, device="cuda")
    kk.call("kk_dropout_bwd", xd, dx, rows, H, S, _seed(), 5, p1, 6, p2, 7, dp, 0)
    dx16 = torch.empty(rows, H, device="cuda", dtype=torch.bfloat16)
    kk.call("kk_dropout_bwd", xd, dx16, rows, H, S, _seed(), 5, p1, 6, p2, 7, dp, 1)
    assert torch.equal(dx16, dx.bfloat16()), "bf16 output = rounded fp32 output"
    assert torch.equal(dx.cpu(), (out - rd).cpu()) or float((dx.cpu() - (out.cpu() - res)).abs().max()) < 1e-5
    out2 = torch.empty_like(out)
    kk.call("kk_dropout_fwd", xd, rd, 0, out2, rows, H, S, _seed(), 5, p1, 6, p2, 7, dp)
    assert torch.equal(out, out2), "same seed, same mask"
    kk.call("kk_dropout_fwd", xd, rd, 0, out2, rows, H, S, _seed(99), 5, p1, 6, p2, 7, dp)
    assert not torch.equal(out, out2), "new seed, new mask"
    pe = torch.randn(S, H, generator=g)                # row-periodic residual (decoder input + PE)
    kk.call("kk_dropout_fwd", xd, dev(pe), S, out2, rows, H, S, _seed(), 5, 0.0, 0, 0.0, 0, 0.0)
    close(out2, x + pe.repeat(64, 1), 1e-6, 1e-6, "p=0 periodic residual")


def test_fused_dropout_masks_match_between_forward_and_backward(kk):
    g = torch.Generator().manual_seed(1)
    # GLU
    rows, Fd, p = 300, 96, 0.2
    h, dg = torch.randn(rows, 2 * Fd, generator=g) + 1.0, torch.randn(rows, Fd, generator=g)
    g0, g1 = torch.empty(rows, Fd, device="cuda"), torch.empty(rows, Fd, device="cuda")
    kk.call("kk_glu_fwd", dev(h), g0, rows, Fd, None, 0, 0.0, 0)
    kk.call("kk_glu_fwd", dev(h), g1, rows, Fd, _seed(), 3, p, 0)
    mask = torch.where(g0.abs() > 1e-6, g1 / g0, torch.ones_like(g0)).cpu()
    assert bool(((mask.abs() < 1e-5) | ((mask - 1 / (1 - p)).abs() < 1e-4)).all())
    assert abs(float((mask > 0).float().mean()) - (1 - p)) < 0.02
    d0, d1 = torch.empty(rows, 2 * Fd, device="cuda"), torch.empty(rows, 2 * Fd, device="cuda")
    kk.call("kk_glu_bwd", dev(dg * mask), dev(h), d0, rows, Fd, None, 0, 0.0, 0)       # explicit mask, no RNG
    kk.call("kk_glu_bwd", dev(dg), dev(h), d1, rows, Fd, _seed(), 3, p, 0)              # regenerated mask
    close(d1, d0, 1e-6, 1e-5, "glu backward regenerates the forward mask")
    # embedding + PE dropout
    B, P, H, V = 4, 30, 64, 59
    ids, stress = torch.randint(1, V, (B, P), generator=g), torch.randint(0, 3, (B, P), generator=g)
    emb, semb, pe = torch.randn(V, H, generator=g), torch.randn(3, H, generator=g), torch.randn(P, H, generator=g)
    o0, o1 = torch.empty(B, P, H, device="cuda"), torch.empty(B, P, H, device="cuda")
    args = (dev(ids), dev(stress), dev(emb), dev(semb), dev(pe))
    kk.call("kk_embed_fwd", *args, o0, B, P, H, 8.0, None, 0, 0.0)
    kk.call("kk_embed_fwd", *args, o1, B, P, H, 8.0, _seed(), 1, 0.15)
    mask = (o1 / o0).cpu()
    assert abs(float((mask > 0).float().mean()) - 0.85) < 0.03
    dout = torch.randn(B, P, H, generator=g)
    e0, s0, e1, s1 = (torch.zeros(V, H, device="cuda"), torch.zeros(3, H, device="cuda"), torch.zeros(V, H, device="cuda"),
                      torch.zeros(3, H, device="cuda"))
    kk.call("kk_embed_bwd", dev(ids), dev(stress), dev(dout * mask), e0, s0, B, P, H, 8.0, None, 0, 0.0)
    kk.call("kk_embed_bwd", dev(ids), dev(stress), dev(dout), e1, s1, B, P, H, 8.0, _seed(), 1, 0.15)
    close(e1, e0, 1e-4, 1e-5, "embed backward mask")
    # GroupNorm + ReLU + dropout: backward only needs 1/(1-p) because dropped elements are stored as zeros
    Bn, L, C, pv = 2, 50, 32, 0.1
    x, gam, bet = torch.randn(Bn * L, C, generator=g), torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    y0, y1, st = torch.empty(Bn * L, C, device="cuda"), torch.empty(Bn * L, C, device="cuda"), torch.empty(Bn, 2, device="cuda")
    scr = torch.zeros(2 * Bn, dtype=torch.float64, device="cuda")
    kk.call("kk_groupnorm_relu_fwd", dev(x), dev(gam), dev(bet), y0, st, scr, Bn, L, C, 512, None, 0, 0.0)
    kk.call("kk_groupnorm_relu_fwd", dev(x), dev(gam), dev(bet), y1, st, scr, Bn, L, C, 512, _seed(), 2, pv)
    pos = y0 > 1e-6
    mask = torch.where(pos, y1 / y0.clamp(min=1e-6), torch.ones_like(y0)).cpu()
    assert abs(float((mask[pos.cpu()] > 0).float().mean()) - (1 - pv)) < 0.03
    dy = torch.randn(Bn * L, C, generator=g)
    a0, a1 = torch.empty(Bn * L, C, device="cuda"), torch.empty(Bn * L, C, device="cuda")
    gg0, gb0, gg1, gb1 = (torch.zeros(C, device="cuda") for _ in range(4))
    kk.call("kk_groupnorm_relu_bwd", dev(dy * mask), dev(x), y0, dev(gam), st, a0, gg0, gb0, scr, Bn, L, C, 512, 0.0, 0)
    kk.call("kk_groupnorm_relu_bwd", dev(dy), dev(x), y1, dev(gam), st, a1, gg1, gb1, scr, Bn, L, C, 512, pv, 0)
    close(a1, a0, 1e-5, 1e-4, "groupnorm dropout backward")
    close(gg1, gg0, 1e-4, 1e-4, "groupnorm dropout dgamma")


@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("causal", [0, 1])
def test_attention_probability_dropout(kk, math_mode, causal):
    """V = identity (Sk = 64 = head_dim) turns the output into the dropped probability matrix itself, which recovers the
    forward mask; the backward kernels must reproduce a torch autograd run that uses that explicit mask."""
    B, h, S, p = 2, 2, 64, 0.2
    H = h * 64
    g = torch.Generator().manual_seed(7 + causal)
    Q, K = torch.randn(B, S, H, generator=g), torch.randn(B, S, H, generator=g)
    V = torch.eye(64).repeat(B, 1, h).contiguous()                     # [B, 64, h*64]: every head's V is I
    Qd, Kd, Vd = dev(Q), dev(K), dev(V)
    O0, O1 = torch.zeros(B, S, H, device="cuda"), torch.zeros(B, S, H, device="cuda")
    lse = torch.zeros(B, h, S, device="cuda")
    kk.call("kk_attn_fwd", Qd, Kd, Vd, O0, lse, B, h, S, S, H, H, H, H, None, causal, 0.125, None, 0, 0.0, math_mode, 0)
    kk.call("kk_attn_fwd", Qd, Kd, Vd, O1, lse, B, h, S, S, H, H, H, H, None, causal, 0.125, _seed(5), 9, p, math_mode, 0)
    P0, P1 = O0.cpu().view(B, S, h, 64).transpose(1, 2), O1.cpu().view(B, S, h, 64).transpose(1, 2)     # [B,h,q,key]
    big = P0 > 1e-3
    ratio = (P1 / P0.clamp(min=1e-9))[big]
    tol = 1e-3 if math_mode == 0 else 3e-2
    assert bool(((ratio.abs() < tol) | ((ratio - 1 / (1 - p)).abs() < 1.25 * tol / (1 - p) + tol)).all()), "mask values"
    assert abs(float((ratio > 0.5).float().mean()) - (1 - p)) < 0.03, "keep rate"
    mask = torch.where(big, (P1 / P0.clamp(min=1e-9) > 0.5).float() / (1 - p), torch.full_like(P0, 1 / (1 - p)))
    # general V for the backward check, explicit-mask reference in fp64
    V2, dO = torch.randn(B, S, H, generator=g), torch.randn(B, S, H, generator=g)
    Qr, Kr, Vr = (t.clone().double().requires_grad_(True) for t in (Q, K, V2))
    if math_mode:
        Qr, Kr, Vr = (t.detach().bfloat16().double().requires_grad_(True) for t in (Q, K, V2))
    hd = lambda x: x.view(B, S, h, 64).transpose(1, 2)
    s = hd(Qr) @ hd(Kr).transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.triu(torch.full((S, S), float("-inf"), dtype=s.dtype), 1)
    ref = ((torch.softmax(s, -1) * mask.double()) @ hd(Vr)).transpose(1, 2).reshape(B, S, H)
    ref.backward(dO.double())
    V2d, dOd = dev(V2), dev(dO)
    O2 = torch.zeros(B, S, H, device="cuda")
    kk.call("kk_attn_fwd", Qd, Kd, V2d, O2, lse, B, h, S, S, H, H, H, H, None, causal, 0.125, _seed(5), 9, p, math_mode, 0)
    small = big.sum() == big.numel()       # the recovered mask is exact only where P0 is not tiny
    atol, rtol = (1e-2, 3e-3) if math_mode == 0 else (8e-2, 5e-2)   # mask unknown where P0 < 1e-3: error <= 1e-3*|V|/(1-p)
    close(O2, ref, atol, rtol, "attn fwd with dropout")
    delta = torch.zeros(B, h, S, device="cuda")
    kk.call("kk_attn_delta", O2, dOd, delta, B, h, S, H, H, 0)
    dQ, dK, dV = torch.zeros_like(Qd), torch.zeros_like(Kd), torch.zeros_like(V2d)
    kk.call("kk_attn_bwd_dq", Qd, Kd, V2d, dOd, lse, delta, dQ, B, h, S, S, H, H, H, H, H, None, causal, 0.125, _seed(5), 9, p, math_mode, 0, None, 0, None)
    kk.call("kk_attn_bwd_dkv", Qd, Kd, V2d, dOd, lse, delta, dK, dV, B, h, S, S, H, H, H, H, H, H, None, causal, 0.125,
            _seed(5), 9, p, math_mode, 0, None)
    atol, rtol = (2e-2, 1e-2) if math_mode == 0 else (0.15, 0.1)
    close(dV, Vr.grad, atol, rtol, "attn dV with dropout")
    close(dQ, Qr.grad, atol, rtol, "attn dQ with dropout")
    close(dK, Kr.grad, atol, rtol, "attn dK with dropout")


def test_specaugment_mask_structure(kk):
    B, T, H = 16, 100, 128
    x = torch.ones(B, T, H, device="cuda")
    kk.call("kk_specaug", x, B, T, H, _seed(3), 20, 5, 3, 1, 2, 0)
    z = (x.cpu() == 0)
    for b in range(B):
        rows = z[b].all(1)                              # fully masked frames = the time mask
        assert int(rows.sum()) < 5                      # t in [0, min(5, T//4))
        idx = rows.nonzero().flatten()
        assert idx.numel() == 0 or int(idx[-1] - idx[0]) == idx.numel() - 1, "time mask is one contiguous span"
        cols = z[b][~rows].all(0) if bool((~rows).any()) else z[b].all(0)
        assert int(cols.sum()) <= 4                     # two feature masks of f in [0, 3) dims
    assert 0 < int(z.any(-1).any(-1).sum()), "some samples are masked"
    y = torch.ones(B, T, H, device="cuda")
    kk.call("kk_specaug", y, B, T, H, _seed(3), 20, 5, 3, 1, 2, 0)
    assert torch.equal(x, y)                            # the gradient pass sees the same mask


# ---------------------------------------------------------------------------------------------------------------
# bf16 activation storage: every kernel that takes a *_bf16 flag must agree with its own fp32-storage instantiation
# run on the same (bf16-representable) inputs, up to one bf16 rounding of the outputs.
def _r16(t):
    return t.bfloat16().float()


BF = (2e-2, 1e-2)   # (atol, rtol): one bf16 rounding (2^-8 relative) of O(1) outputs, with slack for re-associated sums


@pytest.mark.parametrize("B,h,Sq,Sk,causal,masked,strided", [(2, 2, 64, 64, 0, 1, 0), (8, 8, 64, 64, 0, 1, 1), (3, 2, 47, 47, 0, 1, 1), (2, 2, 33, 64, 1, 0, 0),
                                                              (1, 2, 200, 200, 1, 0, 1),
                                                              (2, 1, 37, 150, 0, 1, 0), (1, 8, 300, 300, 1, 0, 1),
                                                              (1, 2, 1100, 1100, 1, 1, 1), (1, 2, 777, 1030, 0, 1, 0),
                                                              # dynamic batching's shape class: >= 512 row blocks of a RAGGED sequence -> attn_fwd3_q128
                                                              (12, 8, 1333, 1333, 1, 0, 1), (12, 8, 1333, 1333, 0, 1, 1)])
def test_attention_bf16_storage(kk, B, h, Sq, Sk, causal, masked, strided):
    g = torch.Generator().manual_seed(Sq + 7 * Sk + causal)
    H = h * 64
    ld = 3 * H if strided else H          # fused-qkv row stride
    big = [_r16(torch.randn(B, S, ld, generator=g)) for S in (Sq, Sk, Sk)]
    Q, K, V = (dev(t)[..., :H] if not strided else dev(t)[..., i * H:(i + 1) * H] for i, t in enumerate(big))
    if not strided:
        Q, K, V = (t.contiguous() for t in (Q, K, V))
    dO = dev(_r16(torch.randn(B, Sq, H, generator=g)))
    kmd = None
    if masked:
        km = torch.rand(B, Sk, generator=g) < 0.3
        km[:, 0] = False
        kmd = dev(km.to(torch.uint8))
    Q16, K16, V16 = (dev(t).bfloat16()[..., :H] if not strided else dev(t).bfloat16()[..., i * H:(i + 1) * H] for i, t in enumerate(big))
    if not strided:
        Q16, K16, V16 = (t.contiguous() for t in (Q16, K16, V16))
    dO16 = dO.bfloat16()
    seed = torch.tensor([11], dtype=torch.int32, device="cuda")
    for p in (0.0, 0.2):
        O32, O16 = torch.zeros(B, Sq, H, device="cuda"), torch.zeros(B, Sq, H, device="cuda", dtype=torch.bfloat16)
        l32, l16 = torch.zeros(B, h, Sq, device="cuda"), torch.zeros(B, h, Sq, device="cuda")
        kk.call("kk_attn_fwd", Q, K, V, O32, l32, B, h, Sq, Sk, ld, ld, ld, H, kmd, causal, 0.125, seed, 4, p, 1, 0)
        kk.call("kk_attn_fwd", Q16, K16, V16, O16, l16, B, h, Sq, Sk, ld, ld, ld, H, kmd, causal, 0.125, seed, 4, p, 1, 1)
        if Sk > 128:                                       # the route is part of what the case covers (kk_attn.hip: kk_attn_fwd)
            assert kk.last_kernel() == ("attn_fwd3_q128" if -(-Sq // 128) * B * h >= 512 else "attn_fwd3_q64"), kk.last_kernel()
        close(O16, O32, *BF, f"attn fwd bf16 storage p={p}")
        close(l16, l32, 1e-5, 1e-5, "attn lse bf16 storage")
        O32r = O16.float()                                 # backward from the same (rounded) forward output
        d32, d16 = torch.zeros(B, h, Sq, device="cuda"), torch.zeros(B, h, Sq, device="cuda")
        kk.call("kk_attn_delta", O32r, dO, d32, B, h, Sq, H, H, 0)
        kk.call("kk_attn_delta", O16, dO16, d16, B, h, Sq, H, H, 1)
        close(d16, d32, 1e-4, 1e-5, "attn delta bf16 storage")
        g32 = [torch.zeros(B, S, H, device="cuda") for S in (Sq, Sk, Sk)]
        g16 = [torch.zeros(B, S, H, device="cuda", dtype=torch.bfloat16) for S in (Sq, Sk, Sk)]
        kk.call("kk_attn_bwd_dq", Q, K, V, dO, l32, d32, g32[0], B, h, Sq, Sk, ld, ld, ld, H, H, kmd, causal, 0.125, seed, 4, p, 1, 0, None, 0, None)
        kk.call("kk_attn_bwd_dq", Q16, K16, V16, dO16, l16, d16, g16[0], B, h, Sq, Sk, ld, ld, ld, H, H, kmd, causal, 0.125, seed, 4, p, 1, 1, None, 0, None)
        kk.call("kk_attn_bwd_dkv", Q, K, V, dO, l32, d32, g32[1], g32[2], B, h, Sq, Sk, ld, ld, ld, H, H, H, kmd, causal, 0.125,
                seed, 4, p, 1, 0, None)
        kk.call("kk_attn_bwd_dkv", Q16, K16, V16, dO16, l16, d16, g16[1], g16[2], B, h, Sq, Sk, ld, ld, ld, H, H, H, kmd, causal, 0.125,
                seed, 4, p, 1, 1, None)
        for a, b, n in zip(g16, g32, "QKV"):
            close(a, b, 3e-2, 1e-2, f"attn d{n} bf16 storage p={p}")


def test_norms_bf16_storage(kk):
    g = torch.Generator().manual_seed(5)
    rows, H = 333, 512
    x, dy = dev(torch.randn(rows, H, generator=g)), dev(_r16(torch.randn(rows, H, generator=g)))
    gam, bet = dev(1 + 0.1 * torch.randn(H, generator=g)), dev(0.1 * torch.randn(H, generator=g))
    # LayerNorm: x fp32 (residual stream), y / dy in bf16
    y32, y16 = torch.empty(rows, H, device="cuda"), torch.empty(rows, H, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    kk.call("kk_layernorm_fwd", x, gam, bet, y32, mean, rstd, rows, H, 0)
    kk.call("kk_layernorm_fwd", x, gam, bet, y16, mean, rstd, rows, H, 1)
    assert torch.equal(y16, y32.bfloat16())
    out = []
    for flag, d in ((0, dy), (1, dy.bfloat16())):
        dx, dg, db = torch.ones(rows, H, device="cuda"), torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
        kk.call("kk_layernorm_bwd", d, x, gam, mean, rstd, dx, 1, dg, db, None, rows, H, flag)
        out.append((dx, dg, db))
    for a, b, n in zip(out[1], out[0], ("dx", "dgamma", "dbeta")):
        close(a, b, 1e-5, 1e-5, f"ln bwd {n} with bf16 dy")
    # RMSNorm: x (and dx) bf16, y / dy fp32
    xr = _r16(x)
    gain, res = gam, dev(torch.randn(rows, H, generator=g))
    ya, yb = torch.empty(rows, H, device="cuda"), torch.empty(rows, H, device="cuda")
    ra, rb = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    kk.call("kk_rmsnorm_fwd", xr, gain, res, ya, ra, rows, H, 0)
    kk.call("kk_rmsnorm_fwd", xr.bfloat16(), gain, res, yb, rb, rows, H, 1)
    close(yb, ya, 1e-6, 1e-6, "rms fwd bf16 x")
    dxa, dxb = torch.empty(rows, H, device="cuda"), torch.empty(rows, H, device="cuda", dtype=torch.bfloat16)
    dga, dgb = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    kk.call("kk_rmsnorm_bwd", dy, xr, gain, ra, dxa, dga, None, rows, H, 0)
    kk.call("kk_rmsnorm_bwd", dy, xr.bfloat16(), gain, rb, dxb, dgb, None, rows, H, 1)
    close(dxb, dxa, *BF, "rms dx bf16")
    close(dgb, dga, 1e-4, 1e-5, "rms dgain bf16 x")


@pytest.mark.parametrize("rope", [0, 1])
def test_headnorm_bf16_storage(kk, rope):
    g = torch.Generator().manual_seed(9 + rope)
    B, S, h = 2, 50, 4
    H = h * 64
    x, dy = dev(_r16(torch.randn(B * S, 3 * H, generator=g))), dev(_r16(torch.randn(B * S, 3 * H, generator=g)))
    gains = [dev(1 + 0.1 * torch.randn(64, generator=g)) for _ in range(3)]
    c, s_ = O.rope_tables(S, 64)
    ct, st_ = dev(c), dev(s_)
    y32, y16 = torch.empty_like(x), torch.empty_like(x, dtype=torch.bfloat16)
    kk.call("kk_headnorm_rope_fwd", x, 3 * H, y32, 3 * H, B * S, h, S, 3, *gains, 3 if rope else 0, ct, st_, 0)
    kk.call("kk_headnorm_rope_fwd", x.bfloat16(), 3 * H, y16, 3 * H, B * S, h, S, 3, *gains, 3 if rope else 0, ct, st_, 1)
    close(y16, y32, *BF, "headnorm fwd bf16")
    dx32, dx16 = torch.empty_like(x), torch.empty_like(x, dtype=torch.bfloat16)
    dg32, dg16 = [torch.zeros(64, device="cuda") for _ in range(3)], [torch.zeros(64, device="cuda") for _ in range(3)]
    kk.call("kk_headnorm_rope_bwd", dy, 3 * H, x, 3 * H, dx32, 3 * H, B * S, h, S, 3, *gains, *dg32, None, 3 if rope else 0, ct, st_, 0)
    nb = kk.load().kk_headnorm_bwd_blocks(B * S, h)
    part = torch.full((3, nb, 64), 5.0, device="cuda")           # bf16 run: gain gradients through partial rows + reduce
    kk.call("kk_headnorm_rope_bwd", dy.bfloat16(), 3 * H, x.bfloat16(), 3 * H, dx16, 3 * H, B * S, h, S, 3, *gains, *dg16,
            part, 3 if rope else 0, ct, st_, 1)
    kk.call("kk_partials_reduce", kk.reduce_table([(part[j], dg16[j], None, nb, 64, 64) for j in range(3)], "cuda"), 3, 64)
    close(dx16, dx32, *BF, "headnorm dx bf16")
    for a, b in zip(dg16, dg32):
        close(a, b, 1e-3, 1e-4, "headnorm dgain bf16")


def test_elementwise_bf16_storage(kk):
    g = torch.Generator().manual_seed(21)
    rows, Fd = 300, 256
    seed = torch.tensor([3], dtype=torch.int32, device="cuda")
    hh, dg = dev(_r16(torch.randn(rows, 2 * Fd, generator=g))), dev(_r16(torch.randn(rows, Fd, generator=g)))
    for p in (0.0, 0.1):
        g32, g16 = torch.empty(rows, Fd, device="cuda"), torch.empty(rows, Fd, device="cuda", dtype=torch.bfloat16)
        kk.call("kk_glu_fwd", hh, g32, rows, Fd, seed, 5, p, 0)
        kk.call("kk_glu_fwd", hh.bfloat16(), g16, rows, Fd, seed, 5, p, 1)
        # bf16 storage evaluates GELU by kk_gelu_pair_fast (|error| < 5e-7 before the rounding to bf16), fp32 storage by the
        # exact erf: the rounded results differ in a last bit now and then, and the dropout masks are identical
        def same(a16, a32, what):
            ref = a32.bfloat16()
            assert float((a16 == ref).float().mean()) > 0.995, what
            assert torch.equal(a16 == 0, ref == 0) or p == 0.0, what + ": mask"
            assert float((a16.float() - a32).abs().max()) <= 2 ** -8 * float(a32.abs().max()) + 1e-6, what
        same(g16, g32, "glu fwd: bf16 storage = rounded fp32 result")
        d32, d16 = torch.empty(rows, 2 * Fd, device="cuda"), torch.empty(rows, 2 * Fd, device="cuda", dtype=torch.bfloat16)
        kk.call("kk_glu_bwd", dg, hh, d32, rows, Fd, seed, 5, p, 0)
        kk.call("kk_glu_bwd", dg.bfloat16(), hh.bfloat16(), d16, rows, Fd, seed, 5, p, 1)
        same(d16, d32, "glu bwd")
    # im2col3 (fp32 in, bf16 columns) and its transpose (bf16 columns in, fp32 out)
    B, L, C = 2, 600, 64
    x = dev(torch.randn(B * L, C, generator=g))
    c32, c16 = torch.empty(B * L, 3 * C, device="cuda"), torch.empty(B * L, 3 * C, device="cuda", dtype=torch.bfloat16)
    kk.call("kk_im2col3_fwd", x, c32, B, L, C, 512, 0)
    kk.call("kk_im2col3_fwd", x, c16, B, L, C, 512, 1)
    assert torch.equal(c16, c32.bfloat16())
    dcol = dev(_r16(torch.randn(B * L, 3 * C, generator=g)))
    a32, a16 = torch.empty(B * L, C, device="cuda"), torch.empty(B * L, C, device="cuda")
    kk.call("kk_im2col3_bwd", dcol, a32, B, L, C, 512, 0)
    kk.call("kk_im2col3_bwd", dcol.bfloat16(), a16, B, L, C, 512, 1)
    assert torch.equal(a16, a32)
    # rowdot with a bf16 x (stop head on the bf16 decoder output)
    xr, w, b = dev(_r16(torch.randn(B * L, C, generator=g))), dev(torch.randn(C, generator=g)), dev(torch.randn(1, generator=g))
    o32, o16 = torch.empty(B * L, device="cuda"), torch.empty(B * L, device="cuda")
    kk.call("kk_rowdot_fwd", xr, w, b, None, o32, B * L, C, L, 0, 0)
    kk.call("kk_rowdot_fwd", xr.bfloat16(), w, b, None, o16, B * L, C, L, 0, 1)
    close(o16, o32, 1e-5, 1e-5, "rowdot fwd bf16 x")
    dout = dev(torch.randn(B * L, generator=g))
    dw32, dw16, db32, db16 = (torch.zeros(n, device="cuda") for n in (C, C, 1, 1))
    kk.call("kk_rowdot_bwd", dout, xr, w, None, None, dw32, db32, B * L, C, L, 0, 0, None)
    kk.call("kk_rowdot_bwd", dout, xr.bfloat16(), w, None, None, dw16, db16, B * L, C, L, 0, 1, None)
    close(dw16, dw32, 1e-4, 1e-5, "rowdot dw bf16 x")
    close(db16, db32, 1e-4, 1e-5, "rowdot db bf16 x")
    # column sums of a bf16 matrix; fp32 -> bf16 cast
    X = dev(_r16(torch.randn(777, 200, generator=g)))
    s32, s16 = torch.zeros(200, device="cuda"), torch.zeros(200, device="cuda")
    kk.call("kk_colsum_acc", X, 200, 777, 200, s32, 0)
    kk.call("kk_colsum_acc", X.bfloat16(), 200, 777, 200, s16, 1)
    close(s16, s32, 1e-4, 1e-5, "colsum bf16")
    src = dev(torch.randn(4096 + 8, generator=g))
    dst = torch.empty(4096 + 8, device="cuda", dtype=torch.bfloat16)
    kk.call("kk_cast_f32_bf16", src, dst, src.numel())
    assert torch.equal(dst, src.bfloat16()), "cast must round to nearest even like torch"


def test_memory_path_bf16_storage(kk):
    """bucket_embed_add writes the cross-attention memory, SpecAugment masks it in place: both in bf16 storage."""
    g = torch.Generator().manual_seed(4)
    B, T, H, nb = 2, 40, 64, 16
    x, pitch, energy = torch.randn(B, T, H, generator=g), torch.rand(B, T, generator=g), torch.rand(B, T, generator=g)
    bins = torch.linspace(0, 1, nb - 1)
    pemb, eemb = torch.randn(nb, H, generator=g), torch.randn(nb, H, generator=g)
    lens = torch.tensor([40, 25])
    outs = []
    for flag in (0, 1):
        out = torch.empty(B, T, H, device="cuda", dtype=torch.bfloat16 if flag else torch.float32)
        pi, ei = torch.empty(B, T, dtype=torch.int32, device="cuda"), torch.empty(B, T, dtype=torch.int32, device="cuda")
        fm = torch.empty(B, T, dtype=torch.uint8, device="cuda")
        kk.call("kk_bucket_embed_add_fwd", dev(x), dev(pitch), dev(energy), dev(bins), dev(bins), dev(pemb), dev(eemb), dev(lens),
                out, pi, ei, fm, B, T, H, nb, flag)
        outs.append(out)
    assert torch.equal(outs[1], outs[0].bfloat16())
    seed = torch.tensor([3], dtype=torch.int32, device="cuda")
    a, b = outs[0].clone(), outs[1].clone()
    kk.call("kk_specaug", a, B, T, H, seed, 20, 5, 3, 1, 2, 0)
    kk.call("kk_specaug", b, B, T, H, seed, 20, 5, 3, 1, 2, 1)
    assert torch.equal(b, a.bfloat16()) and bool((b == 0).any())


@pytest.mark.parametrize("rows,H", [(333, 512), (4096, 128), (40, 1024)])
def test_norm_bwd_partials_and_reduce(kk, rows, H):
    """Column reductions through per-workgroup partial rows + one kk_partials_reduce launch = the atomics path."""
    g = torch.Generator().manual_seed(rows + H)
    x, dy = dev(torch.randn(rows, H, generator=g)), dev(torch.randn(rows, H, generator=g))
    gam, bet = dev(1 + 0.1 * torch.randn(H, generator=g)), dev(0.1 * torch.randn(H, generator=g))
    y, mean, rstd = torch.empty(rows, H, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    kk.call("kk_layernorm_fwd", x, gam, bet, y, mean, rstd, rows, H, 0)
    nb = kk.load().kk_norm_bwd_blocks(rows, H)
    ref, got = [], []
    for use_part in (0, 1):
        dx = torch.zeros(rows, H, device="cuda")
        dg, db, dgn = (torch.full((H,), 0.5, device="cuda") for _ in range(3))
        p_ln, p_rms = torch.full((nb, 2 * H), 9.0, device="cuda"), torch.full((nb, H), 9.0, device="cuda")
        kk.call("kk_layernorm_bwd", dy, x, gam, mean, rstd, dx, 0, dg, db, p_ln if use_part else None, rows, H, 0)
        dxr = torch.empty(rows, H, device="cuda")
        kk.call("kk_rmsnorm_bwd", dy, x, gam, rstd, dxr, dgn, p_rms if use_part else None, rows, H, 0)
        if use_part:
            assert float(dg[0]) == 0.5 and float(dgn[0]) == 0.5, "with partials the kernels must not touch the gradient vectors"
            table = kk.reduce_table([(p_ln, dg, db, nb, 2 * H, H), (p_rms, dgn, None, nb, H, H)], "cuda")
            kk.call("kk_partials_reduce", table, 2, 2 * H)
        (got if use_part else ref).extend([dx, dxr, dg, db, dgn])
    for a, b, n in zip(got, ref, ("ln dx", "rms dx", "dgamma", "dbeta", "dgain")):
        close(a, b, 2e-4, 1e-4, n)


@pytest.mark.parametrize("ffn,ln,bf", [(1, 1, 1), (0, 1, 1), (1, 0, 0), (0, 1, 0)])
def test_sublayer_tail_equals_separate_kernels(kk, ffn, ln, bf):
    """kk_sublayer_out_fwd == kk_rmsnorm_fwd -> kk_dropout_fwd (residual, two dropouts, DropPath) -> kk_layernorm_fwd with
    the same seed and call sites (identical masks)."""
    g = torch.Generator().manual_seed(3 + ffn + 2 * ln)
    rows, H, S = 203, 512, 29
    y = dev(torch.randn(rows, H, generator=g))
    if bf and ffn:
        y = y.bfloat16()
    res, gain = dev(torch.randn(rows, H, generator=g)), dev(1 + 0.1 * torch.randn(H, generator=g))
    gam, bet = dev(1 + 0.1 * torch.randn(H, generator=g)), dev(0.1 * torch.randn(H, generator=g))
    seed = torch.tensor([77], dtype=torch.int32, device="cuda")
    p1, p2, dpr = 0.2, (0.2 if ffn else 0.0), 0.1
    # separate kernels
    t = y
    rs_a = torch.zeros(rows, device="cuda")
    if ffn:
        t = torch.empty(rows, H, device="cuda")
        kk.call("kk_rmsnorm_fwd", y, gain, None, t, rs_a, rows, H, 1 if y.dtype == torch.bfloat16 else 0)
    xa = torch.empty(rows, H, device="cuda")
    kk.call("kk_dropout_fwd", t, res, 0, xa, rows, H, S, seed, 40, p1, 41, p2, 42, dpr)
    ndt = torch.bfloat16 if bf else torch.float32
    na, ma, ra = torch.empty(rows, H, device="cuda", dtype=ndt), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    if ln:
        kk.call("kk_layernorm_fwd", xa, gam, bet, na, ma, ra, rows, H, 1 if bf else 0)
    # fused
    xb = torch.empty(rows, H, device="cuda")
    nb, mb, rb, rs_b = torch.empty_like(na), torch.empty_like(ma), torch.empty_like(ra), torch.zeros(rows, device="cuda")
    kk.call("kk_sublayer_out_fwd", y, 1 if y.dtype == torch.bfloat16 else 0, gain if ffn else None, rs_b if ffn else None, res, xb,
            gam if ln else None, bet if ln else None, nb if ln else None, 1 if bf else 0, mb if ln else None, rb if ln else None,
            rows, H, S, seed, 40, p1, 41, p2, 42, dpr)
    close(xb, xa, 1e-6, 1e-6, "fused tail: residual stream")
    assert float((xb - res).abs().min()) == 0.0 and 0.1 < float(((xb - res) == 0).float().mean()) < 0.7   # masks really applied
    if ffn:
        close(rs_b, rs_a, 1e-6, 1e-6, "fused tail: rms 1/rms")
    if ln:
        close(mb, ma, 1e-6, 1e-6, "fused tail: LN mean")
        close(rb, ra, 1e-5, 1e-5, "fused tail: LN rstd")
        close(nb, na, 2e-2 if bf else 1e-5, 1e-2 if bf else 1e-5, "fused tail: LN output")


@pytest.mark.parametrize("rows,K,ffn,ln,nbf", [(4096, 512, 0, 1, 1), (203, 512, 0, 1, 1), (8192, 512, 0, 1, 1), (1333, 1536, 1, 1, 1),
                                               (4096, 2048, 1, 0, 0), (77, 64, 0, 1, 0), (2100, 576, 1, 1, 1), (4096, 1280, 0, 0, 1)])
def test_linear_tail_fwd_matches_two_launches(kk, rows, K, ffn, ln, nbf):
    """kk_linear_tail_fwd (row-owner Linear + sub-layer tail in one launch) against (a) the two launches it replaces — kk_gemm with a
    bf16 C, then kk_sublayer_out_fwd: BIT-identical (same MFMA shape and k order, same tail arithmetic, same masks) — and (b) a
    float64 evaluation of y = x.W^T + b from the same bf16 operands.  K = 512: resident x panel, no barrier in the k-loop; 1536 /
    2048 / 1280: the chunked panel with its per-256-k meetings; 576 / 64: a short last chunk / a single k-tile; ragged rows."""
    g = torch.Generator().manual_seed(rows + K)
    H, S = 512, 29
    x = dev(torch.randn(rows, K, generator=g)).bfloat16()
    W = dev(torch.randn(H, K, generator=g) / K ** 0.5).bfloat16()
    bias = dev(0.1 * torch.randn(H, generator=g))
    res, gain = dev(torch.randn(rows, H, generator=g)), dev(1 + 0.1 * torch.randn(H, generator=g))
    gam, bet = dev(1 + 0.1 * torch.randn(H, generator=g)), dev(0.1 * torch.randn(H, generator=g))
    seed = torch.tensor([91], dtype=torch.int32, device="cuda")
    p1, p2, dpr = 0.2, (0.2 if ffn else 0.0), 0.1
    ndt = torch.bfloat16 if nbf else torch.float32
    assert kk.load().kk_linear_tail_supported(rows, H, K) == 1
    # the two launches
    ya = torch.empty(rows, H, device="cuda", dtype=torch.bfloat16)
    kk.call("kk_gemm", 0, 0, rows, H, K, 1.0, x, K, W, K, 0.0, ya, H, bias, None, 0, 0, 0, kk.KK_MATH_BF16, 1 | 2 | 4)
    xa, rs_a = torch.empty(rows, H, device="cuda"), torch.zeros(rows, device="cuda")
    na, ma, ra = torch.zeros(rows, H, device="cuda", dtype=ndt), torch.zeros(rows, device="cuda"), torch.zeros(rows, device="cuda")
    kk.call("kk_sublayer_out_fwd", ya, 1, gain if ffn else None, rs_a if ffn else None, res, xa, gam if ln else None, bet if ln else None,
            na if ln else None, nbf, ma if ln else None, ra if ln else None, rows, H, S, seed, 40, p1, 41, p2, 42, dpr)
    # one launch
    yb = torch.zeros(rows, H, device="cuda", dtype=torch.bfloat16)
    xb, rs_b = torch.empty(rows, H, device="cuda"), torch.zeros(rows, device="cuda")
    nb, mb, rb = torch.zeros_like(na), torch.zeros_like(ma), torch.zeros_like(ra)
    kk.call("kk_linear_tail_fwd", x, K, W, bias, K, yb if ffn else None, 1, gain if ffn else None, rs_b if ffn else None, res, xb,
            gam if ln else None, bet if ln else None, nb if ln else None, nbf, mb if ln else None, rb if ln else None,
            rows, H, S, seed, 40, p1, 41, p2, 42, dpr)
    torch.cuda.synchronize()
    assert kk.last_kernel() == "linear_tail_fwd"
    assert torch.equal(xb, xa), f"residual stream differs: {float((xb - xa).abs().max())}"
    if ffn:
        assert torch.equal(yb, ya) and torch.equal(rs_b, rs_a)
    if ln:
        assert torch.equal(mb, ma), f"LN mean differs: {float((mb - ma).abs().max())}"
        assert torch.equal(rb, ra), f"LN rstd differs: {float((rb - ra).abs().max())}"
        assert torch.equal(nb, na), f"LN output differs: {float((nb.float() - na.float()).abs().max())}"
    # float64 anchor of the projection under the masks: where nothing was dropped, x_out - res = y * scale
    y64 = x.double() @ W.double().t() + bias.double()
    if ffn:
        y64 = y64 / torch.sqrt((y64 ** 2).mean(1, keepdim=True) + torch.finfo(torch.float32).eps) * gain.double()
    d = (xb - res).double()
    kept = d != 0
    assert 0.3 < float(kept.float().mean()) < 0.9
    ratio = (d / y64)[kept & (y64.abs() > 0.05)]
    scales = torch.tensor([1 / (0.8 * 0.9), 1 / (0.8 * 0.8 * 0.9)] if ffn else [1 / (0.8 * 0.9)], device="cuda", dtype=torch.float64)
    err = (ratio[:, None] / scales[None, :] - 1).abs().min(1).values
    assert float(err.max()) < 2e-2, f"projection under the masks: {float(err.max())}"        # bf16 rounding of y: 2^-8


@pytest.mark.parametrize("ffn,bf,acc", [(1, 1, 1), (0, 1, 1), (1, 0, 0), (0, 0, 1)])
def test_sublayer_tail_backward_equals_separate_kernels(kk, ffn, bf, acc):
    """kk_sublayer_in_bwd == kk_layernorm_bwd -> kk_dropout_bwd -> (kk_rmsnorm_bwd) -> kk_colsum_acc, same masks."""
    g = torch.Generator().manual_seed(11 + ffn + 2 * bf)
    rows, H, S = 221, 512, 17
    dt = torch.bfloat16 if bf else torch.float32
    x = dev(torch.randn(rows, H, generator=g))
    dn = dev(torch.randn(rows, H, generator=g)).to(dt)
    y = dev(torch.randn(rows, H, generator=g)).to(dt)
    gam, bet, gain = (dev(1 + 0.1 * torch.randn(H, generator=g)) for _ in range(3))
    dres0 = dev(torch.randn(rows, H, generator=g))
    seed = torch.tensor([5], dtype=torch.int32, device="cuda")
    p1, p2, dpr = 0.2, (0.2 if ffn else 0.0), 0.1
    n_, mean, rstd = torch.empty(rows, H, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    kk.call("kk_layernorm_fwd", x, gam, bet, n_, mean, rstd, rows, H, 0)
    rstd_f = torch.empty(rows, device="cuda")
    kk.call("kk_rmsnorm_fwd", y, gain, None, torch.empty(rows, H, device="cuda"), rstd_f, rows, H, bf)
    # separate kernels (atomics paths)
    dres_a = dres0.clone() if acc else torch.zeros(rows, H, device="cuda")
    dgam_a, dbet_a, dgain_a, dbias_a = (torch.zeros(H, device="cuda") for _ in range(4))
    kk.call("kk_layernorm_bwd", dn, x, gam, mean, rstd, dres_a, acc, dgam_a, dbet_a, None, rows, H, bf)
    masked = torch.empty(rows, H, device="cuda", dtype=torch.float32 if ffn else dt)
    kk.call("kk_dropout_bwd", dres_a, masked, rows, H, S, seed, 40, p1, 41, p2, 42, dpr, 0 if ffn else bf)
    if ffn:
        dy_a = torch.empty(rows, H, device="cuda", dtype=dt)
        kk.call("kk_rmsnorm_bwd", masked, y, gain, rstd_f, dy_a, dgain_a, None, rows, H, bf)
    else:
        dy_a = masked
    kk.call("kk_colsum_acc", dy_a, H, rows, H, dbias_a, bf)
    # fused + one reduce
    dres_b = dres0.clone()
    dy_b = torch.empty(rows, H, device="cuda", dtype=dt)
    nb = kk.load().kk_sublayer_in_bwd_blocks(rows)
    part = torch.full((nb, 4 * H), 3.0, device="cuda")
    kk.call("kk_sublayer_in_bwd", dn, bf, x, gam, mean, rstd, dres_b, acc, y if ffn else None, gain if ffn else None,
            rstd_f if ffn else None, dy_b, bf, part, rows, H, S, seed, 40, p1, 41, p2, 42, dpr)
    dgam_b, dbet_b, dgain_b, dbias_b = (torch.zeros(H, device="cuda") for _ in range(4))
    table = kk.reduce_table([(part, dgam_b, dbet_b, nb, 2 * H, H, 4 * H),
                             (part[:, 2 * H:], dbias_b, dgain_b if ffn else None, nb, 2 * H if ffn else H, H, 4 * H)], "cuda")
    kk.call("kk_partials_reduce", table, 2, 2 * H)
    close(dres_b, dres_a, 2e-5, 1e-5, "fused bwd tail: stream gradient")
    close(dy_b, dy_a, 3e-2 if bf else 2e-5, 1e-2 if bf else 1e-5, "fused bwd tail: dy")
    close(dgam_b, dgam_a, 2e-3, 1e-4, "fused bwd tail: dgamma")
    close(dbet_b, dbet_a, 2e-3, 1e-4, "fused bwd tail: dbeta")
    # (the separate path sums the bf16-ROUNDED dy, the fused kernel sums before rounding: ~sqrt(rows) * 2^-9 * |dy| apart)
    close(dbias_b, dbias_a, 0.5 if bf else 2e-3, 1e-2 if bf else 1e-4, "fused bwd tail: bias column sums")
    if ffn:
        close(dgain_b, dgain_a, 2e-3, 1e-4, "fused bwd tail: RMSNorm gain")


def test_attention_dq_computes_delta_in_kernel(kk):
    """kk_attn_bwd_dq with O given: Delta written by the kernel == kk_attn_delta, and dQ identical to the two-launch form."""
    g = torch.Generator().manual_seed(2)
    B, h, S = 2, 4, 200
    H = h * 64
    q, k, v, do = (dev(torch.randn(B * S, H, generator=g)).bfloat16() for _ in range(4))
    o, lse = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, h, S, device="cuda")
    kk.call("kk_attn_fwd", q, k, v, o, lse, B, h, S, S, H, H, H, H, None, 1, 0.125, None, 0, 0.0, 1, 1)
    d_ref, d_new = torch.empty(B, h, S, device="cuda"), torch.full((B, h, S), 9.0, device="cuda")
    kk.call("kk_attn_delta", o, do, d_ref, B, h, S, H, H, 1)
    dq_ref, dq_new = torch.empty_like(q), torch.empty_like(q)
    kk.call("kk_attn_bwd_dq", q, k, v, do, lse, d_ref, dq_ref, B, h, S, S, H, H, H, H, H, None, 1, 0.125, None, 0, 0.0, 1, 1, None, 0, None)
    kk.call("kk_attn_bwd_dq", q, k, v, do, lse, d_new, dq_new, B, h, S, S, H, H, H, H, H, None, 1, 0.125, None, 0, 0.0, 1, 1, o, H, None)
    close(d_new, d_ref, 1e-5, 1e-5, "delta computed in the dQ kernel")
    close(dq_new, dq_ref, 1e-3, 1e-3, "dQ with in-kernel delta")


@pytest.mark.parametrize("B,h,S", [(1, 2, 48), (1, 1, 200), (2, 4, 256)])
def test_weight_warming_on_launches_of_a_few_workgroups(kk, B, h, S):
    """kk_attn_warm_next with ONE matrix (the second slot empty) on launches of fewer than 8 workgroups: the forward and the dQ
    half terminate and their outputs are those of the launches without warming (the warming loads are never read)."""
    g = torch.Generator().manual_seed(S)
    H = h * 64
    q, k, v, do = (dev(torch.randn(B * S, H, generator=g)).bfloat16() for _ in range(4))
    W = dev(torch.randn(H, H, generator=g)).bfloat16()
    lib = kk.load()
    out = []
    for warm in (False, True, True):
        o, lse = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, h, S, device="cuda")
        if warm:
            lib.kk_attn_warm_next(W.data_ptr(), W.numel() * 2, None, 0)
        kk.call("kk_attn_fwd", q, k, v, o, lse, B, h, S, S, H, H, H, H, None, 1, 0.125, None, 0, 0.0, 1, 1)
        dl = torch.empty(B, h, S, device="cuda")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        if warm:
            lib.kk_attn_warm_next(None, 0, W.data_ptr(), W.numel() * 2)     # (a lone second slot moves to the first)
        kk.call("kk_attn_delta", o, do, dl, B, h, S, H, H, 1)
        kk.call("kk_attn_bwd", q, k, v, do, lse, dl, dq, dk, dv, B, h, S, S, H, H, H, H, H, H, H, None, 1, 0.125, None, 0, 0.0, 1, 1, None, None)
        torch.cuda.synchronize()
        out.append((o, lse, dq, dk, dv))
    for t0, t1, t2 in zip(*out):
        assert torch.equal(t0, t1) and torch.equal(t0, t2)


@pytest.mark.parametrize("T,F,H,p", [(200, 96, 128, 0.0), (1000, 1536, 512, 0.2), (77, 192, 64, 0.1), (4096, 1536, 512, 0.2), (4000, 1000, 256, 0.1)])
def test_gemm_dgrad_glu_epilogue(kk, T, F, H, p):
    """kk_gemm_dgrad_glu == kk_gemm (dgrad) -> kk_glu_bwd -> kk_colsum_acc, up to the bf16 rounding of the intermediate dG
    that the fused form never makes."""
    g = torch.Generator().manual_seed(T + F)
    bf = torch.bfloat16
    dy = dev(torch.randn(T, H, generator=g)).to(bf)
    W = dev(torch.randn(H, F, generator=g) * 0.1).to(bf)
    h1 = dev(torch.randn(T, 2 * F, generator=g)).to(bf)
    seed = torch.tensor([21], dtype=torch.int32, device="cuda")
    dg = torch.empty(T, F, device="cuda", dtype=bf)
    kk.call("kk_gemm", 0, 1, T, F, H, 1.0, dy, H, W, F, 0.0, dg, F, None, None, 0, 0, 1, 1, 7)
    dh_a = torch.empty(T, 2 * F, device="cuda", dtype=bf)
    kk.call("kk_glu_bwd", dg, h1, dh_a, T, F, seed, 9, p, 1)
    bias_a = torch.zeros(2 * F, device="cuda")
    kk.call("kk_colsum_acc", dh_a, 2 * F, T, 2 * F, bias_a, 1)
    nb = kk.load().kk_gemm_dgrad_glu_blocks(T)
    part = torch.full((nb, 2 * F), 4.0, device="cuda")
    dh_b = torch.full((T, 2 * F), 7.0, device="cuda", dtype=bf)
    kk.call("kk_gemm_dgrad_glu", T, F, H, dy, H, W, h1, dh_b, part, seed, 9, p)
    bias_b = torch.zeros(2 * F, device="cuda")
    kk.call("kk_partials_reduce", kk.reduce_table([(part, bias_b, None, nb, 2 * F, 2 * F)], "cuda"), 1, 2 * F)
    close(dh_b, dh_a, 0.05 * math.sqrt(H / 64), 3e-2, "fused GLU backward")
    zero = (dh_a.float() == 0)
    if p > 0:
        assert 0.5 * p < float(zero.float().mean()) < 1.5 * p + 0.01
        assert bool(((dh_b.float() == 0) == zero).float().mean() > 0.999), "same dropout mask"
    close(bias_b, bias_a, 0.05 * math.sqrt(T) * math.sqrt(H / 64), 3e-2, "linear1 bias gradient from the epilogue partials")


@pytest.mark.gpu
@pytest.mark.parametrize("T,F,H,p", [(64, 64, 64, 0.0), (333, 192, 128, 0.1), (4096, 2048, 512, 0.1), (70, 40, 64, 0.2), (4096, 1536, 512, 0.2), (8000, 1000, 256, 0.1)])
def test_gemm_linear_glu_epilogue(kk, T, F, H, p):
    """kk_gemm_linear_glu == kk_gemm (+bias, bf16 out) -> kk_glu_fwd: same h1 bits, same gate, same dropout mask."""
    g = torch.Generator().manual_seed(T + F)
    bf = torch.bfloat16
    x = dev(torch.randn(T, H, generator=g)).to(bf)
    W = dev(torch.randn(2 * F, H, generator=g) * 0.1).to(bf)
    b = dev(torch.randn(2 * F, generator=g))
    seed = torch.tensor([33], dtype=torch.int32, device="cuda")
    h_a = torch.empty(T, 2 * F, device="cuda", dtype=bf)
    kk.call("kk_gemm", 0, 0, T, 2 * F, H, 1.0, x, H, W, H, 0.0, h_a, 2 * F, b, None, 0, 0, 1, 1, 7)
    g_a = torch.empty(T, F, device="cuda", dtype=bf)
    kk.call("kk_glu_fwd", h_a, g_a, T, F, seed, 13, p, 1)
    h_b = torch.full((T, 2 * F), 7.0, device="cuda", dtype=bf)
    g_b = torch.full((T, F), 7.0, device="cuda", dtype=bf)
    kk.call("kk_gemm_linear_glu", T, F, H, x, H, W, b, h_b, g_b, F, seed, 13, p)
    assert torch.equal(h_b, h_a), "h1 from the dual-panel tile is the plain GEMM's, bit for bit"
    assert torch.equal(g_b, g_a), "gate + mask"
    ref = torch.nn.functional.gelu(h_a[:, :F].float()) * h_a[:, F:].float()
    keep = g_a.float() != 0
    close(g_a.float()[keep], (ref / (1 - p))[keep], 2e-2, 2e-2, "gate value")


@pytest.mark.gpu
@pytest.mark.parametrize("T,shapes,split", [(256, [(64, 64)], 0), (1000, [(512, 192), (72, 64), (192, 512), (64, 136)], 0),
                                            (4096, [(512, 2048), (4096, 512), (512, 512), (1536, 512)], 0),
                                            (4096, [(512, 512), (1536, 512)], 2), (520, [(128, 64)] * 8, 3),
                                            (4100, [(1536, 512), (512, 512), (512, 512), (512, 512), (3072, 512), (512, 1536)], 0),
                                            (2000, [(6144, 512)], 0), (1500, [(1000, 520), (200, 1536), (3072, 136)], 0)])
def test_gemm_wgrad_group(kk, T, shapes, split):
    """kk_gemm_wgrad_group == one kk_gemm(ta=1, tb=1, beta=1) per problem: bit for bit without k-slices (same tile code,
    same order of accumulation), to fp32 atomics' reordering with them."""
    g = torch.Generator().manual_seed(T)
    bf = torch.bfloat16
    probs = []
    for M, N in shapes:
        wide = dev(torch.randn(T, M + 8, generator=g)).to(bf)          # a strided view, like a slice of the fused q|k|v gradient
        dy, x = wide[:, 8:], dev(torch.randn(T, N, generator=g)).to(bf)
        probs.append((dy, x, dev(torch.randn(M, N, generator=g))))
    ref = []
    for dy, x, dw in probs:
        r = dw.clone()
        kk.call("kk_gemm", 1, 1, dy.shape[1], x.shape[1], T, 1.0, dy, dy.stride(0), x, x.stride(0), 1.0, r, x.shape[1], None, None,
                0, 0, 1, 1, 3)
        ref.append(r)
    kk.call("kk_gemm_wgrad_group", kk.wgrad_table(probs), len(probs), split, 0, None, None, None)
    torch.cuda.synchronize()
    for (dy, x, dw), r in zip(probs, ref):
        if split == 0 and sum(-(-m // 128) * -(-n // 64) for m, n in shapes) * 2 > 384:      # (no k-slices: the grouped launch's 128x64 tiles fill the chip)
            assert torch.equal(dw, r)
        else:
            close(dw, r, 2e-3 * math.sqrt(T / 256), 1e-4, "grouped weight gradient (k-sliced)")
    with pytest.raises(RuntimeError):
        kk.call("kk_gemm_wgrad_group", kk.wgrad_table(probs * 9), len(probs) * 9, 0, 0, None, None, None)
    # overwrite (the first micro-batch of an accumulation cycle): dw = product whatever dw held, never k-sliced
    fresh = []
    for dy, x, dw in probs:
        r = torch.zeros_like(dw)
        kk.call("kk_gemm", 1, 1, dy.shape[1], x.shape[1], T, 1.0, dy, dy.stride(0), x, x.stride(0), 1.0, r, x.shape[1], None, None,
                0, 0, 1, 1, 3)
        fresh.append(r)
        dw.fill_(float("nan"))
    kk.call("kk_gemm_wgrad_group", kk.wgrad_table(probs), len(probs), 0, 1, None, None, None)
    for (dy, x, dw), r in zip(probs, fresh):
        close(dw, r, 2e-3 * math.sqrt(T / 256), 1e-4, "grouped weight gradient, overwrite")
    with pytest.raises(RuntimeError):
        kk.call("kk_gemm_wgrad_group", kk.wgrad_table(probs), len(probs), 2, 1, None, None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,h,Sk,time_major", [(2, 8, 600, False), (1, 16, 37, True), (3, 8, 1, False), (1, 8, 1500, True)])
def test_attention_decode_step(kk, bf16, B, h, Sk, time_major):
    """kk_attn_fwd at Sq = 1 (the incremental path of transformers.py:237-253: one decoder step against the KV cache / the memory) takes
    the decode kernel — one (batch, head) per workgroup — and equals a float64 softmax(q.K^T / 8).V under a key mask, for both layouts
    generate() uses: batch-major K|V rows (cross-attention) and the time-major cache [t][B*H] seen as ONE batch of B*h heads."""
    g = torch.Generator().manual_seed(Sk + B)
    H = h * 64
    dt = torch.bfloat16 if bf16 else torch.float32
    q = dev(torch.randn(B, H, generator=g)).to(dt)
    K, V = dev(torch.randn(B, Sk, H, generator=g)).to(dt), dev(torch.randn(B, Sk, H, generator=g)).to(dt)
    mask = torch.rand(B, Sk, generator=g) < 0.3
    mask[:, 0] = False
    if B > 1:
        mask[B - 1] = True                                     # a fully masked batch item: zero output, lse = inf
    if time_major:                                             # ONE batch, B*h heads, rows = time steps; the mask is per time step
        assert B == 1
    out, lse = torch.empty(B, H, device="cuda", dtype=dt), torch.empty(B, h, 1, device="cuda")
    kk.call("kk_attn_fwd", q, K, V, out, lse, B, h, 1, Sk, H, H, H, H, dev(mask.to(torch.uint8)), 0, 0.125, _seed(), 0, 0.0,
            1 if bf16 else 0, 1 if bf16 else 0)
    torch.cuda.synchronize()
    qd, Kd, Vd = q.double().cpu().view(B, h, 64), K.double().cpu().view(B, Sk, h, 64), V.double().cpu().view(B, Sk, h, 64)
    s = torch.einsum("bhd,bkhd->bhk", qd, Kd) * 0.125
    s = s.masked_fill(mask[:, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1).nan_to_num(0.0)
    ref = torch.einsum("bhk,bkhd->bhd", p, Vd).reshape(B, H)
    close(out.float(), ref.float(), 2e-2 if bf16 else 2e-5, 1e-2 if bf16 else 1e-5, "decode attention")
    ref_lse = torch.logsumexp(s, dim=-1)
    live = ~mask.all(dim=1)
    close(lse.view(B, h)[live.cuda()], ref_lse[live].float(), 1e-4, 1e-5, "decode attention log-sum-exp")
    if B > 1:
        assert bool((out[B - 1] == 0).all()) and bool(torch.isinf(lse[B - 1]).all())


@pytest.mark.gpu
def test_copy_many(kk):
    g = torch.Generator().manual_seed(5)
    srcs = [dev(torch.randn(n, generator=g)) for n in (1, 7, 4096, 5000, 40000)] + \
           [dev(torch.randint(0, 100, (n,), generator=g)) for n in (8, 513)] + [dev(torch.randint(0, 2, (33,), generator=g)).to(torch.uint8)]
    srcs.append(dev(torch.randn(1001, generator=g))[1:])                      # 4-byte aligned only: the byte path
    dsts = [torch.full_like(s, 3) for s in srcs]
    kk.copy_many(list(zip(dsts, srcs)))
    torch.cuda.synchronize()
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)
    with pytest.raises(RuntimeError):
        kk.call("kk_copy_many", None, None, None, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("T,parts,heads,K,S,rope", [(64, 1, 1, 64, 64, 0), (333, 3, 2, 128, 111, 3), (4096, 3, 8, 512, 512, 3),
                                                    (1000, 12, 8, 512, 125, 0), (70, 2, 3, 192, 35, 1), (8192, 3, 8, 512, 1024, 3),
                                                    (4096, 12, 8, 512, 512, 0), (8000, 1, 8, 256, 1000, 1), (5000, 3, 5, 192, 625, 3)])
def test_gemm_qkv_headnorm_epilogue(kk, T, parts, heads, K, S, rope):
    """kk_gemm_qkv_headnorm == kk_gemm (bf16 out) -> kk_headnorm_rope_fwd per group of three parts: same raw bits, same
    normalised bits; and the normalised rows have unit RMS before the gain."""
    g = torch.Generator().manual_seed(T + parts)
    bf = torch.bfloat16
    H = heads * 64
    N = parts * H
    x = dev(torch.randn(T, K, generator=g)).to(bf)
    W = dev(torch.randn(N, K, generator=g) * 0.1).to(bf)
    gains = [dev(1.0 + 0.2 * torch.randn(64, generator=g)) for _ in range(parts)]
    c, s = (dev(t) for t in O.rope_tables(S, 64))
    raw_a = torch.empty(T, N, device="cuda", dtype=bf)
    kk.call("kk_gemm", 0, 0, T, N, K, 1.0, x, K, W, K, 0.0, raw_a, N, None, None, 0, 0, 1, 1, 7)
    y_a = torch.empty_like(raw_a)
    for p0 in range(0, parts, 3):
        gg = gains[p0:p0 + 3] + [None] * 3
        kk.call("kk_headnorm_rope_fwd", raw_a[:, p0 * H:], N, y_a[:, p0 * H:], N, T, heads, S, min(3, parts - p0), gg[0], gg[1], gg[2],
                (rope >> p0) & 7, c, s, 1)
    raw_b, y_b = torch.full_like(raw_a, 7.0), torch.full_like(raw_a, 7.0)
    kk.call("kk_gemm_qkv_headnorm", T, parts, heads, K, x, K, W, None, raw_b, N, y_b, N, S, kk.pointer_table(gains), rope, c, s)
    torch.cuda.synchronize()
    assert torch.equal(raw_b, raw_a), "projection saved for the backward"
    assert torch.equal(y_b, y_a), "normalised (+ rotated) heads"
    v = y_b[:, (parts - 1) * H:(parts - 1) * H + 64].float() / gains[parts - 1]
    if not (rope >> (parts - 1)) & 1:
        close(v.pow(2).mean(-1), torch.ones(T, device="cuda"), 2e-2, 0, "unit RMS per head")


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,Sq,Sk,causal,rope,bf16,p", [(2, 4, 200, 200, 1, 1, 1, 0.0), (2, 2, 64, 64, 0, 1, 1, 0.1), (1, 8, 512, 512, 1, 1, 1, 0.1),
                                                         (1, 2, 900, 1000, 0, 1, 1, 0.1), (1, 2, 1000, 1000, 1, 1, 1, 0.2),
                                                         (2, 2, 130, 70, 0, 0, 1, 0.0), (2, 2, 100, 100, 1, 1, 0, 0.0)])
def test_attention_backward_headnorm_epilogue(kk, B, h, Sq, Sk, causal, rope, bf16, p):
    """kk_attn_bwd_dq / kk_attn_bwd_dkv with the head-norm descriptors == the plain kernels followed by
    kk_headnorm_rope_bwd: same gradient of the raw projections (to the storage type's rounding: the row reductions are
    summed in a different order), same gain gradients after kk_partials_reduce."""
    g = torch.Generator().manual_seed(B * Sq + Sk)
    H = h * 64
    dt = torch.bfloat16 if bf16 else torch.float32
    io = 1 if bf16 else 0
    math_ = 1
    raw_q = dev(torch.randn(B * Sq, H, generator=g)).to(dt)
    raw_kv = dev(torch.randn(B * Sk, 2 * H, generator=g)).to(dt)
    gains = [dev(1.0 + 0.2 * torch.randn(64, generator=g)) for _ in range(3)]
    S = max(Sq, Sk)
    c, s = ((dev(t) for t in O.rope_tables(S, 64)) if rope else (None, None))
    q_n, kv_n = torch.empty_like(raw_q), torch.empty_like(raw_kv)
    kk.call("kk_headnorm_rope_fwd", raw_q, H, q_n, H, B * Sq, h, Sq, 1, gains[0], None, None, 1 if rope else 0, c, s, io)
    kk.call("kk_headnorm_rope_fwd", raw_kv, 2 * H, kv_n, 2 * H, B * Sk, h, Sk, 2, gains[1], gains[2], None, 1 if rope else 0, c, s, io)
    k_n, v_n = kv_n, kv_n[:, H:]
    do = dev(torch.randn(B * Sq, H, generator=g)).to(dt)
    o, lse = torch.empty(B * Sq, H, device="cuda", dtype=dt), torch.empty(B, h, Sq, device="cuda")
    seed = torch.tensor([77], dtype=torch.int32, device="cuda")
    kk.call("kk_attn_fwd", q_n, k_n, v_n, o, lse, B, h, Sq, Sk, H, 2 * H, 2 * H, H, None, causal, 0.125, seed, 5, p, math_, io)
    delta = torch.empty(B, h, Sq, device="cuda")
    # reference: plain kernels, then the norm's backward
    dq_n, dkv_n = torch.empty_like(raw_q), torch.empty_like(raw_kv)
    kk.call("kk_attn_bwd_dq", q_n, k_n, v_n, do, lse, delta, dq_n, B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, None, causal, 0.125, seed, 5, p,
            math_, io, o, H, None)
    kk.call("kk_attn_bwd_dkv", q_n, k_n, v_n, do, lse, delta, dkv_n, dkv_n[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, 2 * H, 2 * H, None,
            causal, 0.125, seed, 5, p, math_, io, None)
    dq_a, dkv_a = torch.empty_like(raw_q), torch.empty_like(raw_kv)
    dg_a = [torch.zeros(64, device="cuda") for _ in range(3)]
    kk.call("kk_headnorm_rope_bwd", dq_n, H, raw_q, H, dq_a, H, B * Sq, h, Sq, 1, gains[0], None, None, dg_a[0], None, None, None,
            1 if rope else 0, c, s, io)
    kk.call("kk_headnorm_rope_bwd", dkv_n, 2 * H, raw_kv, 2 * H, dkv_a, 2 * H, B * Sk, h, Sk, 2, gains[1], gains[2], None, dg_a[1], dg_a[2],
            None, None, 1 if rope else 0, c, s, io)
    # fused
    nbq, nbk = kk.load().kk_attn_bwd_blocks(B, h, Sq), kk.load().kk_attn_bwd_blocks(B, h, Sk)
    pq, pkv = torch.full((1, nbq, 64), 5.0, device="cuda"), torch.full((2, nbk, 64), 5.0, device="cuda")
    dq_b, dkv_b = torch.full_like(raw_q, 7.0), torch.full_like(raw_kv, 7.0)
    delta_b = torch.empty_like(delta)
    kk.call("kk_attn_bwd_dq", q_n, k_n, v_n, do, lse, delta_b, dq_b, B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, None, causal, 0.125, seed, 5, p,
            math_, io, o, H, kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)]))
    kk.call("kk_attn_bwd_dkv", q_n, k_n, v_n, do, lse, delta_b, dkv_b, dkv_b[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, 2 * H, 2 * H, None,
            causal, 0.125, seed, 5, p, math_, io,
            kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)]))
    dg_b = [torch.zeros(64, device="cuda") for _ in range(3)]
    kk.call("kk_partials_reduce", kk.reduce_table([(pq[0], dg_b[0], None, nbq, 64, 64), (pkv[0], dg_b[1], None, nbk, 64, 64),
                                                   (pkv[1], dg_b[2], None, nbk, 64, 64)], "cuda"), 3, 64)
    torch.cuda.synchronize()
    tol = (2e-2, 2e-2) if bf16 else (2e-5, 2e-5)
    close(dq_b, dq_a, *tol, "d raw q")
    bad = ((dkv_b.float() - dkv_a.float()).abs() > tol[0] + tol[1] * dkv_a.float().abs()).nonzero().tolist()
    where = [(i, j, float(dkv_b[i, j]), float(dkv_a[i, j]), float(dkv_n[i, j])) for i, j in bad[:8]]      # (a one-off failure was seen once: say where)
    close(dkv_b, dkv_a, *tol, f"d raw k|v (row, column, fused, unfused, d normalised: {where})")
    if bf16:
        assert float((dq_b == dq_a).float().mean()) > 0.99 and float((dkv_b == dkv_a).float().mean()) > 0.99, "almost all bits equal"
    for j, name in enumerate(("q", "k", "v")):
        close(dg_b[j], dg_a[j], 2e-3 * math.sqrt(B * max(Sq, Sk)), 2e-3, f"gain gradient {name}")


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,Sq,Sk,causal,rope,p,masked", [(8, 8, 512, 512, 1, 1, 0.2, 0), (8, 8, 512, 512, 0, 0, 0.2, 1), (2, 8, 1024, 1024, 1, 1, 0.2, 0),
                                                            (1, 2, 900, 1000, 0, 1, 0.1, 1), (2, 2, 300, 384, 0, 0, 0.0, 0), (2, 4, 200, 200, 1, 1, 0.0, 0),
                                                            (1, 2, 130, 400, 0, 0, 0.1, 0), (1, 2, 1000, 1000, 1, 1, 0.2, 1), (3, 4, 41, 41, 0, 1, 0.15, 1)])
def test_attention_backward_two_pass(kk, B, h, Sq, Sk, causal, rope, p, masked):
    """kk_attn_bwd_ws (the dK/dV kernel also stores dS; dQ = dS . K as a pass without softmax work) against kk_attn_bwd given the same
    Delta: dK, dV and the k / v gain partials bit for bit (the same kernel body), dQ and the q gain partials within the rounding of
    a different summation order over the key units (same bf16 dS, same K: fp32 sums of the same products) — and against the
    workspace being poisoned beforehand (every tile the pass reads must have been written by this call).  The last case (one key
    tile) takes kk_attn_bwd inside the entry point."""
    g = torch.Generator().manual_seed(B * Sq + Sk + causal + 17)
    H = h * 64
    dt = torch.bfloat16
    raw_q = dev(torch.randn(B * Sq, H, generator=g)).to(dt)
    raw_kv = dev(torch.randn(B * Sk, 2 * H, generator=g)).to(dt)
    gains = [dev(1.0 + 0.2 * torch.randn(64, generator=g)) for _ in range(3)]
    c, s = ((dev(t) for t in O.rope_tables(max(Sq, Sk), 64)) if rope else (None, None))
    q_n, kv_n = torch.empty_like(raw_q), torch.empty_like(raw_kv)
    kk.call("kk_headnorm_rope_fwd", raw_q, H, q_n, H, B * Sq, h, Sq, 1, gains[0], None, None, 1 if rope else 0, c, s, 1)
    kk.call("kk_headnorm_rope_fwd", raw_kv, 2 * H, kv_n, 2 * H, B * Sk, h, Sk, 2, gains[1], gains[2], None, 1 if rope else 0, c, s, 1)
    k_n, v_n = kv_n, kv_n[:, H:]
    km = None
    if masked:
        km = torch.zeros(B, Sk, dtype=torch.uint8)
        km[:, Sk - 37:] = 1
        km[0, 5] = 1
        km = dev(km)
    do = dev(torch.randn(B * Sq, H, generator=g)).to(dt)
    o, lse = torch.empty(B * Sq, H, device="cuda", dtype=dt), torch.empty(B, h, Sq, device="cuda")
    seed = torch.tensor([91], dtype=torch.int32, device="cuda")
    kk.call("kk_attn_fwd", q_n, k_n, v_n, o, lse, B, h, Sq, Sk, H, 2 * H, 2 * H, H, km, causal, 0.125, seed, 5, p, 1, 1)
    delta = torch.empty(B, h, Sq, device="cuda")
    kk.call("kk_attn_delta", o, do, delta, B, h, Sq, H, H, 1)
    nbq, nbk = kk.load().kk_attn_bwd_blocks(B, h, Sq), kk.load().kk_attn_bwd_blocks(B, h, Sk)
    need = kk.load().kk_attn_bwd_ws_bytes(B, h, Sq, Sk)
    assert need == B * h * ((Sk + 31) // 32) * ((Sq + 127) // 128) * 4 * 2048
    pol = kk.load().kk_attn_bwd_two_pass
    assert pol(8, 8, 512, 512, 0) == 0 and pol(8, 8, 1024, 1024, 1) == 0 and pol(8, 8, 64, 64, 0) == 0      # (and nowhere beside the third-generation pair launch)
    ws = torch.full((need // 2,), float("nan"), device="cuda", dtype=torch.bfloat16)

    def run(two_pass, hn):
        pq, pkv = torch.full((1, nbq, 64), 5.0, device="cuda"), torch.full((2, nbk, 64), 5.0, device="cuda")
        dq, dkv = torch.full_like(raw_q, 7.0), torch.full_like(raw_kv, 7.0)
        hq = kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)]) if hn else None
        hkv = kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)]) if hn else None
        args = (q_n, k_n, v_n, do, lse, delta, dq, dkv, dkv[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km, causal, 0.125, seed, 5, p,
                1, 1, hq, hkv)
        if two_pass:
            ws.fill_(float("nan"))
            kk.call("kk_attn_bwd_ws", *args, ws, need)
        else:
            kk.call("kk_attn_bwd", *args)
        torch.cuda.synchronize()
        return dq, dkv, pq, pkv

    for hn in (True, False):
        ref, new = run(False, hn), run(True, hn)
        assert torch.equal(ref[1], new[1]) and torch.equal(ref[3].sort(dim=1).values, new[3].sort(dim=1).values), \
            f"hn={hn}: dK, dV (and the k / v gain partial rows, in the launch's own block order): the same kernel body"
        assert torch.isfinite(new[0].float()).all(), "every dS tile the pass read was written by this call"
        close(new[0], ref[0], 4e-2, 2e-2, f"hn={hn}: dQ")
        if hn:
            gq_new, gq_ref = new[2][0].sum(0), ref[2][0].sum(0)
            # (the epilogue rounds dQ to bf16 before the norm's backward, as the unfused path stores it: a different summation order flips
            #  the last bit of about half the elements, and the gain gradient sums B * Sq * h such rows per column)
            close(gq_new, gq_ref, 5e-3 * math.sqrt(B * Sq * h), 1e-2, "gain gradient q")
    # a workspace that is too small (or none): the pair launch runs, bit for bit
    ref = run(False, True)
    dq, dkv = torch.full_like(raw_q, 7.0), torch.full_like(raw_kv, 7.0)
    pq, pkv = torch.full((1, nbq, 64), 5.0, device="cuda"), torch.full((2, nbk, 64), 5.0, device="cuda")
    hq = kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)])
    hkv = kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)])
    kk.call("kk_attn_bwd_ws", q_n, k_n, v_n, do, lse, delta, dq, dkv, dkv[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km, causal, 0.125,
            seed, 5, p, 1, 1, hq, hkv, ws, need - 2048)
    torch.cuda.synchronize()
    assert torch.equal(dq, ref[0]) and torch.equal(dkv, ref[1]) and torch.equal(pq, ref[2])


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,S,causal,masked", [(8, 8, 512, 1, 0), (8, 8, 512, 0, 1), (2, 8, 1024, 1, 0), (2, 4, 1000, 0, 1), (3, 4, 200, 1, 1)])
def test_attention_backward_generations_agree(kk, B, h, S, causal, masked):
    """The third-generation backward (one wave group per workgroup, two workgroups per CU: what kk_attn_bwd_dq / _dkv / kk_attn_bwd run)
    against the FIRST-generation kernels (fp32-free register-staged path, taken here through fp32 storage of the same bf16 values): the
    same dropout masks, gradients within bf16 rounding of each other."""
    g = torch.Generator().manual_seed(B * S + causal + 5)
    H = h * 64
    q, kv = dev(torch.randn(B * S, H, generator=g)).bfloat16(), dev(torch.randn(B * S, 2 * H, generator=g)).bfloat16()
    do = dev(torch.randn(B * S, H, generator=g)).bfloat16()
    km = None
    if masked:
        km = torch.zeros(B, S, dtype=torch.uint8)
        km[:, S - 23:] = 1
        km = dev(km)
    seed = torch.tensor([7], dtype=torch.int32, device="cuda")
    o, lse = torch.empty_like(q), torch.empty(B, h, S, device="cuda")
    kk.call("kk_attn_fwd", q, kv, kv[:, H:], o, lse, B, h, S, S, H, 2 * H, 2 * H, H, km, causal, 0.125, seed, 5, 0.2, 1, 1)
    delta = torch.empty(B, h, S, device="cuda")
    kk.call("kk_attn_delta", o, do, delta, B, h, S, H, H, 1)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    kk.call("kk_attn_bwd", q, kv, kv[:, H:], do, lse, delta, dq, dkv, dkv[:, H:], B, h, S, S, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km, causal, 0.125, seed, 5,
            0.2, 1, 1, None, None)
    # first generation: fp32 storage of the same values (math mode bf16: operands rounded at staging, same MFMA arithmetic)
    qf, kvf, dof = q.float(), kv.float(), do.float()
    dqf, dkvf = torch.empty_like(qf), torch.empty_like(kvf)
    kk.call("kk_attn_bwd_dq", qf, kvf, kvf[:, H:], dof, lse, delta, dqf, B, h, S, S, H, 2 * H, 2 * H, H, H, km, causal, 0.125, seed, 5, 0.2, 1, 0, None, 0, None)
    kk.call("kk_attn_bwd_dkv", qf, kvf, kvf[:, H:], dof, lse, delta, dkvf, dkvf[:, H:], B, h, S, S, H, 2 * H, 2 * H, H, 2 * H, 2 * H, km, causal, 0.125,
            seed, 5, 0.2, 1, 0, None)
    torch.cuda.synchronize()
    close(dq, dqf, 3e-2, 2e-2, "dQ")
    close(dkv, dkvf, 3e-2, 2e-2, "dK | dV")


@pytest.mark.gpu
@pytest.mark.parametrize("B,S,h,K", [(8, 512, 8, 512), (4, 1024, 8, 512), (3, 700, 8, 192), (16, 512, 4, 256), (8, 64, 8, 512), (5, 41, 8, 512),
                                     (8, 1024, 8, 512), (11, 777, 8, 256),
                                     (8, 1024, 8, 1024), (11, 777, 8, 1088)])      # (the last two: K >= 1024 takes the 128x128 tile of kk_gemm16x.hip, route asserted)
def test_gemm_dgrad_delta_epilogue(kk, B, S, h, K):
    """kk_gemm_dgrad_delta == kk_gemm (dgrad, bf16 result: bit-identical) + kk_attn_delta on that result (fp32 row sums of
    the same rounded products, summed in a different order)."""
    g = torch.Generator().manual_seed(B * S + K)
    M, N = B * S, h * 64
    assert kk.load().kk_gemm_dgrad_delta_supported(M, N, K) == 1
    assert kk.load().kk_gemm_dgrad_delta_supported(512, N, K) == 1, "the encoder's 512-row output projections take the tile for the epilogue"
    dy = dev(torch.randn(M, K, generator=g)).bfloat16()
    W = dev(torch.randn(K, N, generator=g) / math.sqrt(K)).bfloat16()
    o = dev(torch.randn(M, N, generator=g)).bfloat16()
    dx_ref, dx = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.full((M, N), 3.0, device="cuda", dtype=torch.bfloat16)
    kk.call("kk_gemm", 0, 1, M, N, K, 1.0, dy, K, W, N, 0.0, dx_ref, N, None, None, 0, 0, 0, 1, 7)
    d_ref, d_new = torch.empty(B, h, S, device="cuda"), torch.full((B, h, S), 9.0, device="cuda")
    kk.call("kk_attn_delta", o, dx_ref, d_ref, B, h, S, N, N, 1)
    kk.call("kk_gemm_dgrad_delta", M, N, K, dy, K, W, N, dx, N, o, N, d_new, S, h)
    assert kk.last_kernel().startswith("g16x<0,1,128,128,3,0," if K >= 1024 else "gemm16_w8<0,1,"), kk.last_kernel()
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref), "the GEMM result must not change"
    want = (dx_ref.float() * o.float()).view(B, S, h, 64).sum(-1).permute(0, 2, 1)
    close(d_ref, want, 1e-4, 1e-4, "kk_attn_delta vs torch")
    close(d_new, want, 1e-4, 1e-4, "delta from the GEMM epilogue")


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,Sq,Sk,causal,rope,p,masked", [(1, 8, 512, 512, 1, 1, 0.1, 0), (2, 4, 512, 512, 0, 0, 0.2, 1), (1, 2, 1000, 1000, 1, 1, 0.2, 0),
                                                            (1, 2, 900, 1000, 0, 1, 0.1, 1), (2, 2, 300, 384, 0, 0, 0.0, 0), (2, 2, 200, 200, 1, 1, 0.0, 0),
                                                            (1, 2, 130, 400, 0, 0, 0.1, 0), (8, 8, 64, 64, 0, 1, 0.15, 1), (3, 4, 41, 41, 0, 1, 0.15, 1),
                                                            (2, 2, 64, 50, 0, 0, 0.0, 0),
                                                            (12, 8, 1333, 1333, 1, 1, 0.2, 0), (12, 8, 1333, 1333, 0, 0, 0.2, 1), (4, 8, 1024, 1024, 1, 1, 0.2, 1)])
def test_attention_backward_pair_launch(kk, B, h, Sq, Sk, causal, rope, p, masked):
    """kk_attn_bwd (dQ | dK, dV as the two halves of one grid, Delta an input) == kk_attn_bwd_dq + kk_attn_bwd_dkv given the
    same Delta: bit-identical gradients and gain-gradient partial rows (same code, only the launch differs).  The last case
    (different numbers of query and key blocks) takes the two-launch fall-back inside the entry point."""
    g = torch.Generator().manual_seed(B * Sq + Sk + causal)
    H = h * 64
    dt = torch.bfloat16
    raw_q = dev(torch.randn(B * Sq, H, generator=g)).to(dt)
    raw_kv = dev(torch.randn(B * Sk, 2 * H, generator=g)).to(dt)
    gains = [dev(1.0 + 0.2 * torch.randn(64, generator=g)) for _ in range(3)]
    c, s = ((dev(t) for t in O.rope_tables(max(Sq, Sk), 64)) if rope else (None, None))
    q_n, kv_n = torch.empty_like(raw_q), torch.empty_like(raw_kv)
    kk.call("kk_headnorm_rope_fwd", raw_q, H, q_n, H, B * Sq, h, Sq, 1, gains[0], None, None, 1 if rope else 0, c, s, 1)
    kk.call("kk_headnorm_rope_fwd", raw_kv, 2 * H, kv_n, 2 * H, B * Sk, h, Sk, 2, gains[1], gains[2], None, 1 if rope else 0, c, s, 1)
    k_n, v_n = kv_n, kv_n[:, H:]
    km = None
    if masked:
        km = torch.zeros(B, Sk, dtype=torch.uint8)
        km[:, Sk - 37:] = 1
        km[0, 5] = 1
        km = dev(km)
    do = dev(torch.randn(B * Sq, H, generator=g)).to(dt)
    o, lse = torch.empty(B * Sq, H, device="cuda", dtype=dt), torch.empty(B, h, Sq, device="cuda")
    seed = torch.tensor([91], dtype=torch.int32, device="cuda")
    kk.call("kk_attn_fwd", q_n, k_n, v_n, o, lse, B, h, Sq, Sk, H, 2 * H, 2 * H, H, km, causal, 0.125, seed, 5, p, 1, 1)
    delta = torch.empty(B, h, Sq, device="cuda")
    kk.call("kk_attn_delta", o, do, delta, B, h, Sq, H, H, 1)
    nbq, nbk = kk.load().kk_attn_bwd_blocks(B, h, Sq), kk.load().kk_attn_bwd_blocks(B, h, Sk)

    def run(pair):
        pq, pkv = torch.full((1, nbq, 64), 5.0, device="cuda"), torch.full((2, nbk, 64), 5.0, device="cuda")
        dq, dkv = torch.full_like(raw_q, 7.0), torch.full_like(raw_kv, 7.0)
        hq = kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)])
        hkv = kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)])
        if pair:
            kk.call("kk_attn_bwd", q_n, k_n, v_n, do, lse, delta, dq, dkv, dkv[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km,
                    causal, 0.125, seed, 5, p, 1, 1, hq, hkv)
        else:
            kk.call("kk_attn_bwd_dq", q_n, k_n, v_n, do, lse, delta, dq, B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, km, causal, 0.125, seed, 5, p,
                    1, 1, None, 0, hq)
            kk.call("kk_attn_bwd_dkv", q_n, k_n, v_n, do, lse, delta, dkv, dkv[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, 2 * H, 2 * H, km,
                    causal, 0.125, seed, 5, p, 1, 1, hkv)
        torch.cuda.synchronize()
        return dq, dkv, pq, pkv

    ref, new = run(False), run(True)
    for a_, b_, name in zip(ref, new, ("d raw q", "d raw k|v", "gain partials q", "gain partials k, v")):
        if name.startswith("gain"):      # (a causal pair launch hands its dK/dV blocks out in another order: the same partial rows, permuted)
            a_, b_ = a_.sort(dim=1).values, b_.sort(dim=1).values
        assert torch.equal(a_, b_), f"{name}: the pair launch must reproduce the two launches bit for bit"
    assert torch.isfinite(new[0].float()).all() and torch.isfinite(new[1].float()).all()
    # without the head-norm epilogues
    dq_a, dkv_a = torch.full_like(raw_q, 7.0), torch.full_like(raw_kv, 7.0)
    dq_b, dkv_b = torch.full_like(raw_q, 7.0), torch.full_like(raw_kv, 7.0)
    kk.call("kk_attn_bwd_dq", q_n, k_n, v_n, do, lse, delta, dq_a, B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, km, causal, 0.125, seed, 5, p, 1, 1, None, 0, None)
    kk.call("kk_attn_bwd_dkv", q_n, k_n, v_n, do, lse, delta, dkv_a, dkv_a[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, 2 * H, 2 * H, km, causal, 0.125,
            seed, 5, p, 1, 1, None)
    kk.call("kk_attn_bwd", q_n, k_n, v_n, do, lse, delta, dq_b, dkv_b, dkv_b[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km, causal,
            0.125, seed, 5, p, 1, 1, None, None)
    torch.cuda.synchronize()
    assert torch.equal(dq_a, dq_b) and torch.equal(dkv_a, dkv_b), "plain gradients: pair launch == two launches"
    # round 5: the forward that stores the keep decisions + the pair launch that reads them (kk_attn_fwd_kb / kk_attn_bwd_kb): the same
    # output, log-sum-exp rows and gradients bit for bit — with and without the head-norm epilogues, ragged sequences included
    nbytes = kk.load().kk_attn_keep_bytes(B, h, Sq, Sk)
    assert (nbytes > 0) == (Sk > 128)
    if nbytes > 0 and p > 0.0:
        keep = torch.full((nbytes,), 0xA5, dtype=torch.uint8, device="cuda")
        o2, lse2 = torch.empty_like(o), torch.empty_like(lse)
        kk.call("kk_attn_fwd_kb", q_n, k_n, v_n, o2, lse2, B, h, Sq, Sk, H, 2 * H, 2 * H, H, km, causal, 0.125, seed, 5, p, 1, 1, keep)
        assert kk.last_kernel().startswith("attn_fwd3_q")
        assert torch.equal(o2, o) and torch.equal(lse2, lse)
        pair_ok = -(-Sq // 128) == -(-Sk // 128)
        dq_k, dkv_k = torch.full_like(raw_q, 7.0), torch.full_like(raw_kv, 7.0)
        kk.call("kk_attn_bwd_kb", q_n, k_n, v_n, do, lse, delta, dq_k, dkv_k, dkv_k[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km,
                causal, 0.125, seed, 5, p, 1, 1, None, None, keep)
        assert kk.last_kernel() == ("attn_bwd_pair3k" if pair_ok else "attn_bwd_dkv3"), kk.last_kernel()
        pq, pkv = torch.full((1, nbq, 64), 5.0, device="cuda"), torch.full((2, nbk, 64), 5.0, device="cuda")
        dq_h, dkv_h = torch.full_like(raw_q, 7.0), torch.full_like(raw_kv, 7.0)
        hq = kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)])
        hkv = kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)])
        kk.call("kk_attn_bwd_kb", q_n, k_n, v_n, do, lse, delta, dq_h, dkv_h, dkv_h[:, H:], B, h, Sq, Sk, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km,
                causal, 0.125, seed, 5, p, 1, 1, hq, hkv, keep)
        torch.cuda.synchronize()
        assert torch.equal(dq_k, dq_b) and torch.equal(dkv_k, dkv_b), "plain gradients: reading the keep bits == hashing"
        assert torch.equal(dq_h, new[0]) and torch.equal(dkv_h, new[1]), "head-norm epilogues: reading the keep bits == hashing"
        assert torch.equal(pq, new[2]) and torch.equal(pkv, new[3])



def test_cast_ranges_bucket_payload(kk):
    """kk_cast_ranges: every range of a gradient bucket narrowed to its bf16 payload twin (and widened back, scaled) by one thin launch;
    elements outside the ranges untouched; more ranges than one launch's table holds; ragged lengths."""
    import ctypes as C
    g = torch.Generator().manual_seed(3)
    n_el = 3_000_000
    src = dev(torch.randn(n_el, generator=g))
    ranges, pos = [], 0
    for i in range(61):                                            # > 48 ranges: two launches
        ln = int(torch.randint(1, 60000, (1,), generator=g)) if i % 5 else 4096 * 7
        ranges.append((pos, pos + ln))
        pos += ((ln + 1023) // 1024 + (i % 3)) * 1024
    assert pos <= n_el
    beg, end = (C.c_int64 * len(ranges))(*[b for b, _ in ranges]), (C.c_int64 * len(ranges))(*[e for _, e in ranges])
    d16 = torch.full((n_el,), 7.0, device="cuda", dtype=torch.bfloat16)
    kk.call("kk_cast_ranges", src, d16, beg, end, len(ranges), 1, 1.0)
    want = torch.full((n_el,), 7.0, device="cuda", dtype=torch.bfloat16)
    for b, e in ranges:
        want[b:e] = src[b:e].bfloat16()
    assert torch.equal(d16, want)
    back = torch.full((n_el,), -3.0, device="cuda")
    kk.call("kk_cast_ranges", d16, back, beg, end, len(ranges), 0, 0.5)
    wantf = torch.full((n_el,), -3.0, device="cuda")
    for b, e in ranges:
        wantf[b:e] = d16[b:e].float() * 0.5
    assert torch.equal(back, wantf)
