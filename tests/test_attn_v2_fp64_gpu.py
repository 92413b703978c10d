"""GPU suite: the kernels the bf16 bench path ACTUALLY runs — attn_fwd3 (attn_fwd2 up to 128 keys), attn_bwd_pair3 (kk_attn_bwd), kk_gemm_dgrad_delta — against
direct fp64 torch references at the bench shapes (8 x 8 heads x 512^2 and 1024^2), with everything the step turns on: attention-
probability dropout, causal or key-padding masks, RoPE, and the head-norm backward epilogues.  (until now these kernels
were compared with the repository's own first-generation kernels only.)

The reference is reference-shaped, not kernel-shaped: softmax(q k^T / 8 + masks) * dropout mask @ v in float64 through torch autograd
(model/transformers.py:393-398 = F.scaled_dot_product_attention with dropout_p; :260-277 the per-head RMSNorm + RoPE in front of it).
The dropout mask is not an input of the C ABI — it is a hash of (seed, site, batch, head, query, key) — so it is RECOVERED from the
kernel under test: with V = [one-hot rows for one 64-key block, zero elsewhere] the output of the forward IS the dropped probability
matrix of that block (the trick of test_attention_probability_dropout, block by block), and the same launch without dropout gives
the undropped one; their ratio is the mask.  A mask bias or a forward / backward mask disagreement shows up as an O(1) error."""
import math

import pytest
import torch

from oracle import kokoro_oracle as O

pytestmark = pytest.mark.gpu

F32_EPS = float(torch.finfo(torch.float32).eps)


@pytest.fixture(scope="module")
def kk():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from kokoro_ruslan_amd import lib
    lib.load()
    return lib


def _rel(got, ref):
    got, ref = got.double(), ref.double()
    return float((got - ref).norm() / ref.norm().clamp(min=1e-30))


def _cos(got, ref):
    got, ref = got.double().flatten(), ref.double().flatten()
    return float(got @ ref / (got.norm() * ref.norm()).clamp(min=1e-30))


def decode_keep_bits(buf, B, h, Sq, Sk):
    """[B, h, Sq', Sk'] bool from the packed keep decisions of kk_attn_fwd_kb (include/kokoro_hip.h: 128 bytes per 32 x 32 unit,
    dword 2 r + hf = bits over the unit's queries for key (r & 3) + 8 (r >> 2) + 4 hf), Sq' / Sk' rounded up to 32."""
    nq, nk = -(-Sq // 32), -(-Sk // 32)
    w = buf.view(torch.int32).view(B, h, nq, nk, 32)
    bits = ((w.unsqueeze(-1) >> torch.arange(32, device=buf.device, dtype=torch.int32)) & 1).bool()      # [.., dword, query]
    d = torch.arange(32, device=buf.device)
    r, hf = d >> 1, d & 1
    key_of = (r & 3) + 8 * (r >> 2) + 4 * hf
    by_key = torch.empty_like(bits)
    by_key[..., key_of, :] = bits                                                                        # [.., key, query]
    return by_key.permute(0, 1, 2, 5, 3, 4).reshape(B, h, nq * 32, nk * 32)[:, :, :Sq, :Sk]


def _headnorm64(raw, gain, cos, sin, B, S, h):
    """fp64 per-head RMSNorm (eps of fp32, as nn.RMSNorm(eps=None) on fp32 activations) * gain, then RoPE (rotate-half)."""
    x = raw.view(B, S, h, 64)
    n = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + F32_EPS) * gain
    if cos is not None:
        n = n * cos[None, :S, None, :] + O._rotate_half(n) * sin[None, :S, None, :]
    return n.reshape(B * S, h * 64)


@pytest.mark.parametrize("S,causal,masked", [(512, 1, 0), (512, 0, 1), (1024, 1, 0), (1024, 0, 1)])
def test_v2_attention_forward_and_pair_backward_against_fp64(kk, S, causal, masked):
    B, h, p, site = 8, 8, 0.2, 5
    H = h * 64
    g = torch.Generator().manual_seed(100 * S + causal)
    rope = bool(causal)                                   # decoder self-attention: causal + RoPE; cross-attention: key mask, no RoPE
    bf = torch.bfloat16
    raw_q = torch.randn(B * S, H, generator=g).cuda().to(bf)
    raw_kv = torch.randn(B * S, 2 * H, generator=g).cuda().to(bf)
    gains = [(1.0 + 0.2 * torch.randn(64, generator=g)).cuda() for _ in range(3)]
    cos, sin = ((t.cuda() for t in O.rope_tables(S, 64)) if rope else (None, None))
    q_n, kv_n = torch.empty_like(raw_q), torch.empty_like(raw_kv)
    kk.call("kk_headnorm_rope_fwd", raw_q, H, q_n, H, B * S, h, S, 1, gains[0], None, None, 1 if rope else 0, cos, sin, 1)
    kk.call("kk_headnorm_rope_fwd", raw_kv, 2 * H, kv_n, 2 * H, B * S, h, S, 2, gains[1], gains[2], None, 1 if rope else 0, cos, sin, 1)
    k_n, v_n = kv_n, kv_n[:, H:]
    # the head norm itself against fp64 (one bf16 rounding of O(1) values)
    for got, raw, gi, rp in ((q_n, raw_q, 0, rope), (kv_n[:, :H], raw_kv[:, :H], 1, rope), (kv_n[:, H:], raw_kv[:, H:], 2, False)):
        ref = _headnorm64(raw.double(), gains[gi].double(), cos.double() if rp else None, sin.double() if rp else None, B, S, h)
        assert _rel(got, ref) < 4e-3, "per-head RMSNorm + RoPE forward"
    km = None
    if masked:
        kmh = torch.zeros(B, S, dtype=torch.uint8)
        for b in range(B):
            kmh[b, S - 17 * (b + 1):] = 1                 # ragged padding, like mel lengths
        kmh[0, 5] = 1
        km = kmh.cuda()
    seed = torch.tensor([91], dtype=torch.int32, device="cuda")
    thr = int(p * 65536.0 + 0.5)
    inv_keep = 65536.0 / (65536 - thr)                    # the kernels' quantised 1 / (1 - p) (kk_attn.hip ProbDrop)

    def fwd(v, pdrop, out, lse):
        kk.call("kk_attn_fwd", q_n, k_n, v, out, lse, B, h, S, S, H, 2 * H, v.stride(0), H, km, causal, 0.125, seed, site, pdrop, 1, 1)

    # ---- recover the dropped and the undropped probabilities block by block (V = one-hot rows of one 64-key block)
    lse_ = torch.empty(B, h, S, device="cuda")
    P0 = torch.empty(B, h, S, S, device="cuda")
    P1 = torch.empty(B, h, S, S, device="cuda")
    eye = torch.eye(64, device="cuda", dtype=bf).repeat(1, h)               # [64, H]: every head's block is I
    for j in range(S // 64):
        vj = torch.zeros(B, S, H, device="cuda", dtype=bf)
        vj[:, 64 * j:64 * (j + 1)] = eye
        vj = vj.view(B * S, H)
        for pd, dst in ((0.0, P0), (p, P1)):
            o = torch.empty(B * S, H, device="cuda", dtype=bf)
            fwd(vj, pd, o, lse_)
            dst[:, :, :, 64 * j:64 * (j + 1)] = o.view(B, S, h, 64).permute(0, 2, 1, 3).float()
    known = P0 > 1e-6
    ratio = P1 / P0.clamp(min=1e-30)
    keep = ratio > 0.5 * inv_keep
    r = ratio[known]
    assert bool((((r - inv_keep).abs() < 0.03) | (r.abs() < 1e-3)).all()), "a dropped probability is 0 or P / (1 - p), nothing else"
    rate = float(keep[known].float().mean())
    assert abs(rate - (1 - p)) < 2e-3, f"keep rate {rate}"
    # no bias along queries, keys, heads: every marginal keep rate within 4 sigma
    for dim in ((0, 1, 3), (0, 1, 2), (0, 2, 3)):
        cnt = known.float().sum(dim)
        kr = (keep & known).float().sum(dim) / cnt.clamp(min=1)
        sig = (p * (1 - p) / cnt.clamp(min=1)).sqrt()
        assert bool((((kr - (1 - p)).abs() < 4.5 * sig + 2e-3) | (cnt < 64)).all()), f"keep-rate bias along dims {dim}"
    mask = torch.where(keep, inv_keep, 0.0).double()
    mask = torch.where(known, mask, torch.full_like(mask, 1.0))     # (P0 == 0: masked or underflowed, contributes nothing)

    # ---- round 5: the forward that STORES the keep decisions (kk_attn_fwd_kb): the stored bits are the recovered mask, and the launch
    # changes nothing else
    nbytes = kk.load().kk_attn_keep_bytes(B, h, S, S)
    assert nbytes == B * h * (S // 32) ** 2 * 128
    keep_buf = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device="cuda")
    o_kb, lse_kb = torch.empty(B * S, H, device="cuda", dtype=bf), torch.empty(B, h, S, device="cuda")
    kk.call("kk_attn_fwd_kb", q_n, k_n, v_n, o_kb, lse_kb, B, h, S, S, H, 2 * H, v_n.stride(0), H, km, causal, 0.125, seed, site, p, 1, 1, keep_buf)
    assert kk.last_kernel() == ("attn_fwd3_q128" if (S // 128) * B * h >= 512 else "attn_fwd3_q64")
    stored = decode_keep_bits(keep_buf, B, h, S, S)
    assert bool((stored == keep)[known].all()), "stored keep bits == the mask the kernel applied"

    # ---- fp64 reference of the attention proper, on the kernel's own (bf16) inputs
    hd = lambda x: x.view(B, S, h, 64).transpose(1, 2)
    qr, kr_, vr = (t.double().clone().requires_grad_(True) for t in (q_n, k_n[:, :H], v_n[:, :H]))
    s = hd(qr) @ hd(kr_).transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.triu(torch.full((S, S), float("-inf"), dtype=s.dtype, device="cuda"), 1)
    if km is not None:
        s = s.masked_fill(km.bool()[:, None, None, :], float("-inf"))
    Pref = torch.softmax(s, -1)
    # the undropped probabilities the kernel produced (bf16 outputs of an fp32 softmax)
    assert float((P0.double() - Pref).abs().max()) < 4e-3 and _rel(P0, Pref.detach()) < 6e-3, "softmax probabilities"
    out_ref = ((Pref * mask) @ hd(vr)).transpose(1, 2).reshape(B * S, H)
    lse_ref = torch.logsumexp(s, -1)

    # ---- forward under test (general V)
    o = torch.empty(B * S, H, device="cuda", dtype=bf)
    lse = torch.empty(B, h, S, device="cuda")
    fwd(v_n, p, o, lse)
    assert torch.equal(o, o_kb) and torch.equal(lse, lse_kb), "storing the keep bits must not change the forward"
    assert _rel(o, out_ref.detach()) < 8e-3 and _cos(o, out_ref.detach()) > 0.9999, ("attention output", _rel(o, out_ref.detach()))
    assert float((o.double() - out_ref.detach()).abs().max()) < 6e-2
    assert float((lse.double() - lse_ref.detach()).abs().max()) < 2e-3, "log-sum-exp rows"

    # ---- backward reference: dO -> d(q_n, k_n, v_n) by autograd, then through the head norm at the raw projections
    do = torch.randn(B * S, H, generator=g).cuda().to(bf)
    out_ref.backward(do.double())
    raws = [raw_q.double().clone().requires_grad_(True), raw_kv[:, :H].double().clone().requires_grad_(True),
            raw_kv[:, H:].double().clone().requires_grad_(True)]
    gd = [t.double().clone().requires_grad_(True) for t in gains]
    want_raw, want_gain = [], []
    for raw, g64, up, rp in zip(raws, gd, (qr.grad, kr_.grad, vr.grad), (rope, rope, False)):
        n = _headnorm64(raw, g64, cos.double() if rp else None, sin.double() if rp else None, B, S, h)
        dr, dg = torch.autograd.grad(n, (raw, g64), grad_outputs=up)
        want_raw.append(dr)
        want_gain.append(dg)

    # ---- kk_gemm_dgrad_delta (the w_o dgrad that produces dO and Delta in the step) against fp64, on its own operands
    dy = torch.randn(B * S, H, generator=g).cuda().to(bf)
    Wo = (torch.randn(H, H, generator=g) / math.sqrt(H)).cuda().to(bf)
    dctx = torch.empty(B * S, H, device="cuda", dtype=bf)
    delta_g = torch.empty(B, h, S, device="cuda")
    assert kk.load().kk_gemm_dgrad_delta_supported(B * S, H, H) == 1
    kk.call("kk_gemm_dgrad_delta", B * S, H, H, dy, H, Wo, H, dctx, H, o, H, delta_g, S, h)
    dctx_ref = dy.double() @ Wo.double()
    assert _rel(dctx, dctx_ref) < 4e-3, "w_o dgrad"
    d_ref = (dctx_ref * o.double()).view(B, S, h, 64).sum(-1).permute(0, 2, 1)
    assert float((delta_g.double() - d_ref).abs().max()) < 3e-2 * float(d_ref.abs().max()), "Delta from the GEMM epilogue"

    # ---- the pair launch under test: Delta of (dO, O) as the step computes it, head-norm epilogues on
    delta = torch.empty(B, h, S, device="cuda")
    kk.call("kk_attn_delta", o, do, delta, B, h, S, H, H, 1)
    d_want = (do.double() * o.double()).view(B, S, h, 64).sum(-1).permute(0, 2, 1)
    assert float((delta.double() - d_want).abs().max()) < 1e-3 * max(1.0, float(d_want.abs().max()))
    nb = kk.load().kk_attn_bwd_blocks(B, h, S)
    pq, pkv = torch.zeros(1, nb, 64, device="cuda"), torch.zeros(2, nb, 64, device="cuda")
    dq, dkv = torch.full_like(raw_q, 7.0), torch.full_like(raw_kv, 7.0)
    hq = kk.attn_headnorm([(raw_q, gains[0], pq[0], cos, sin)])
    hkv = kk.attn_headnorm([(raw_kv, gains[1], pkv[0], cos, sin), (raw_kv[:, H:], gains[2], pkv[1], None, None)])
    kk.call("kk_attn_bwd", q_n, k_n, v_n, do, lse, delta, dq, dkv, dkv[:, H:], B, h, S, S, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km,
            causal, 0.125, seed, site, p, 1, 1, hq, hkv)
    dgs = [torch.zeros(64, device="cuda") for _ in range(3)]
    kk.call("kk_partials_reduce", kk.reduce_table([(pq[0], dgs[0], None, nb, 64, 64), (pkv[0], dgs[1], None, nb, 64, 64),
                                                   (pkv[1], dgs[2], None, nb, 64, 64)], "cuda"), 3, 64)
    torch.cuda.synchronize()
    got_raw = (dq, dkv[:, :H], dkv[:, H:])
    for name, got, want in zip(("d raw q", "d raw k", "d raw v"), got_raw, want_raw):
        assert bool(torch.isfinite(got.float()).all()), name
        rel, cs = _rel(got, want), _cos(got, want)
        assert rel < 1.5e-2 and cs > 0.9998, (name, rel, cs)
        assert float((got.double() - want).abs().max()) < 0.05 * float(want.abs().max()) + 1e-3, name
    for name, got, want in zip("qkv", dgs, want_gain):
        assert _rel(got, want) < 1e-2, (f"gain gradient {name}", _rel(got, want))
    # ---- round 5: the pair launch that READS the stored decisions (kk_attn_bwd_kb -> attn_bwd_pair3k): bit-identical to the hashing one
    assert kk.last_kernel() != "attn_bwd_pair3k"
    pq2, pkv2 = torch.zeros(1, nb, 64, device="cuda"), torch.zeros(2, nb, 64, device="cuda")
    dq2, dkv2 = torch.full_like(raw_q, 7.0), torch.full_like(raw_kv, 7.0)
    hq2 = kk.attn_headnorm([(raw_q, gains[0], pq2[0], cos, sin)])
    hkv2 = kk.attn_headnorm([(raw_kv, gains[1], pkv2[0], cos, sin), (raw_kv[:, H:], gains[2], pkv2[1], None, None)])
    kk.call("kk_attn_bwd_kb", q_n, k_n, v_n, do, lse, delta, dq2, dkv2, dkv2[:, H:], B, h, S, S, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, km,
            causal, 0.125, seed, site, p, 1, 1, hq2, hkv2, keep_buf)
    assert kk.last_kernel() == "attn_bwd_pair3k"
    torch.cuda.synchronize()
    assert torch.equal(dq2, dq) and torch.equal(dkv2, dkv), "reading the keep bits == hashing them again"
    assert torch.equal(pq2, pq) and torch.equal(pkv2, pkv)
