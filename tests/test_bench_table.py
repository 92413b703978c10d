"""CPU suite: bench.py's kernel table (the roofline leg's bookkeeping) — names and algorithmic FLOPs / bytes of the launches
the engine makes at the bench shape, from synthetic profile records (no GPU)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("kk_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_kernel_table_accounts_for_the_attention_backward_pair_and_its_delta_gemm():
    b = _bench()
    B, h, S, H = 8, 8, 512, 512
    recs = [
        # kk_attn_bwd: (B, h, Sq, Sk, 7 row strides, causal, scale, site, p_drop, math, io_bf16)
        ("kk_attn_bwd", (B, h, S, S, 1536, 1536, 1536, H, 1536, 1536, 1536, 1, 0.125, 2003, 0.2, 1, 1), 0.035),
        ("kk_attn_bwd", (B, h, S, S, H, 6144, 6144, H, H, 6144, 6144, 0, 0.125, 2011, 0.2, 1, 1), 0.041),
        # kk_gemm_dgrad_delta: (M, N, K, lddy, ldw, lddx, ldo, S, heads)
        ("kk_gemm_dgrad_delta", (B * S, H, H, H, H, H, H, S, h), 0.008),
        # the plain dgrad of the same shape lands on the same kernel instantiation
        ("kk_gemm", (0, 1, B * S, H, H, 1.0, H, H, 0.0, H, 0, 0, 0, 1, 7), 0.0075),
        ("kk_gemm_qkv_headnorm", (B * S, 3, h, H, H, 1536, 1536, S, 3), 0.020),
        ("kk_gemm_qkv_headnorm", (512, 3, h, H, H, 1536, 1536, 64, 3), 0.010),
        ("kk_gemm_dgrad_glu", (B * S, 1536, H, H, 2003, 0.2), 0.027),
        # round 5: the entry points with the stored dropout keep bits (the buffer is a tensor: same scalars as the plain ones)
        ("kk_attn_bwd_kb", (B, h, S, S, 1536, 1536, 1536, H, 1536, 1536, 1536, 1, 0.125, 2003, 0.2, 1, 1), 0.030),
        ("kk_attn_fwd_kb", (B, h, S, S, 1536, 1536, 1536, H, 1, 0.125, 2003, 0.2, 1, 1), 0.012),
        ("kk_attn_bwd_kb", (B, h, 64, 64, 1536, 1536, 1536, H, 1536, 1536, 1536, 0, 0.125, 2003, 0.15, 1, 1), 0.013),      # one tile: no bits, the hashing kernel
    ]
    t = b.kernel_table(recs, True)
    pair = t["attn_bwd_pair3_kernel (dQ | dK, dV in one launch, two workgroups per CU)"]
    full = 4 * 2.0 * B * h * S * S * 64            # the backward of a 2-matmul forward is 4 matmuls; recomputation earns nothing
    assert pair["launches"] == 2 and abs(pair["flops"] - (0.5 * full + full)) < 1.0          # causal = lower triangle
    assert pair["bytes"] == 2 * 2.0 * B * h * 64 * 8 * S
    pk = t["attn_bwd_pair3k_kernel (dQ | dK, dV in one launch, two workgroups per CU, stored dropout keep bits)"]
    assert pk["launches"] == 1 and abs(pk["flops"] - 0.5 * full) < 1.0 and pk["bytes"] == 2.0 * B * h * 64 * 8 * S + 2 * B * h * S * S / 8 * 0.5
    fk = t["attn_fwd3_q64_kernel (flash forward, 2 workgroups per CU, 64-query blocks x 4 key slots)"]
    assert fk["launches"] == 1 and abs(fk["flops"] - 0.25 * full) < 1.0
    assert t["attn_bwd_pair3_kernel (dQ | dK, dV in one launch, two workgroups per CU), one-tile sequences (text encoder)"]["launches"] == 1
    w8 = [k for k in t if k.startswith("gemm16_kernel_w8<false,true,3>")]
    assert len(w8) == 1 and t[w8[0]]["launches"] == 2 and t[w8[0]]["flops"] == 2 * 2.0 * B * S * H * H
    # the decoder's q|k|v projection and the linear2 dgrad + GLU' take the large-tile family (kk_gemm16x.hip) at 4096 rows ...
    hn = t["g16x_kernel<false,false,128,192,3,3,2,2,4> (q|k|v projection + head-norm epilogue, 128x192 tiles)"]
    assert hn["launches"] == 1 and abs(hn["floor_us"] - 2.0 * B * S * 1536 * H * (1 / 128 + 1 / 192) / (256 * 54.0 * 2.4e9) * 1e6) < 1e-6
    assert t["gemm16_kernel<false,false,64,64,3,3> (q|k|v projection + head-norm epilogue)"]["launches"] == 1      # the encoder's 512 rows
    assert t["g16x_kernel<false,true,128,192,3,1,2,2,4> (dY.W2 + GLU backward epilogue, 128x192 tiles, loader waves)"]["flops"] == 2.0 * B * S * 1536 * H
    # ... the K = 512 plain GEMMs do not (long reductions only), and the formulas mirror kk_gemm16.hip's policy
    assert b.x_tile("plain1", 4096, 512, 512) is None and b.x_tile("plain1", 4096, 512, 3072) is None
    assert b.x_tile("plain1", 8192, 512, 3072)[:2] == (128, 128) and b.x_tile("hn", 8192, 1536, 512)[:2] == (256, 192)
    assert b.x_tile("hn", 4096, 512, 512) is None and b.x_tile("glu_fwd", 4096, 1536, 512)[:2] == (256, 192)
    assert abs(b.attn_ffn_flops(8, 1024, 128) / b.train_flops(8, 1024, 128) - 0.967) < 2e-3          # 96.7 % of the total at cfg-4
    shapes = [k for k in t if k.startswith("  shape")]
    assert any("kk_attn_bwd B=8 h=8 Sq=512 Sk=512 causal=1" in k for k in shapes)
    assert sum("ta=0 tb=1 M=4096 N=512 K=512" in k for k in shapes) == 1


def test_train_flops_formula_reproduces_the_survey_table():
    """SURVEY 8d / BASELINE.md section 4: 870.1 GFLOP per step at 8x512x64, 1974.4 GFLOP at 8x1024x128."""
    b = _bench()
    assert abs(b.train_flops(8, 512, 64) / 1e9 - 870.1) < 0.1
    assert abs(b.train_flops(8, 1024, 128) / 1e9 - 1974.4) < 0.1
    assert abs(b.train_flops(8, 512, 64) / (8 * 512) / 1e6 - 212.4) < 0.1


def test_ragged_workload_respects_the_frame_budget(monkeypatch):
    """configs[2]'s resident workload: every batch B * T <= 16384, 4 <= B <= 32, ragged lengths, durations summing to the lengths."""
    import torch
    b = _bench()
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    batches = b.ragged_workload(n_utts=120)
    assert len(batches) >= 8
    for x in batches:
        B, T = x["mel_specs"].shape[:2]
        assert B * T <= 16384 and 4 <= B <= 32 and int(x["mel_lengths"].max()) == T
        assert torch.equal(x["phoneme_durations"].sum(1), x["mel_lengths"])
    assert len({tuple(x["mel_specs"].shape[:2]) for x in batches}) >= 6


def test_bench_detaches_the_in_step_exchange_before_rank0_only_legs():
    """After the timed region rank 0 runs the eager roofline leg ALONE while the other ranks wait at a barrier: with eng.dp_comm still
    set, its forward_backward would issue RCCL collectives that no other rank matches (a hang at N > 1 that a 1-rank rehearsal cannot
    show).  bench.py must drop the exchange before anything rank-0-only launches kernels — checked on the source, in order."""
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    timed = src.index("regions.append(float(dt_r))")
    detach = src.index("eng.dp_comm = None", timed)
    first_alone = min(src.index("roofline_leg(eng, kk, [batch]", timed), src.index("stack_fraction(eng, batch)", timed),
                      src.index("extra_shapes(eng, kk, args.math)", timed))
    assert timed < detach < first_alone
