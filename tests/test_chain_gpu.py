"""GPU suite: the Stage-0 persistent sub-layer launch (csrc/kk_chain.hip) stores the bits of the four launches it
records — the decoder's self-attention sub-layer forward at 8 x 512 x hidden 512, dropout on, keep bits stored — and refuses what it
does not carry.  The engine does not use it (profiles/r06_xcd_affine_probe.txt: 0.86x); the test keeps the mechanism honest."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kk():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from kokoro_ruslan_amd import lib
    lib.load()
    return lib


def _setup(kk, T):
    from kokoro_ruslan_amd import spec
    B, H, h = 8, 512, 8
    N = B * T
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
    bf = torch.bfloat16
    t = dict(n1=rnd(N, H).to(bf), Wqkv=rnd(3 * H, H, sc=H ** -0.5).to(bf), Wo=rnd(H, H, sc=H ** -0.5).to(bf), bo=rnd(H, sc=0.1),
             gq=1 + rnd(64, sc=0.1), gk=1 + rnd(64, sc=0.1), gv=1 + rnd(64, sc=0.1), lng=1 + rnd(H, sc=0.1), lnb=rnd(H, sc=0.1),
             x_res=rnd(N, H), seed=torch.tensor([1234], dtype=torch.int32, device="cuda"))
    cos, sin = (x.cuda() for x in spec.rope_tables(4000, 64))
    t["cos"], t["sin"] = cos[:T], sin[:T]
    t["ptrs"] = kk.pointer_table([t["gq"], t["gk"], t["gv"]])
    t["keep_bytes"] = kk.load().kk_attn_keep_bytes(B, h, T, T)
    return B, H, h, N, t


def _outputs(B, H, h, N, T, keep_bytes):
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="cuda")
    bf = torch.bfloat16
    return dict(raw=z(N, 3 * H, dt=bf), nrm=z(N, 3 * H, dt=bf), ctx=z(N, H, dt=bf), lse=z(B, h, T), keep=z(max(keep_bytes, 16), dt=torch.uint8),
                proj=z(N, H, dt=bf), x_out=z(N, H), n=z(N, H, dt=bf), mean=z(N), rstd=z(N))


def _sublayer(kk, o, t, B, H, h, N, T, causal=1):
    p, dpr = 0.2, 0.05
    kk.call("kk_gemm_qkv_headnorm", N, 3, h, H, t["n1"], H, t["Wqkv"], None, o["raw"], 3 * H, o["nrm"], 3 * H, T, t["ptrs"], 3, t["cos"], t["sin"])
    q, k, v = o["nrm"], o["nrm"][:, H:], o["nrm"][:, 2 * H:]
    kk.call("kk_attn_fwd_kb", q, k, v, o["ctx"], o["lse"], B, h, T, T, 3 * H, 3 * H, 3 * H, H, None, causal, 0.125, t["seed"], 2003, p,
            kk.KK_MATH_BF16, 1, o["keep"] if t["keep_bytes"] else None)
    kk.call("kk_gemm", 0, 0, N, H, H, 1.0, o["ctx"], H, t["Wo"], H, 0.0, o["proj"], H, t["bo"], None, 0, 0, 0, kk.KK_MATH_BF16, 1 | 2 | 4)
    kk.call("kk_sublayer_out_fwd", o["proj"], 1, None, None, t["x_res"], o["x_out"], t["lng"], t["lnb"], o["n"], 1, o["mean"], o["rstd"], N, H, T,
            t["seed"], 2000, p, 2001, 0.0, 2002, dpr)


def test_chained_sublayer_launch_is_bit_identical_to_its_four_launches(kk):
    T = 512
    B, H, h, N, t = _setup(kk, T)
    a, c = _outputs(B, H, h, N, T, t["keep_bytes"]), _outputs(B, H, h, N, T, t["keep_bytes"])
    _sublayer(kk, a, t, B, H, h, N, T)
    sync = torch.zeros(512, dtype=torch.int32, device="cuda")
    for flags in (0, 1):                                   # XCD-local and agent-scope barrier atomics
        for k in c:
            c[k].zero_()
        kk.call("kk_chain_begin")
        _sublayer(kk, c, t, B, H, h, N, T)                 # recorded, not launched
        torch.cuda.synchronize()
        assert float(c["n"].float().abs().sum()) == 0.0, "between begin and launch nothing runs"
        kk.call("kk_chain_launch", 0, sync, None, flags)
        assert kk.last_kernel() == "chain_sa_fwd<0>"
        torch.cuda.synchronize()
        assert int(sync[0]) == 0, "a group barrier timed out"
        for k in a:
            assert torch.equal(a[k], c[k]), f"{k}: the chained launch must store the four launches' bits (flags {flags})"
    # 40 back-to-back launches (the counters reset themselves; placement rotates between launches): still the same bits
    for _ in range(40):
        kk.call("kk_chain_begin")
        _sublayer(kk, c, t, B, H, h, N, T)
        kk.call("kk_chain_launch", 0, sync, None, 0)
    torch.cuda.synchronize()
    assert int(sync[0]) == 0 and all(torch.equal(a[k], c[k]) for k in a)


def test_chained_launch_refuses_what_it_does_not_carry(kk):
    T = 512
    B, H, h, N, t = _setup(kk, T)
    c = _outputs(B, H, h, N, T, t["keep_bytes"])
    sync = torch.zeros(512, dtype=torch.int32, device="cuda")
    kk.call("kk_chain_begin")
    _sublayer(kk, c, t, B, H, h, N, T, causal=0)           # full attention: not the self-attention chain
    with pytest.raises(RuntimeError, match="not one item per XCD|not a chain"):
        kk.call("kk_chain_launch", 0, sync, None, 0)
    kk.call("kk_chain_begin")
    kk.call("kk_gemm", 0, 0, N, H, H, 1.0, c["ctx"], H, t["Wo"], H, 0.0, c["proj"], H, t["bo"], None, 0, 0, 0, kk.KK_MATH_BF16, 1 | 2 | 4)
    with pytest.raises(RuntimeError, match="launches recorded"):
        kk.call("kk_chain_launch", 0, sync, None, 0)
    # after a refusal the thread launches normally again
    a = _outputs(B, H, h, N, T, t["keep_bytes"])
    _sublayer(kk, a, t, B, H, h, N, T)
    torch.cuda.synchronize()
    assert float(a["n"].float().abs().sum()) > 0
    kk.call("kk_chain_begin")
    kk.call("kk_chain_abort")
    _sublayer(kk, c, t, B, H, h, N, T)
    torch.cuda.synchronize()
    assert torch.equal(a["n"], c["n"])
