"""GPU suite (round 6): keep bits from a launch of their own.  kk_attn_keep_gen must write exactly the bits the hashing
forward (kk_attn_fwd_kb) stores — same function, same layout — for every unit the forward visits, and kk_attn_fwd_rb (which READS them)
must give the hashing launch's output bit for bit; the backward's keep-bit pair launch then works from generated bits as from stored ones."""
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


def _qkv(B, h, S, Sk, seed):
    g = torch.Generator().manual_seed(seed)
    H = h * 64
    mk = lambda n: (torch.randn(B * n, H, generator=g) * 0.7).cuda().to(torch.bfloat16)
    return mk(S), mk(Sk), mk(Sk)


@pytest.mark.parametrize("B,h,S,Sk,causal,masked", [(8, 8, 512, 512, 1, 0), (8, 8, 512, 512, 0, 0), (8, 8, 1024, 1024, 1, 0),
                                                     (3, 8, 1333, 1333, 1, 0), (5, 8, 579, 579, 0, 1), (2, 8, 900, 1000, 0, 1)])
def test_generated_keep_bits_equal_the_forwards_and_the_reading_forward_is_bit_identical(kk, B, h, S, Sk, causal, masked):
    H, p, site = h * 64, 0.2, 2003
    q, k, v = _qkv(B, h, S, Sk, 5)
    seed = torch.tensor([4321], dtype=torch.int32, device="cuda")
    nbytes = kk.load().kk_attn_keep_bytes(B, h, S, Sk)
    assert nbytes > 0
    km = None
    if masked:
        km = torch.zeros(B, Sk, dtype=torch.uint8, device="cuda")
        for b in range(B):
            km[b, Sk - 17 * (b + 1):] = 1
    o0, l0 = torch.zeros(B * S, H, dtype=torch.bfloat16, device="cuda"), torch.zeros(B, h, S, device="cuda")
    stored = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device="cuda")
    kk.call("kk_attn_fwd_kb", q, k, v, o0, l0, B, h, S, Sk, H, H, H, H, km, causal, 0.125, seed, site, p, kk.KK_MATH_BF16, 1, stored)
    assert kk.last_kernel() in ("attn_fwd3_q128", "attn_fwd3_q64")
    gen = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device="cuda")        # same background: untouched units must be the same units
    sites = kk.keep_sites([(gen, site, p, B, h, S, Sk, causal)])
    kk.call("kk_attn_keep_gen", sites, 1, seed, 0, 0)
    assert kk.last_kernel() == "attn_keep_gen"
    torch.cuda.synchronize()
    nQU, nKU = (S + 31) // 32, (Sk + 31) // 32
    a, b_ = stored.view(B * h, nQU, nKU, 128), gen.view(B * h, nQU, nKU, 128)
    # every unit the forward wrote, the generator wrote identically; the generator never writes a unit above the causal diagonal
    wrote_f = (a != 0x5A).any(dim=3)
    wrote_g = (b_ != 0x5A).any(dim=3)
    assert bool((a[wrote_f] == b_[wrote_f]).all()), "generated bits differ from the forward's ballots"
    if causal:
        qu, ku = torch.meshgrid(torch.arange(nQU), torch.arange(nKU), indexing="ij")
        assert not bool(wrote_g[:, (ku > qu).cuda()].any())
    assert int(wrote_f.sum()) > 0 and bool((wrote_g | ~wrote_f).all()), "the generator covers every unit the forward visits"
    # the reading forward
    o1, l1 = torch.zeros_like(o0), torch.zeros_like(l0)
    kk.call("kk_attn_fwd_rb", q, k, v, o1, l1, B, h, S, Sk, H, H, H, H, km, causal, 0.125, seed, site, p, kk.KK_MATH_BF16, 1, gen)
    assert kk.last_kernel() in ("attn_fwd3_q128r", "attn_fwd3_q64r")
    torch.cuda.synchronize()
    assert torch.equal(o0, o1) and torch.equal(l0, l1), "kk_attn_fwd_rb must give the hashing forward's bits"
    # a second seed value: new bits, still equal
    seed.fill_(77)
    kk.call("kk_attn_fwd_kb", q, k, v, o0, l0, B, h, S, Sk, H, H, H, H, km, causal, 0.125, seed, site, p, kk.KK_MATH_BF16, 1, stored)
    kk.call("kk_attn_keep_gen", sites, 1, seed, 0, 0)
    kk.call("kk_attn_fwd_rb", q, k, v, o1, l1, B, h, S, Sk, H, H, H, H, km, causal, 0.125, seed, site, p, kk.KK_MATH_BF16, 1, gen)
    torch.cuda.synchronize()
    assert torch.equal(o0, o1) and torch.equal(l0, l1)


def test_keep_gen_many_sites_in_one_launch(kk):
    """Twelve sites (the decoder's 6 causal + 6 full attentions at 8 x 512) in ONE launch == twelve single-site launches."""
    B, h, S = 8, 8, 512
    seed = torch.tensor([99], dtype=torch.int32, device="cuda")
    nbytes = kk.load().kk_attn_keep_bytes(B, h, S, S)
    many = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(12)]
    one = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(12)]
    ent = lambda bufs: [(bufs[i], 2003 + 32 * (i // 2) + 8 * (i % 2), 0.2 if i % 2 == 0 else 0.15, B, h, S, S, i % 2 == 0) for i in range(12)]
    kk.call("kk_attn_keep_gen", kk.keep_sites(ent(many)), 12, seed, 0, 0)
    for e in ent(one):
        kk.call("kk_attn_keep_gen", kk.keep_sites([e]), 1, seed, 0, 0)
    torch.cuda.synchronize()
    for a, b in zip(many, one):
        assert torch.equal(a, b)
    assert not torch.equal(many[0], many[2]) and float(many[1].float().mean()) > 0
    with pytest.raises(RuntimeError, match="no keep-bit array"):
        kk.call("kk_attn_keep_gen", kk.keep_sites([(many[0], 1, 0.2, B, h, 64, 64, 0)]), 1, seed, 0, 0)
