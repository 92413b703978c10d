"""GPU suite, engine level: the HIP train step (forward, 6 losses, all gradients, optimizer boundary) against
(1) the committed golden vectors — outputs of the reference itself — and (2) the CPU oracle on the same inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import kokoro_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from kokoro_ruslan_amd import engine
    return engine


def _load(golden_dir, name):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    d = O.ModelDims(*[int(x) for x in fx["dims"]])
    batch = {k.split("/", 1)[1]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("batch/")}
    seed = int(fx["seed"])
    P = O.init_params(d, seed)
    g = torch.Generator().manual_seed(seed + 1)
    for n, p in P.items():
        if p.dim() == 1:
            p.add_(torch.randn(p.shape, generator=g) * 0.1)
    return fx, d, batch, P


def _engine(eng_mod, d, P, math_mode="f32", storage="auto", **hp_kw):
    from kokoro_ruslan_amd.spec import ModelDims, StepHyper
    e = eng_mod.KokoroEngine(ModelDims(**d.__dict__), StepHyper(**hp_kw), math_mode=math_mode, init=False, total_steps=20000,
                             storage=storage)
    e.load_params(P)
    return e


def _cuda(batch):
    return {k: v.cuda() for k, v in batch.items()}


def _relerr(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


@pytest.mark.parametrize("name", ["tiny_full", "tiny_ragged", "mid_chunked", "full_dims"])
def test_train_step_parity_fp32(eng_mod, golden_dir, name):
    fx, d, batch, P = _load(golden_dir, name)
    e = _engine(eng_mod, d, P)
    e.zero_grad()
    out = e.forward_backward(_cuda(batch), loss_scale=1.0)
    torch.cuda.synchronize()
    # (1) against the reference's own outputs (golden)
    for k in ("mel", "log_dur", "stop", "pitch", "energy"):
        ref = torch.from_numpy(fx[f"out/{k}"])
        err = float((out[k].cpu() - ref).abs().max())
        assert err < 2e-4, f"{name}: output {k} max|err| {err:.3e}"
    losses = out["losses"].cpu().double().numpy()
    assert abs(losses[1] - fx["losses"][1]) < 1e-4, "mel-L1 must be within 1e-4 of the reference (north_star bar)"
    np.testing.assert_allclose(losses, fx["losses"], atol=1e-4, rtol=1e-5)
    names = list(O.param_shapes(d))
    G = e.grads()
    got_norms = np.array([float(G[n].double().norm()) for n in names])
    np.testing.assert_allclose(got_norms, fx["grad_norms"], rtol=2e-3, atol=1e-6)
    # (2) every gradient tensor against the oracle (autograd on the CPU restatement)
    Go, _, _ = O.grads_of(P, O.make_buffers(d), batch, d, O.StepHyper())
    worst = max(((_relerr(G[n], Go[n]) if float(Go[n].norm()) > 1e-7 else float(G[n].abs().max())), n) for n in names)
    assert worst[0] < 2e-3, f"{name}: worst gradient relative error {worst}"
    # length-regulator indices: bit-exact
    idx, lens, _ = O.length_regulate_index(batch["phoneme_durations"].numpy(), batch["mel_specs"].shape[1])
    assert np.array_equal(out["lr_idx"].cpu().numpy(), idx) and np.array_equal(out["lr_lens"].cpu().numpy(), lens)


def test_no_gradient_into_encoder_from_mel_loss(eng_mod, golden_dir):
    """with only the mel loss active the text encoder must receive exactly zero gradient."""
    fx, d, batch, P = _load(golden_dir, "tiny_ragged")
    e = _engine(eng_mod, d, P, duration_loss_weight=0.0, stop_token_loss_weight=0.0, pitch_loss_weight=0.0,
                energy_loss_weight=0.0)
    e.zero_grad()
    e.forward_backward(_cuda(batch))
    G = e.grads()
    assert float(G["text_embedding.weight"].abs().max()) == 0.0
    assert float(G["transformer_encoder_layers.0.self_attn.w_q.weight"].abs().max()) == 0.0
    assert float(G["decoder.layers.0.ff.linear1.weight"].abs().max()) > 0.0


def test_optimizer_boundary_matches_reference(eng_mod, golden_dir):
    """Pre-clip, clip, AdamW, EMA and weight-norm projection against the reference's post-step state."""
    fx, d, batch, P = _load(golden_dir, "tiny_ragged")
    names = list(O.param_shapes(d))
    wn = float(fx["step/max_weight_norm"])
    e = _engine(eng_mod, d, P, dec_ffn_max_weight_norm=wn, gradient_accumulation_steps=1)
    e.zero_grad()
    e.forward_backward(_cuda(batch), loss_scale=float(fx["step/grad_scale"]))     # grads x40 so every clip fires
    e.optimizer_step(batch["mel_specs"].shape[1])
    torch.cuda.synchronize()
    st = e.opt_stats()
    assert st["last_skip"] == 0.0 and st["attempt"] == 1.0
    assert abs(st["last_grad_norm"] - float(fx["step/grad_norm"])) < 2e-3 * float(fx["step/grad_norm"])
    assert abs(st["last_clip_coef"] - float(fx["step/clip_coef"])) < 2e-3 * float(fx["step/clip_coef"])
    assert abs(st["last_base_lr"] - 5e-5) < 1e-12          # step 0 runs at the full OneCycle initial LR (quirk kept)
    sd, esd = e.state_dict(), e.state_dict(ema=True)
    got = np.array([float(sd[n].double().norm()) for n in names])
    np.testing.assert_allclose(got, fx["step/param_norms"], rtol=1e-5, atol=1e-8)
    got = np.array([float((sd[n].cpu().double() - P[n].double()).norm()) for n in names])
    np.testing.assert_allclose(got, fx["step/delta_norms"], rtol=5e-3, atol=1e-8)
    for n in names:
        if f"step_param/{n}" in fx.files:
            ref = fx[f"step_param/{n}"]
            np.testing.assert_allclose(sd[n].cpu().numpy(), ref, atol=1e-6 + 1e-5 * np.abs(ref).max(), rtol=0)
            ref = fx[f"step_ema/{n}"]      # EMA moves by (1-0.9999)*delta: compare values, the delta is below fp32 ulp
            np.testing.assert_allclose(esd[n].cpu().numpy(), ref, atol=2e-7 + 2e-7 * np.abs(ref).max(), rtol=0)
    # buffers are EMA-tracked like the reference's state_dict loop and must stay put
    assert torch.equal(esd["positional_encoding.pe"].cpu(), O.make_buffers(d)["positional_encoding.pe"])
    # and against the oracle, tensor by tensor
    Go, _, _ = O.grads_of(P, O.make_buffers(d), batch, d, O.StepHyper(), loss_scale=float(fx["step/grad_scale"]))
    P2 = {n: P[n].clone() for n in names}
    ema = {n: P[n].clone() for n in names}
    hp = O.StepHyper(dec_ffn_max_weight_norm=wn)
    O.optimizer_step(P2, Go, O.OptState(), hp, hp.learning_rate, hp.max_grad_norm, ema, None)
    for n in names:
        dref = (P2[n] - P[n]).double()
        dgot = (sd[n].cpu() - P[n]).double()
        assert float((dgot - dref).norm()) <= 5e-3 * float(dref.norm()) + 1e-9, n
        assert float((esd[n].cpu() - ema[n]).abs().max()) <= 2e-7 + 2e-7 * float(ema[n].abs().max()), ("ema", n)


def test_multi_step_training_tracks_oracle(eng_mod, golden_dir):
    """Three optimizer steps with gradient accumulation 2 (6 micro-batches): parameters track the oracle."""
    fx, d, _, P = _load(golden_dir, "tiny_full")
    names = list(O.param_shapes(d))
    hp = O.StepHyper(warmup_steps=2, learning_rate=1e-3)
    e = _engine(eng_mod, d, P, warmup_steps=2, learning_rate=1e-3, gradient_accumulation_steps=2)
    e.total_steps = 10
    Bf = O.make_buffers(d)
    Po = {n: P[n].clone() for n in names}
    ema = {n: P[n].clone() for n in names}
    st, sch = O.OptState(), O.LRSchedule(hp, 10)
    for step in range(3):
        acc = {n: torch.zeros_like(P[n]) for n in names}
        for mb in range(2):
            batch = O.synthetic_batch(2, 40, 6, d, seed=100 + step * 2 + mb, ragged=True)
            e.train_step(_cuda(batch))
            G, _, _ = O.grads_of(Po, Bf, batch, d, hp, loss_scale=0.5)
            for n in names:
                acc[n] += G[n]
        O.optimizer_step(Po, acc, st, hp, sch.base_lr(step), hp.max_grad_norm, ema, None)
    torch.cuda.synchronize()
    sd = e.state_dict()
    assert e.opt_stats()["attempt"] == 3.0
    for n in names:
        dref = (Po[n] - P[n]).double()
        dgot = (sd[n].cpu() - P[n]).double()
        assert float((dgot - dref).norm()) <= 2e-2 * float(dref.norm()) + 1e-8, n


def test_nonfinite_gradients_skip_the_step(eng_mod, golden_dir):
    fx, d, batch, P = _load(golden_dir, "tiny_full")
    e = _engine(eng_mod, d, P, gradient_accumulation_steps=1)
    e.zero_grad()
    e.forward_backward(_cuda(batch))
    e.arena.G["mel_projection_out.bias"][0] = float("nan")
    before = e.arena.p.clone()
    e.optimizer_step(40)
    torch.cuda.synchronize()
    st = e.opt_stats()
    assert st["last_skip"] == 1.0 and st["skipped"] == 1.0
    assert torch.equal(e.arena.p, before), "a skipped step must leave every parameter untouched"
    # the device records which tensor caused the skip (first segment with a non-finite norm, how many, at which boundary)
    assert e.arena.names[int(st["bad_seg"]) - 1] == "mel_projection_out.bias" and st["bad_count"] == 1.0 and st["bad_attempt"] == 0.0


def test_ema_update_every_counts_successful_steps(eng_mod, golden_dir):
    """ema_update_every = N (config.py:87, trainer.py:1499-1502): _update_ema runs after every SUCCESSFUL optimizer step and the EMA
    weights move when the count of such steps so far is a multiple of N — steps 0, N, 2N, ...; a skipped (non-finite) boundary does
    not count.  Decided on the device (kk_opt_prepare -> step_consts[0] = 2), so a replayed optimizer graph follows it.  The first
    move against the oracle's optimizer_step, every move against decay * ema + (1 - decay) * p of that step."""
    fx, d, batch, P = _load(golden_dir, "tiny_full")
    names = list(O.param_shapes(d))
    e = _engine(eng_mod, d, P, gradient_accumulation_steps=1, ema_update_every=3, ema_decay=0.9)
    b = _cuda(batch)
    moved = []
    for attempt in range(6):                                # attempt 2 is poisoned: successful steps are attempts 0, 1, 3, 4, 5
        e.zero_grad()
        e.forward_backward(b)
        if attempt == 2:
            e.arena.G["mel_projection_out.bias"][0] = float("nan")
        before = e.arena.ema.clone()
        e.optimizer_step(40)
        torch.cuda.synchronize()
        moved.append(not torch.equal(e.arena.ema, before))
        if moved[-1]:
            want = 0.9 * before.double() + 0.1 * e.arena.p.double()
            assert float((e.arena.ema.double() - want).abs().max()) <= 1e-6 * (1.0 + float(want.abs().max())), attempt
    assert moved == [True, False, False, False, True, False]          # successful steps 0 and 3 (= attempt 4)
    st = e.opt_stats()
    assert st["skipped"] == 1.0 and st["attempt"] == 6.0
    # the oracle through the same rule: two successful steps, one EMA move
    hp = O.StepHyper(ema_update_every=3, ema_decay=0.9)
    Go, _, _ = O.grads_of(P, O.make_buffers(d), batch, d, O.StepHyper())
    P2, ema2, st2 = {n: P[n].clone() for n in names}, {n: P[n].clone() for n in names}, O.OptState()
    O.optimizer_step(P2, {n: g.clone() for n, g in Go.items()}, st2, hp, hp.learning_rate, hp.max_grad_norm, ema2, None)
    after_first = {n: v.clone() for n, v in ema2.items()}
    O.optimizer_step(P2, {n: g.clone() for n, g in Go.items()}, st2, hp, hp.learning_rate, hp.max_grad_norm, ema2, None)
    assert all(torch.equal(ema2[n], after_first[n]) for n in names) and any(not torch.equal(after_first[n], P[n]) for n in names)
    e1 = _engine(eng_mod, d, P, gradient_accumulation_steps=1, ema_update_every=3, ema_decay=0.9)
    e1.zero_grad()
    e1.forward_backward(b)
    e1.optimizer_step(40)
    esd = e1.state_dict(ema=True)
    for n in names:
        assert float((esd[n].cpu() - after_first[n]).abs().max()) <= 2e-6 + 2e-6 * float(after_first[n].abs().max()), ("ema", n)


@pytest.mark.parametrize("epoch", [0, 2, 3, 7])
def test_legacy_cosine_restart_schedule(eng_mod, golden_dir, epoch):
    """use_onecycle_lr = False (trainer.py:789-799): every segment steps at eta_min + (learning_rate x its group multiplier - eta_min) x
    the epoch's cosine-restart factor (tests/test_dropin.py pins the factor against torch's scheduler) — no warm-up, no OneCycle.  One
    optimizer step per epoch value against the oracle's optimizer_step under the same rule; T_0 = 3: epoch 3 is a restart."""
    fx, d, batch, P = _load(golden_dir, "tiny_full")
    names = list(O.param_shapes(d))
    kw = dict(use_onecycle_lr=False, lr_T_0=3, lr_T_mult=2, lr_eta_min=1e-6)
    e = _engine(eng_mod, d, P, gradient_accumulation_steps=1, **kw)
    e.lr_epoch = epoch
    e.zero_grad()
    e.forward_backward(_cuda(batch))
    e.optimizer_step(40)
    torch.cuda.synchronize()
    hp = O.StepHyper(**kw)
    st = e.opt_stats()
    assert abs(st["last_base_lr"] - O.legacy_group_lr(hp, 1.0, epoch)) <= 1e-12
    if epoch in (0, 3):
        assert abs(st["last_base_lr"] - hp.learning_rate) <= 1e-15          # start of a period: the initial lr
    Go, _, _ = O.grads_of(P, O.make_buffers(d), batch, d, O.StepHyper())
    P2 = {n: P[n].clone() for n in names}
    O.optimizer_step(P2, Go, O.OptState(), hp, hp.learning_rate, hp.max_grad_norm, None, None, lr_of=lambda m: O.legacy_group_lr(hp, m, epoch))
    sd = e.state_dict()
    for n in names:
        dref, dgot = (P2[n] - P[n]).double(), (sd[n].cpu() - P[n]).double()
        assert float((dgot - dref).norm()) <= 5e-3 * float(dref.norm()) + 1e-9, n


@pytest.mark.parametrize("storage", ["f32", "bf16-dec", "bf16"])
def test_bf16_math_mode_close_to_fp32(eng_mod, golden_dir, storage):
    """bf16 MFMA arithmetic, with fp32 or bf16 operand storage: losses and gradient directions stay on the reference."""
    fx, d, batch, P = _load(golden_dir, "mid_chunked")
    e = _engine(eng_mod, d, P, math_mode="bf16", storage=storage)
    e.zero_grad()
    out = e.forward_backward(_cuda(batch))
    torch.cuda.synchronize()
    losses = out["losses"].cpu().double().numpy()
    np.testing.assert_allclose(losses, fx["losses"], atol=3e-2, rtol=3e-2)
    names = list(O.param_shapes(d))
    Go, _, _ = O.grads_of(P, O.make_buffers(d), batch, d, O.StepHyper())
    G = e.grads()
    cos = []
    for n in names:
        a, b = G[n].cpu().double().flatten(), Go[n].double().flatten()
        if float(b.norm()) > 1e-6:
            cos.append(float(a @ b / (a.norm() * b.norm() + 1e-30)))
    assert min(cos) > 0.97 and float(np.mean(cos)) > 0.995, (min(cos), float(np.mean(cos)))


def test_bf16_weight_shadow_tracks_master(eng_mod, golden_dir):
    """The bf16 copy of the weights (the GEMMs' B operand in the bf16 mode) is exactly round(master) after loads,
    optimizer steps (incl. the weight-norm projection) and skipped steps; fp32_math() validates on the masters."""
    fx, d, batch, P = _load(golden_dir, "tiny_full")
    e = _engine(eng_mod, d, P, math_mode="bf16", dec_ffn_max_weight_norm=6.0, gradient_accumulation_steps=1)
    assert e.arena.p16 is not None and torch.equal(e.arena.p16, e.arena.p.bfloat16())
    b = _cuda(batch)
    for _ in range(3):
        e.train_step(b)
    torch.cuda.synchronize()
    assert e.opt_stats()["attempt"] == 3
    assert torch.equal(e.arena.p16, e.arena.p.bfloat16())
    ref = _engine(eng_mod, d, {n: e.arena.P[n].clone() for n in P})          # fp32 engine on the same weights
    with e.fp32_math():
        l16 = e.forward_backward(b, backward=False)["losses"].clone()
    l32 = ref.forward_backward(b, backward=False)["losses"]
    torch.testing.assert_close(l16, l32, rtol=1e-5, atol=1e-6)     # same kernels, same masters (split-K atomics reorder sums)
    with pytest.raises(ValueError):
        _engine(eng_mod, d, P, math_mode="f32", storage="bf16")


def test_state_dict_roundtrip_and_names(eng_mod, golden_dir):
    fx, d, batch, P = _load(golden_dir, "tiny_full")
    e = _engine(eng_mod, d, P)
    sd = e.state_dict()
    assert list(sd.keys()) == O.state_dict_order(d)
    for n in P:
        assert torch.equal(sd[n].cpu(), P[n])
    e2 = _engine(eng_mod, d, O.init_params(d, 99))
    e2.load_state_dict({k: v.clone() for k, v in sd.items()})
    assert torch.equal(e2.arena.p, e.arena.p)
    with pytest.raises(RuntimeError):
        e2.load_state_dict({"bogus": torch.zeros(1)})


def test_training_dropout_is_deterministic_per_seed_and_gradients_are_consistent(eng_mod, golden_dir):
    """With dropout / DropPath / SpecAugment on: same seed ⇒ identical losses and gradients, a new seed changes them,
    and the analytic gradient agrees with a central finite difference taken under the SAME masks — i.e. the backward
    kernels regenerate the forward masks at every site."""
    fx, d, batch, P = _load(golden_dir, "tiny_ragged")
    names = list(O.param_shapes(d))

    def run(seed, params=None, backward=True):
        e = run.e
        if params is not None:
            e.load_params(params, reset_ema=False)
        e.rng.fill_(seed)
        e.zero_grad()
        out = e.forward_backward(_cuda(batch), backward=backward)
        torch.cuda.synchronize()
        return float(out["losses"][0]), {n: g.clone() for n, g in e.grads().items()}
    run.e = _engine(eng_mod, d, P)
    run.e.train_dropout = True
    l1, g1 = run(100)
    l2, g2 = run(100)
    l3, g3 = run(101)
    assert abs(l1 - l2) <= 2e-6 * abs(l1)                # identical masks => identical forward (up to the order of fp32 atomics)
    for n in names:                                      # gradient sums use fp32 atomics: equal up to summation order
        assert float((g1[n] - g2[n]).abs().max()) <= 1e-5 + 1e-4 * float(g1[n].abs().max()), n
    assert l1 != l3
    run.e.train_dropout = False
    l0, _ = run(100)
    assert abs(l1 - l0) > 1e-4, "dropout must change the loss"
    run.e.train_dropout = True
    # directional derivative along a random direction restricted to a few tensors of every kind
    gen = torch.Generator().manual_seed(0)
    # (decoder-side and predictor tensors only: a finite difference on ENCODER weights also moves the detached
    #  length-regulated memory, which the analytic gradient deliberately excludes — SURVEY §0 fact 5)
    pick = [n for n in names if any(k in n for k in ("decoder.layers.0.ff.linear1.weight", "decoder.layers.1.cross_attn.w_k.weight",
            "decoder.layers.0.self_attn.w_q.weight", "pitch_predictor.conv_layers.0.weight", "mel_projection_in.weight",
            "decoder.layers.0.norm2.weight", "pitch_embedding.weight", "decoder.layers.1.ff.linear2.weight"))]
    report = []
    for n in pick:
        Vn = torch.randn(P[n].shape, generator=gen)
        analytic = float((g1[n].cpu().double() * Vn.double()).sum())
        eps = 1e-3
        lp, _ = run(100, {**P, n: P[n] + eps * Vn}, backward=False)    # same seed argument => the masks of g1
        lm, _ = run(100, {**P, n: P[n] - eps * Vn}, backward=False)
        report.append((n, analytic, (lp - lm) / (2 * eps)))
    bad = [(n, a_, f_) for n, a_, f_ in report if abs(f_ - a_) > 0.1 * abs(a_) + 0.02]
    assert not bad, report


def test_global_loss_normalisers_two_shards_on_one_gpu(eng_mod, golden_dir):
    """Data parallel with ragged shards, emulated on one GPU: each shard's backward uses the GLOBAL loss sums/counts
    (engine.loss_sync + kk_losses_finalize), gradients are summed — the result is the global-batch gradient."""
    fx, d, _, P = _load(golden_dir, "tiny_full")
    glob = O.synthetic_batch(4, 40, 6, d, seed=3)
    g = torch.Generator().manual_seed(9)
    glob["mel_lengths"] = torch.tensor([40, 17, 33, 25])
    glob["phoneme_lengths"] = torch.tensor([6, 3, 5, 4])
    shards = [{k: v[i:i + 2].contiguous() for k, v in glob.items()} for i in (0, 2)]
    e = _engine(eng_mod, d, P)
    accs, mds = [], []
    e.self_cleaning_acc = False                         # (the accumulator is normally left zero by its last reader: keep it to look at it)
    for sh in shards:                                   # what the ranks' loss forward kernels accumulate
        e.forward_backward(_cuda(sh), backward=False)
        accs.append(e.loss_acc.clone())
        mds.append(e.max_dur.clone())
    e.self_cleaning_acc = True
    e.loss_acc.zero_()
    assert not torch.equal(accs[0][5:], accs[1][5:]), "shards must have different valid counts"
    g_acc, g_md = accs[0] + accs[1], torch.maximum(mds[0], mds[1])

    class FakeAllReduce:                               # stands in for the loss exchange over 2 ranks
        def __init__(self, capturable):
            self.capturable = capturable

        def loss_sync(self, acc, md):
            acc.copy_(g_acc)
            md.copy_(g_md)
    fake_all_reduce = FakeAllReduce(False)
    e.loss_sync, e.dp_loss_scale = fake_all_reduce, 1.0
    e.zero_grad()
    for sh in shards:
        out = e.forward_backward(_cuda(sh))
    summed = e.arena.g.clone()
    global_losses = out["losses"].clone()               # re-finalised from the reduced accumulator = global-batch losses
    e.loss_sync = None
    e.zero_grad()
    ref_out = e.forward_backward(_cuda(glob))
    torch.cuda.synchronize()
    torch.testing.assert_close(global_losses, ref_out["losses"], rtol=2e-5, atol=1e-6)
    ref = e.arena.g
    assert float((summed - ref).abs().max()) <= 3e-5 * max(1.0, float(ref.abs().max()))
    with pytest.raises(RuntimeError, match="hipGraph cannot hold"):      # a torch.distributed exchange is eager only ...
        e.loss_sync = fake_all_reduce
        e.train_step_graphed(_cuda(glob))
    # ... one whose collectives go through the step's own communicator is captured with the step: replay == eager
    e2 = _engine(eng_mod, d, P, gradient_accumulation_steps=1)
    e2.loss_sync, e2.dp_loss_scale = FakeAllReduce(True), 1.0
    sh = _cuda(shards[0])
    for _ in range(3):                                  # eager, capture, replay
        e2.train_step_graphed(sh)
    e3 = _engine(eng_mod, d, P, gradient_accumulation_steps=1)
    e3.loss_sync, e3.dp_loss_scale = FakeAllReduce(False), 1.0
    for _ in range(3):
        e3.train_step(sh)
    torch.cuda.synchronize()
    assert e2.opt_stats()["attempt"] == e3.opt_stats()["attempt"] == 3
    assert float((e2.arena.p - e3.arena.p).abs().max()) <= 2e-6 * float(e3.arena.p.abs().max())


def test_bf16_mode_odd_ragged_shapes(eng_mod, golden_dir):
    """bf16 storage + DMA GEMM core + grouped attention on shapes that are multiples of nothing (B 3, T 437, P 53, ragged
    lengths): finite, and the gradients point where the fp32 parity mode's do."""
    fx, d, _, P = _load(golden_dir, "mid_chunked")
    batch = O.synthetic_batch(3, 437, 53, d, seed=8)
    batch["mel_lengths"] = torch.tensor([437, 301, 399])
    batch["phoneme_lengths"] = torch.tensor([53, 37, 50])
    ref = _engine(eng_mod, d, P)
    ref.zero_grad()
    lo = ref.forward_backward(_cuda(batch))["losses"].clone()
    e = _engine(eng_mod, d, P, math_mode="bf16")
    assert e.storage == "bf16"
    e.zero_grad()
    lb = e.forward_backward(_cuda(batch))["losses"]
    torch.cuda.synchronize()
    assert bool(torch.isfinite(lb).all())
    np.testing.assert_allclose(lb.cpu().numpy(), lo.cpu().numpy(), atol=3e-2, rtol=3e-2)
    cos = []
    for n in O.param_shapes(d):
        a, b = e.arena.G[n].double().flatten(), ref.arena.G[n].double().flatten()
        assert bool(torch.isfinite(a).all()), n
        if float(b.norm()) > 1e-6:
            cos.append(float(a @ b / (a.norm() * b.norm() + 1e-30)))
    assert min(cos) > 0.97 and float(np.mean(cos)) > 0.995, (min(cos), float(np.mean(cos)))
    # and with every random mask on: finite, optimizer step not skipped
    e.train_dropout = True
    for _ in range(3):
        e.train_step(_cuda(batch))
    st = e.opt_stats()
    assert st["skipped"] == 0 and bool(torch.isfinite(e.arena.p).all())


# ---------------------------------------------------------------------------------------------------------------------
# round 2: expanded length, bounded workspace, the step driver's device branches, graph replay at the bench shape
# ---------------------------------------------------------------------------------------------------------------------
def test_expanded_length_parity(eng_mod, golden_dir):
    """T' = max_b sum(dur) > T (model.py:607-628) against the reference's own outputs: [B, T'] pitch / energy predictions,
    mel / losses / gradients on the first T frames; T' < T raises the reference's size-mismatch RuntimeError."""
    fx = np.load(os.path.join(golden_dir, "expanded_length.npz"))
    d = O.ModelDims(*[int(x) for x in fx["dims"]])
    seed = int(fx["seed"])
    P = O.init_params(d, seed)
    g = torch.Generator().manual_seed(seed + 1)
    for n, p in P.items():
        if p.dim() == 1:
            p.add_(torch.randn(p.shape, generator=g) * 0.1)
    e = _engine(eng_mod, d, P)
    names = list(O.param_shapes(d))
    for tag in ("longer", "longer_chunk"):
        batch = {k.split("/", 2)[2]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(f"{tag}/batch/")}
        T, Tp = batch["mel_specs"].shape[1], int(batch["phoneme_durations"].sum(1).max())
        e.zero_grad()
        out = e.forward_backward(_cuda(batch), expanded_len=Tp)
        torch.cuda.synchronize()
        assert tuple(out["pitch"].shape) == (batch["mel_specs"].shape[0], Tp)
        for k in ("mel", "log_dur", "stop", "pitch", "energy"):
            ref = torch.from_numpy(fx[f"{tag}/out/{k}"])
            err = float((out[k].cpu() - ref).abs().max())
            assert err < 2e-4, f"{tag}: output {k} max|err| {err:.3e}"
        np.testing.assert_allclose(out["losses"].cpu().double().numpy(), fx[f"{tag}/losses"], atol=1e-4, rtol=1e-5)
        got = np.array([float(e.arena.G[n].double().norm()) for n in names])
        np.testing.assert_allclose(got, fx[f"{tag}/grad_norms"], rtol=2e-3, atol=1e-6)
        with pytest.raises(RuntimeError, match="must match the size of tensor"):
            e.forward_backward(_cuda(batch), expanded_len=T - 2)
    # without the hint the engine assumes T' = T; the same batch with consistent durations runs either way identically
    b2 = O.synthetic_batch(2, 48, 7, d, seed=5, ragged=True)
    l0 = e.forward_backward(_cuda(b2), backward=False)["losses"].clone()
    l1 = e.forward_backward(_cuda(b2), backward=False, expanded_len=48)["losses"].clone()
    torch.testing.assert_close(l0, l1, rtol=1e-5, atol=1e-6)          # (split-K fp32 atomics reorder a few sums)


def test_workspace_is_bounded_by_the_largest_shape(eng_mod, golden_dir):
    """Dynamic batching shows the engine a new (B, T, P) nearly every step: the activation workspace must be sized by the
    largest batch, not by the number of shapes, and a shape must compute the same whether it came first or fiftieth."""
    fx, d, _, P = _load(golden_dir, "tiny_full")
    e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
    rs = np.random.RandomState(0)
    shapes = [(4, 160, 24)] + [(int(rs.randint(1, 5)), int(rs.randint(48, 161)), int(rs.randint(4, 25))) for _ in range(50)]
    torch.cuda.synchronize()
    sizes, allocs = [], []
    for i, (B, T, Pn) in enumerate(shapes):
        b = _cuda(O.synthetic_batch(B, T, Pn, d, seed=200 + i, ragged=True))
        e.train_step(b)
        sizes.append(e.workspace_bytes())
        allocs.append(torch.cuda.memory_allocated())
    torch.cuda.synchronize()
    assert e.opt_stats()["skipped"] == 0
    assert sizes[-1] <= sizes[0] * 1.30, (sizes[0], sizes[-1])           # later, smaller shapes add (almost) nothing
    assert allocs[-1] <= allocs[0] * 1.30 + (8 << 20), (allocs[0], allocs[-1])
    assert len(e._tables) <= e.max_tables
    # a shape seen late gives the losses and gradients of a fresh engine
    probe = O.synthetic_batch(3, 77, 11, d, seed=9, ragged=True)
    e2 = _engine(eng_mod, d, {n: e.arena.P[n].clone() for n in P}, math_mode="bf16", gradient_accumulation_steps=1)
    for x in (e, e2):
        x.zero_grad()
        x.forward_backward(_cuda(probe))
    torch.cuda.synchronize()
    torch.testing.assert_close(e.losses, e2.losses, rtol=1e-6, atol=1e-7)
    assert float((e.arena.g - e2.arena.g).abs().max()) <= 1e-4 * float(e2.arena.g.abs().max()) + 1e-7


def test_device_step_driver_against_reference_sequences(eng_mod, golden_dir):
    """kk_opt_prepare over scripted gradient norms: explosion thresholds / emergency clip / EMA / streak against
    step_driver.npz (produced by running the reference's statements), and the device LR schedule through warm-up AND the
    cosine phase against the reference's scheduler sequence lr_seq_60_20.npy."""
    from kokoro_ruslan_amd import lib as kk, spec
    fx = np.load(os.path.join(golden_dir, "step_driver.npz"))
    d = O.ModelDims(vocab=59, mel=20, hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=96, dec_ff=96, var_filter=32,
                    var_kernel=3, var_bins=16, max_len=300)
    P = O.init_params(d, 1)

    def drive(e, norms, check):
        a = e.arena
        seg = a.names.index("decoder.norm.weight")           # a segment without a pre-clip ceiling
        for k, nv in enumerate(norms):
            e.grad_sumsq.zero_()
            e.grad_sumsq[seg] = float(nv) ** 2
            cfg = e._opt_cfg(64)
            kk.call("kk_opt_prepare", e.grad_sumsq, a.seg_preclip, a.seg_lr_mult, a.seg_wd, a.nseg, e.max_dur, cfg, e.opt_state,
                    e.seg_gscale, e.seg_decay, e.seg_stepsize, e.step_consts, None, None)
            check(k, e.opt_stats(), e)
    # ---- explosion tracker: 520 steps ----
    e = _engine(eng_mod, d, P, gradient_accumulation_steps=1)
    e.max_dur.zero_()
    rows = fx["explosion"]

    def chk_expl(k, st, eng):
        norm, thr, clip, expl, ema, streak = rows[k]
        assert abs(st["last_grad_norm"] - norm) <= 1e-9 * norm
        assert abs(st["last_clip_norm"] - clip) < 1e-12, (k, st["last_clip_norm"], clip)
        assert abs(st["expl_ema"] - ema) <= 1e-9 * abs(ema) and st["expl_streak"] == streak, (k, st, rows[k])
        assert abs(st["last_clip_coef"] - min(1.0, clip / (norm + 1e-6))) < 1e-9
    drive(e, rows[:, 0], chk_expl)
    assert e.opt_stats()["attempt"] == len(rows) and e.opt_stats()["skipped"] == 0
    # ---- LR schedule: 20 warm-up steps + 40 one-cycle steps + 3 beyond the end ----
    seq = np.load(os.path.join(golden_dir, "lr_seq_60_20.npy"))          # [63, 10] per-group LR the reference used at step k
    e = _engine(eng_mod, d, P, gradient_accumulation_steps=1, warmup_steps=20)
    e.total_steps = 60
    e.max_dur.zero_()
    table = spec.group_lr_mult_wd(e.hp)
    segs = {gi: next(i for i, n in enumerate(e.arena.names) if n in e.arena.param_names and spec.param_group_of(n) == gi) for gi in range(10)}

    def chk_lr(k, st, eng):
        for gi in range(10):
            want = float(seq[k, gi])
            assert abs(st["last_base_lr"] * table[gi][0] - want) <= 1e-12 + 1e-9 * want, (k, gi)
        bc1 = 1.0 - eng.hp.adam_betas[0] ** (k + 1)
        got = eng.seg_stepsize.cpu().double().numpy()
        for gi, si in segs.items():
            assert abs(got[si] - seq[k, gi] / bc1) <= 2e-7 * seq[k, gi] / bc1 + 1e-15, (k, gi)
    drive(e, np.full(len(seq), 1.0), chk_lr)
    lrs = seq[:, 2]
    assert lrs[27] > lrs[40] > lrs[58] > lrs[59] > 0 and lrs[60] == lrs[61] == lrs[62]     # the cosine phase ran down to its floor


def test_adaptive_loss_scale_and_clip_fire_on_long_batches(eng_mod, golden_dir):
    """trainer.py:2218-2242 on the device: a 1500-frame batch with a 200-frame phoneme scales the loss by 0.75 and enters
    the optimizer boundary with clip norm 0.433 (values from running the reference's statements: step_driver.npz)."""
    fx = np.load(os.path.join(golden_dir, "step_driver.npz"))
    row = next(r for r in fx["adaptive"] if int(r[0]) == 1500 and int(r[1]) == 200)
    scale, clip = float(row[2]), float(row[3])
    assert scale < 1.0 and clip < 1.5
    d = O.ModelDims(vocab=59, mel=20, hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=96, dec_ff=96, var_filter=32,
                    var_kernel=3, var_bins=16, max_len=1600)
    P = O.init_params(d, 2)
    batch = O.synthetic_batch(1, 1500, 30, d, seed=1)
    dur = torch.full((1, 30), 44, dtype=torch.long)          # 200 + 29 x 44 + 24 = 1500 frames, one 200-frame phoneme
    dur[0, 0] = 200
    dur[0, 1:25] += 1
    batch["phoneme_durations"] = dur
    assert int(dur.sum()) == 1500 and int(dur.max()) == 200
    e = _engine(eng_mod, d, P, gradient_accumulation_steps=1)
    b = _cuda(batch)
    e.zero_grad()
    e.forward_backward(b, adaptive=False)
    g0 = e.arena.g.clone()
    e.zero_grad()
    out = e.forward_backward(b, adaptive=True)
    torch.cuda.synchronize()
    ratio = float((e.arena.g.double() * g0.double()).sum() / (g0.double() ** 2).sum())
    assert abs(ratio - scale) < 1e-4, (ratio, scale)
    e.optimizer_step(1500)
    st = e.opt_stats()
    assert abs(st["last_clip_norm"] - clip) < 1e-9, (st["last_clip_norm"], clip)
    # the oracle agrees on the scaled gradients (loss scale folded into the backward seed, losses reported unscaled)
    Go, ls, _ = O.grads_of(P, O.make_buffers(d), batch, d, O.StepHyper(), loss_scale=scale)
    np.testing.assert_allclose(out["losses"].cpu().numpy(), [float(x) for x in ls], atol=1e-4, rtol=1e-5)


def test_graphed_accumulation_matches_eager(eng_mod, golden_dir):
    """train_step_graphed with the reference's default accumulation (G = 2): micro-batch graphs (first / second of the
    cycle) + the boundary graph give the parameters of the eager train_step sequence, masks included."""
    fx, d, _, P = _load(golden_dir, "tiny_full")
    batches = [_cuda(O.synthetic_batch(2, 40, 6, d, seed=300 + i, ragged=True)) for i in range(2)]
    res = []
    for graphed in (False, True):
        e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=2)
        e.train_dropout = True
        for it in range(8):                                  # 4 optimizer steps; the graphs replay from the third cycle on
            b = batches[it % 2]
            (e.train_step_graphed if graphed else e.train_step)(b)
        torch.cuda.synchronize()
        st = e.opt_stats()
        assert st["attempt"] == 4 and st["skipped"] == 0 and e.micro_in_cycle == 0
        res.append(e.arena.p.clone())
        if graphed:
            assert len(e._graphs) == 1 and len(next(iter(e._graphs.values()))["fb"]) == 2
    assert float((res[0] - res[1]).abs().max()) <= 2e-5 * float(res[0].abs().max())
    # the auto path: eager on first sight, graphs once a shape repeats
    e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=2)
    e.train_dropout = True
    for it in range(8):
        e.train_step_auto(batches[it % 2])
    torch.cuda.synchronize()
    assert float((e.arena.p - res[0]).abs().max()) <= 2e-5 * float(res[0].abs().max())
    assert len(e._graphs) == 1


@pytest.mark.parametrize("B,T,Pn", [(8, 512, 64), (8, 1024, 128)])
def test_bench_config_bf16_graph_replay_tracks_fp32_and_oracle(eng_mod, B, T, Pn):
    """The configurations bench.py times — 8 x 512 frames x 64 phonemes (configs[1], the headline) and 8 x 1024 x 128 (configs[3]'s
    per-GPU shape), default 49.4 M-parameter model, bf16, hipGraph replay — checked for what they compute: (1) losses against the
    CPU oracle and the fp32 engine (mel-L1 within 1e-4), (2) every gradient's direction against the fp32 engine, (3) with all
    dropout on, a replayed step is the eager step (same seed, same masks), (4) 12 replayed optimizer steps: no skips, finite,
    the loss moves."""
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = O.ModelDims()
    P = O.init_params(d, 0)
    cpu = synthetic_batch(B, T, Pn, seed=1234)
    b = _cuda(cpu)
    f32 = _engine(eng_mod, d, P, gradient_accumulation_steps=1)
    f32.zero_grad()
    l32 = f32.forward_backward(b)["losses"].clone()
    with torch.no_grad():
        lo = torch.stack([x.float() for x in O.losses(O.forward(P, O.make_buffers(d), cpu, d), cpu, O.StepHyper())])
    torch.testing.assert_close(l32.cpu(), lo, rtol=2e-4, atol=2e-4)                  # mel-L1 within 1e-4 is asserted below
    assert abs(float(l32[1]) - float(lo[1])) < 1e-4
    e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
    e.zero_grad()
    l16 = e.forward_backward(b)["losses"].clone()
    torch.cuda.synchronize()
    np.testing.assert_allclose(l16.cpu().numpy(), lo.numpy(), rtol=3e-2, atol=3e-2)
    cos, rel = [], []
    for n in O.param_shapes(d):
        a, r = e.arena.G[n].double().flatten(), f32.arena.G[n].double().flatten()
        if float(r.norm()) > 1e-7:
            cos.append(float(a @ r / (a.norm() * r.norm() + 1e-30)))
            rel.append(abs(float(a.norm()) / float(r.norm()) - 1.0))
    assert min(cos) > 0.95 and float(np.mean(cos)) > 0.99, (min(cos), float(np.mean(cos)))
    assert float(np.median(rel)) < 0.02, float(np.median(rel))
    gn32 = float(f32.arena.g.double().norm())
    assert abs(float(e.arena.g.double().norm()) / gn32 - 1.0) < 0.02
    del f32
    # (3) dropout on: eager step vs. graph replay from the same state and seed
    e.train_dropout = True
    e2 = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
    e2.train_dropout = True
    for _ in range(3):                                       # eager, capture, replay
        e2.train_step_graphed(b)
    for _ in range(3):
        e.train_step(b)
    torch.cuda.synchronize()
    assert int(e.rng.item()) == int(e2.rng.item())
    # (the embedding / bucket-table gradients are fp32 atomic scatter-adds: their summation order, and with it the last bits
    #  of those gradients, varies from run to run in a few discrete modes whatever the launch form — tools/probes/
    #  enc_stack_eager_vs_graph.py shows two eager runs differing exactly like an eager and a replayed one; after three
    #  steps through bf16 roundings that is up to ~4e-4 on a loss)
    torch.testing.assert_close(e2.losses, e.losses, rtol=2e-3, atol=1e-4)
    assert float((e.arena.p - e2.arena.p).abs().max()) <= 5e-5 * float(e.arena.p.abs().max())
    # (4) keep replaying
    first = e2.losses.clone()
    for _ in range(12):
        e2.train_step_graphed(b)
    torch.cuda.synchronize()
    st = e2.opt_stats()
    assert st["attempt"] == 15 and st["skipped"] == 0
    assert bool(torch.isfinite(e2.losses).all()) and bool(torch.isfinite(e2.arena.p).all())
    assert float(e2.losses[0]) != float(first[0])


def test_bench_config_takes_the_pair_launch_and_it_changes_nothing(eng_mod):
    """At the bench shape every attention backward — the decoder's 12 and, since round 3, the text encoder's six (one 64-row tile:
    wave group 1 of the second-generation kernels idles) — is kk_gemm_dgrad_delta + ONE kk_attn_bwd launch, and the step computes
    what the two-launch form computes: same losses, gradients equal to the rounding of Delta's summation order.  With
    attn_pair_min_seq = 64 (the round-2 dispatch) the encoder keeps two launches."""
    from kokoro_ruslan_amd import lib as kk
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = O.ModelDims()
    P = O.init_params(d, 0)
    b = _cuda(synthetic_batch(8, 512, 64, seed=1234))
    e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
    assert e.attn_bwd_pair and e.attn_pair_min_seq == 64               # (round 5: the engine's default is the round-2 dispatch again —
    e.attn_pair_min_seq = 32                                           #  measured inside the step; the pair launch of one-tile sequences stays tested)
    e.zero_grad()
    kk.profile_start()
    l_pair = e.forward_backward(b)["losses"].clone()
    names = [r[0] for r in kk.profile_stop()]
    g_pair = e.arena.g.clone()
    n_attn = 2 * d.dec_layers + d.enc_layers
    assert names.count("kk_attn_bwd") == n_attn and names.count("kk_gemm_dgrad_delta") == n_attn
    assert "kk_attn_bwd_dkv" not in names and "kk_attn_bwd_dq" not in names
    e.attn_pair_min_seq = 64                                           # the round-2 dispatch: the encoder keeps two launches
    e.zero_grad()
    kk.profile_start()
    e.forward_backward(b)
    names64 = [r[0] for r in kk.profile_stop()]
    assert names64.count("kk_attn_bwd") == 2 * d.dec_layers and names64.count("kk_attn_bwd_dkv") == d.enc_layers
    e.attn_pair_min_seq = 32
    e.attn_bwd_pair = False
    e.zero_grad()
    kk.profile_start()
    l_two = e.forward_backward(b)["losses"].clone()
    names = [r[0] for r in kk.profile_stop()]
    assert names.count("kk_attn_bwd") == 0 and names.count("kk_attn_bwd_dq") == d.enc_layers + 2 * d.dec_layers
    torch.cuda.synchronize()
    assert torch.equal(l_pair, l_two)                                  # (dropout off: the forward is deterministic)
    a, r = g_pair.double(), e.arena.g.double()
    assert float((a - r).norm() / r.norm()) < 2e-3 and float(a @ r / (a.norm() * r.norm())) > 0.99999


def test_memory_tail_on_the_third_stream_changes_nothing(eng_mod):
    """tail_aside: the backward's memory tail (cross-attention K/V weight gradients of all layers, the memory gradient, the two
    bucket-embedding gradients) forked from layer 0's cross-attention backward (1) or from the end of the decoder backward (2) onto
    the third stream computes what the serial order computes — eager and in a replayed graph (the fork and the join are graph edges)."""
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = O.ModelDims()
    P = O.init_params(d, 0)
    b = _cuda(synthetic_batch(8, 512, 64, seed=77))
    ref_g = ref_l = None
    for mode in (0, 1, 2, 3, 4):
        e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
        e.train_dropout = True                                         # the input projection's two dropout sites and SpecAugment are on the tail
        e.tail_aside = mode
        e.zero_grad()
        l = e.forward_backward(b)["losses"].clone()
        g = e.arena.g.clone()
        named = {n: e.arena.G[n].double().flatten().clone() for n in
                 ("decoder.layers.0.cross_attn.w_k.weight", "duration_adaptor.variance_adaptor.pitch_embedding.weight", "mel_projection_in.weight")}
        torch.cuda.synchronize()
        assert bool(torch.isfinite(g).all())
        if mode == 0:
            ref_g, ref_l, ref_named = g.double(), l, named
            continue
        assert torch.equal(l, ref_l)
        # (split-K partial sums land by fp32 atomics: run-to-run order noise only)
        assert float((g.double() - ref_g).norm() / ref_g.norm()) < 1e-5
        for n, a in named.items():
            r = ref_named[n]
            assert float(r.norm()) > 0 and float((a - r).norm() / r.norm()) < 1e-5, n


def test_step_zero_fill_skips_only_what_the_step_overwrites(eng_mod):
    """forward_backward(zero_grads=True) clears the gradient arena minus the tensors the step's grouped weight-gradient launches
    overwrite (recorded from a previous step of the same precision mode).  Poisoned with NaN before a step, every element of the
    arena must come out finite — zeroed or overwritten — in the replayed graph, with accumulation, and after a mode change; and
    the gradients are those of a full zero-fill."""
    from kokoro_ruslan_amd import lib as kk
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = O.ModelDims()
    P = O.init_params(d, 0)
    b = _cuda(synthetic_batch(4, 300, 40, seed=5))
    e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
    assert e.zero_skip_overwritten
    e.forward_backward(b, zero_grads=True)                                  # first step of this mode: full fill, records the set
    rec = e._ow_sets[e._ow_key()]
    skipped = sum(n for _, n in rec)
    assert 0.8 < skipped / e.arena.total < 0.95, skipped / e.arena.total     # the weight matrices: 89 % of the arena
    g_full = e.arena.g.clone()
    e.arena.g.fill_(float("nan"))
    kk.profile_start()
    e.forward_backward(b, zero_grads=True)
    names = [r[0] for r in kk.profile_stop()]
    assert names.count("kk_zero_many") == 1
    torch.cuda.synchronize()
    assert bool(torch.isfinite(e.arena.g).all()), "an element was neither zeroed nor overwritten"
    a, r = e.arena.g.double(), g_full.double()
    assert float((a - r).norm() / r.norm()) < 2e-3                          # (two runs: scatter-add order)
    # the fp32 parity mode groups nothing: its own (empty) record, full fill
    with e.fp32_math():
        e.arena.g.fill_(float("nan"))
        e.forward_backward(b, zero_grads=True)
        assert len(e._ow_sets[e._ow_key()]) == 0
        assert bool(torch.isfinite(e.arena.g).all())
    # graph replay with accumulation: the cycle's first micro-batch fills and overwrites, the second accumulates
    e2 = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=2)
    for _ in range(6):
        e2.train_step_graphed(b)
    e2.arena.g.fill_(float("nan"))
    for _ in range(4):
        e2.train_step_graphed(b)
    torch.cuda.synchronize()
    st = e2.opt_stats()
    assert st["skipped"] == 0 and st["attempt"] == 5 and bool(torch.isfinite(e2.arena.g).all()) and bool(torch.isfinite(e2.arena.p).all())
    # a stale record is an error, not a silent stale gradient
    key = e._ow_key()
    e._ow_sets[key] = frozenset(list(e._ow_sets[key])[:-1] + [(e.arena.G["decoder.norm.weight"].data_ptr(), 512)])
    e._tables.clear()
    with pytest.raises(RuntimeError, match="stale overwrite record"):
        e.forward_backward(b, zero_grads=True)


def test_micro_batch_finite_guard_drops_the_cycle(eng_mod, golden_dir):
    """Reference guards per micro-batch (trainer.py:3233-3296, 2304-2314): an infinite mel-projection bias makes one
    output column infinite while every loss stays finite (non-finite elements are masked out of the means) and every
    gradient stays finite — only the finite-output guard can see it.  The flagged micro-batch must poison its
    accumulation cycle: no optimizer step, parameters untouched; the next cycle trains normally."""
    fx, d, _, P = _load(golden_dir, "tiny_full")
    good = [O.synthetic_batch(2, 40, 6, d, seed=500 + i, ragged=True) for i in range(4)]
    Pbad = {n: p.clone() for n, p in P.items()}
    Pbad["mel_projection_out.bias"][3] = float("inf")
    # the oracle's restatement of the guard agrees on which micro-batches are flagged
    with torch.no_grad():
        out = O.forward(Pbad, O.make_buffers(d), good[1], d)
        ls = O.losses(out, good[1], O.StepHyper())
        assert all(bool(torch.isfinite(x)) for x in ls) and not O.micro_batch_ok(out, ls)
        out = O.forward(P, O.make_buffers(d), good[1], d)
        assert O.micro_batch_ok(out, O.losses(out, good[1], O.StepHyper()))
    e = _engine(eng_mod, d, P, gradient_accumulation_steps=2)
    e.train_step(_cuda(good[0]))                              # cycle 1, micro-batch 1: fine
    live = e.arena.P["mel_projection_out.bias"]
    live[3] = float("inf")                                    # cycle 1, micro-batch 2: infinite output column
    before = e.arena.p.clone()
    losses = e.train_step(_cuda(good[1])).clone()
    torch.cuda.synchronize()
    st = e.opt_stats()
    assert bool(torch.isfinite(losses).all()), "the losses mask non-finite elements out, like the reference's"
    assert bool(torch.isfinite(e.arena.g).all()), "this case is invisible to the gradient check"
    assert st["attempt"] == 1 and st["skipped"] == 1 and st["last_skip"] == 1 and st["micro_bad_total"] == 1 and st["micro_bad"] == 0
    assert torch.equal(e.arena.p, before), "a dropped cycle must leave every parameter untouched"
    live[3] = P["mel_projection_out.bias"][3]                 # healthy again: the next cycle steps
    e.train_step(_cuda(good[2]))
    e.train_step(_cuda(good[3]))
    torch.cuda.synchronize()
    st = e.opt_stats()
    assert st["attempt"] == 2 and st["skipped"] == 1 and st["last_skip"] == 0 and not torch.equal(e.arena.p, before)
    # forward-only calls (validation) never touch the training cycle's flag
    live[3] = float("inf")
    e.forward_backward(_cuda(good[0]), backward=False)
    torch.cuda.synchronize()
    assert e.opt_stats()["micro_bad"] == 0 and e.opt_stats()["micro_bad_total"] == 1


def test_in_graph_bucket_exchange_over_rccl_one_rank(eng_mod, golden_dir):
    """The data-parallel exchange as it runs on 8 GPUs, on one: a 1-rank RCCL communicator through the C ABI (kk_comm_*),
    the buckets issued from inside the backward in reverse-autograd order, captured into the step's hipGraph.  With one
    rank the sum is the identity, so training must be exactly the training without the exchange — eager, replayed, with
    gradient accumulation (only the boundary micro-batch exchanges) and with the bf16 payload (identity up to bf16
    rounding of the gradients)."""
    from kokoro_ruslan_amd import dp, lib as kk
    fx, d, _, P = _load(golden_dir, "tiny_full")
    batches = [_cuda(O.synthetic_batch(2, 40, 6, d, seed=700 + i, ragged=True)) for i in range(2)]

    def run(comm, graphed, G):
        e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=G)
        e.train_dropout = True
        e.dp_comm = comm
        log = []
        if comm is not None:
            orig = comm.reduce_tags                          # (one exchange per GROUP of buckets: dp.BucketedExchange.groups)
            comm.reduce_tags = lambda flat, tags: (log.extend(tags), orig(flat, tags))[1]
        for it in range(4 * G):
            (e.train_step_graphed if graphed else e.train_step)(batches[it % 2])
        torch.cuda.synchronize()
        if comm is not None:
            comm.reduce_tags = orig
        assert e.opt_stats()["attempt"] == 4 and e.opt_stats()["skipped"] == 0
        return e.arena.p.clone(), log
    from kokoro_ruslan_amd.spec import ModelDims
    dims = ModelDims(**d.__dict__)
    comm = dp.BucketedExchange.create(dims, 0, 1, torch.device("cuda"))
    assert comm.backend == "rccl" and kk.load().kk_comm_world() == 1, "RCCL must bind through the C ABI on the GPU box"
    tags = list(comm.plan)
    ref1, _ = run(None, False, 1)
    for graphed in (False, True):
        got, log = run(comm, graphed, 1)
        assert float((got - ref1).abs().max()) <= 2e-5 * float(ref1.abs().max())
        n_calls = 4 if not graphed else 2                     # replays do not go through Python: eager first sight + one capture
        assert log == tags * n_calls, (graphed, log[:8])
    ref2, _ = run(None, False, 2)
    got, log = run(comm, True, 2)
    assert float((got - ref2).abs().max()) <= 2e-5 * float(ref2.abs().max())
    assert log and len(log) % len(tags) == 0 and log[:len(tags)] == tags
    comm.payload = "bf16"
    got, _ = run(comm, True, 1)
    assert float((got - ref1).abs().max()) <= 2e-3 * float(ref1.abs().max())
    assert float((got - ref1).abs().max()) > 0.0, "the bf16 payload must actually round the gradients"
    # the raw entry points
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    kk.call("kk_comm_reduce_bucket", x, 1000, 0)
    y = torch.empty(1000, dtype=torch.float32, device="cuda")
    kk.call("kk_comm_reduce_scatter", x, y, 1000, 0)
    z = torch.empty(1000, dtype=torch.float32, device="cuda")
    kk.call("kk_comm_all_gather", y, z, 1000, 0)
    torch.cuda.synchronize()
    assert torch.equal(x, torch.arange(1000, dtype=torch.float32, device="cuda")) and torch.equal(z, x)
    with pytest.raises(RuntimeError, match="dtype"):
        kk.call("kk_comm_reduce_bucket", x, 1000, 7)
    # the step's second collective through the same communicator: the loss normalisers of ragged shards (kk_comm_loss_sync: fp64 SUM
    # + int64 MAX as one RCCL group on the step's stream).  One rank: identity — so a graph-replayed step with the exchange AND the
    # global normalisers on (what kokoro-train runs on 8 GPUs) must train exactly like the plain engine.
    acc = torch.arange(12, dtype=torch.float64, device="cuda") * 1.5
    md = torch.tensor([41], dtype=torch.int64, device="cuda")
    comm.payload = "f32"
    assert comm.capturable
    comm.loss_sync(acc, md)
    torch.cuda.synchronize()
    assert torch.equal(acc, torch.arange(12, dtype=torch.float64, device="cuda") * 1.5) and int(md) == 41

    def run_ragged(graphed):
        e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
        e.train_dropout = True
        e.dp_comm, e.loss_sync, e.dp_loss_scale = comm, comm, 1.0
        for it in range(4):
            (e.train_step_auto if graphed else e.train_step)(batches[it % 2])
        torch.cuda.synchronize()
        assert e.opt_stats()["attempt"] == 4 and e.opt_stats()["skipped"] == 0
        if graphed:
            assert len(e._graphs) == 1 and any(k[7] for ent in e._graphs.values() for k in ent["fb"]), "the ragged step must replay from a graph (key field 7: a loss exchange is captured)"
        return e.arena.p.clone()
    got_e, got_g = run_ragged(False), run_ragged(True)
    assert float((got_e - ref1).abs().max()) <= 2e-5 * float(ref1.abs().max())
    assert float((got_g - ref1).abs().max()) <= 2e-5 * float(ref1.abs().max())

    # kk_losses_finalize takes the GLOBAL mel length by value (the adaptive loss scale above 1400 frames), so a graph captured
    # under one value must not be replayed under another: one local shape, two global lengths, replayed = eager; and the length matters
    def run_global_t(graphed, lengths):
        e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
        e.train_dropout = True
        e.dp_comm, e.loss_sync, e.dp_loss_scale = comm, comm, 1.0
        for g_t in lengths:
            e.global_mel_length = g_t
            (e.train_step_auto if graphed else e.train_step)(batches[0])
        torch.cuda.synchronize()
        return e.arena.p.clone(), e.opt_stats()["last_grad_norm"]
    # (AdamW normalises the gradient scale away, so the parameters barely see the loss scale; the gradient norm of the last step does:
    #  scale = 1400 / T above 1400 frames.  The sequence ends on a length other than the one its graph was first captured under.)
    seq = (1500, 1500, 2000, 2000, 1500)
    (p_e, n_e), (p_g, n_g) = run_global_t(False, seq), run_global_t(True, seq)
    assert float((p_g - p_e).abs().max()) <= 2e-5 * float(p_e.abs().max())
    assert abs(n_g / n_e - 1.0) < 2e-3, f"a replayed graph must not keep the global mel length of its capture: |g| {n_g} (replayed) vs {n_e} (eager)"
    _, n_c = run_global_t(True, (2000,) * len(seq))
    assert abs(n_c / n_e - 1.0) > 0.1, "the test must be sensitive to the global mel length"
    # global lengths up to 1400 frames share ONE capture per local shape (canonical_mel_length) — a ragged data-parallel run
    # replays instead of re-capturing — and give the eager result
    short = (900, 1000, 1100, 1399, 700, 1400)
    e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
    e.train_dropout = True
    e.dp_comm, e.loss_sync, e.dp_loss_scale = comm, comm, 1.0
    for g_t in short:
        e.global_mel_length = g_t
        e.train_step_auto(batches[0])
    torch.cuda.synchronize()
    assert sum(len(ent["fb"]) for ent in e._graphs.values()) == 1 and sum(len(ent["opt"]) for ent in e._graphs.values()) == 1, \
        "one forward/backward graph and one optimizer graph for every global mel length <= 1400"
    p_s, _ = run_global_t(False, short)
    assert float((e.arena.p - p_s).abs().max()) <= 2e-5 * float(p_s.abs().max())


# ---------------------------------------------------------------------------------------------------------------------
# round 3: the bf16 mode against the reference's OWN mixed precision, and a long bf16 trajectory
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fixture", ["autocast_bf16", "autocast_bf16_t512"])
def test_bf16_mode_against_the_references_own_autocast(eng_mod, golden_dir, fixture):
    """tests/golden/autocast_bf16*.npz = the reference run twice on one batch and set of weights: fp32, and its model under
    torch.autocast(bfloat16) (what `use_mixed_precision` does, trainer.py:3181-3232) — losses of both, the L1 distance between the
    two mel outputs, and per-tensor cosine / norm ratio between the two sets of gradients.  The engine's bf16 mode (the mode
    bench.py times) must stay as close to the reference's fp32 numbers as the reference's own bf16 mode does (x1.5 slack: two
    different roundings of the same computation are not closer to each other than each is to the truth).  Two batches: the
    `full_dims` one (2 x 96 frames) and one at the bench's frame count (2 x 512 frames x 64 phonemes, ragged: the decoder's
    attention over 512 keys, the predictors' 512-frame GroupNorm chunk at full length)."""
    ac = np.load(os.path.join(golden_dir, fixture + ".npz"))
    if fixture == "autocast_bf16":
        fx, d, batch, P = _load(golden_dir, "full_dims")
        for k, v in batch.items():
            assert np.array_equal(v.numpy(), ac[f"batch/{k}"]), "the autocast fixture was made on the full_dims batch"
        np.testing.assert_allclose(ac["losses_fp32"], fx["losses"], rtol=1e-6, atol=1e-7)       # same reference fp32 run
    else:
        _, d, batch, P = _load(golden_dir, fixture)                 # dims, batch and the seed of the weights travel in the fixture
    names = list(O.param_shapes(d))
    ref32 = _engine(eng_mod, d, P)                                  # pinned to the reference's fp32 numbers (test_train_step_parity_fp32)
    ref32.zero_grad()
    o32 = ref32.forward_backward(_cuda(batch))
    # the fp32 engine against the REFERENCE's fp32 losses of this batch (total, mel, dur, stop, pitch, energy): mel-L1 within 1e-4
    np.testing.assert_allclose(o32["losses"].cpu().double().numpy(), ac["losses_fp32"], rtol=2e-4, atol=2e-4)
    assert abs(float(o32["losses"][1]) - float(ac["losses_fp32"][1])) < 1e-4
    e = _engine(eng_mod, d, P, math_mode="bf16")
    assert e.storage == "bf16"
    e.zero_grad()
    o16 = e.forward_backward(_cuda(batch))
    torch.cuda.synchronize()
    l16, l32 = o16["losses"].cpu().double().numpy(), ac["losses_fp32"]
    d_ref = np.abs(ac["losses_autocast"] - ac["losses_fp32"])       # how far the reference's autocast moves each loss
    d_eng = np.abs(l16 - l32)
    assert (d_eng <= 1.5 * d_ref + 2e-3).all(), (d_eng, d_ref)
    B, T = batch["mel_specs"].shape[:2]
    valid = (torch.arange(T)[None, :] < batch["mel_lengths"][:, None])[:, :, None].float()
    mel_l1 = float(((o16["mel"].cpu() - torch.from_numpy(ac["mel_fp32"])).abs() * valid).sum() / (valid.sum() * d.mel))
    mel_l1_cross = (float(((o16["mel"].cpu() - torch.from_numpy(ac["mel_autocast"])).abs() * valid).sum() / (valid.sum() * d.mel))
                    if "mel_autocast" in ac.files else float("nan"))
    print(f"mel-L1 of the bf16 engine's mel output: vs reference fp32 {mel_l1:.3e}, vs reference autocast {mel_l1_cross:.3e}; "
          f"reference autocast vs reference fp32 {float(ac['mel_l1_between']):.3e}; mel-loss delta {d_eng[1]:.3e} (reference's own: {d_ref[1]:.3e})")
    assert mel_l1 <= 1.5 * float(ac["mel_l1_between"]), (mel_l1, float(ac["mel_l1_between"]))
    G16, G32 = e.grads(), ref32.grads()
    worse = []
    for i, n in enumerate(names):
        a, b = G16[n].double().flatten(), G32[n].double().flatten()
        if float(b.norm()) < 1e-7:
            continue
        cos = float(a @ b / (a.norm() * b.norm() + 1e-300))
        ratio = float(a.norm() / b.norm())
        # the reference's own bf16 gradient of this tensor against its fp32 gradient: 1 - cos and |ratio - 1| are the yardsticks
        if (1 - cos) > 1.5 * (1 - float(ac["grad_cos"][i])) + 2e-3 or abs(ratio - 1) > 1.5 * abs(float(ac["grad_norm_ratio"][i]) - 1) + 3e-2:
            worse.append((n, round(cos, 5), round(float(ac["grad_cos"][i]), 5), round(ratio, 4), round(float(ac["grad_norm_ratio"][i]), 4)))
    assert not worse, f"{len(worse)} gradients further from fp32 than the reference's autocast puts them: {worse[:8]}"


def test_bf16_trajectory_200_steps_tracks_fp32(eng_mod):
    """200 optimizer steps at hidden 512 (default dims, B 4 x T 96): the bf16 mode's loss curve against the fp32 parity mode's from the
    same weights, batches and (dropout-free) configuration.  A sign error that bf16 rounding hides in one step, or a biased
    rounding somewhere in the optimizer / shadow-weight path, separates the curves over hundreds of steps."""
    d = O.ModelDims()
    P = O.init_params(d, 3)
    batches = [_cuda(O.synthetic_batch(4, 96, 12, d, seed=50 + i, ragged=True)) for i in range(8)]
    curves = {}
    for mode in ("f32", "bf16"):
        e = _engine(eng_mod, d, P, math_mode=mode, gradient_accumulation_steps=1, learning_rate=3e-4, warmup_steps=20)
        rec = []
        for s in range(200):
            rec.append(e.train_step_graphed(batches[s % 8]).clone())
        torch.cuda.synchronize()
        st = e.opt_stats()
        # (the reference's explosion tracker may legitimately drop a spiking step; fp32 atomics make runs differ in the last bits)
        assert st["attempt"] == 200 and st["skipped"] <= 2, (mode, st)
        curves[mode] = torch.stack(rec).cpu().double().numpy()
    c32, c16 = curves["f32"], curves["bf16"]
    assert np.isfinite(c16).all()
    sm = lambda c: np.convolve(c, np.ones(8) / 8, mode="valid")              # one pass over the 8 batches
    t32, t16 = sm(c32[:, 0]), sm(c16[:, 0])
    assert t32[-1] < 0.85 * t32[0], "the fp32 run must actually train in 200 steps"
    rel = np.abs(t16 - t32) / t32
    print(f"total loss {t32[0]:.4f} -> {t32[-1]:.4f} (fp32), {t16[0]:.4f} -> {t16[-1]:.4f} (bf16); worst relative gap of the smoothed curves {rel.max():.4f}, "
          f"final {rel[-1]:.4f}; mel-L1 final {sm(c32[:, 1])[-1]:.4f} / {sm(c16[:, 1])[-1]:.4f}")
    # observed over repeated runs: worst gap 0.7-2.2 %, final 0.2-1.3 % (run-to-run differences come from fp32 atomics in a few reductions)
    assert rel.max() < 0.06 and rel[-1] < 0.04, (rel.max(), rel[-1], t32[::24].round(4).tolist(), t16[::24].round(4).tolist())
    m32, m16 = sm(c32[:, 1]), sm(c16[:, 1])
    assert (np.abs(m16 - m32) / m32).max() < 0.06, (m32[::24].round(4).tolist(), m16[::24].round(4).tolist())


# ---- round 6: an ENGINE-level numeric check at configs[2]-shaped batches, default dims ------------------------
def _ragged_lengths(B, T, Pn, seed):
    """Valid lengths as a frame-budget batch has them (data/dataset.py:1007-1127 packs utterances of similar length: the longest defines
    the padded shape, the others are 55-100 % of it)."""
    g = torch.Generator().manual_seed(seed)
    mel = (T * (0.55 + 0.45 * torch.rand(B, generator=g))).long().clamp(min=8)
    ph = (Pn * (0.55 + 0.45 * torch.rand(B, generator=g))).long().clamp(min=4)
    mel[0], ph[0] = T, Pn
    return mel, ph


@pytest.mark.parametrize("B,T,Pn", [(12, 1333, 110), (28, 579, 48)])
def test_dynamic_batching_shapes_at_default_dims(eng_mod, B, T, Pn, monkeypatch):
    """What dynamic batching (configs[2]: B * T <= 16384, B 4..32) hands the engine: B != 8, ragged valid lengths, T not a multiple of
    any tile, >= 6 K rows — the routes only such batches take (attn_fwd3_q128 on a ragged sequence, the 256 x 128 large-tile GEMM with a
    ragged M, the row-owner projection + tail launch, the memory tail on the side stream above 4096 rows, the per-kernel encoder for
    B > 8).  Every kernel on them is tested alone; this is the step as a whole: (1) the fp32 engine's losses against the CPU oracle,
    (2) the bf16 engine against the fp32 engine — losses and all 308 gradients by direction and size, as at the bench shapes,
    (3) the routes ASSERTED from the record of the step's launches."""
    from kokoro_ruslan_amd import lib as kk
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = O.ModelDims()
    P = O.init_params(d, 0)
    cpu = synthetic_batch(B, T, Pn, seed=77, lengths=_ragged_lengths(B, T, Pn, 9))
    b = _cuda(cpu)
    f32 = _engine(eng_mod, d, P, gradient_accumulation_steps=1)
    f32.zero_grad()
    l32 = f32.forward_backward(b)["losses"].clone()
    with torch.no_grad():
        lo = torch.stack([x.float() for x in O.losses(O.forward(P, O.make_buffers(d), cpu, d), cpu, O.StepHyper())])
    torch.testing.assert_close(l32.cpu(), lo, rtol=2e-4, atol=2e-4)
    assert abs(float(l32[1]) - float(lo[1])) < 1e-4, "mel-L1 within 1e-4 of the reference restatement"
    e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
    e.zero_grad()
    routes = []
    real_call = kk.call

    def recording_call(name, *args):
        real_call(name, *args)
        routes.append((name, kk.last_kernel()))
    monkeypatch.setattr(kk, "call", recording_call)
    l16 = e.forward_backward(b)["losses"].clone()
    monkeypatch.setattr(kk, "call", real_call)
    torch.cuda.synchronize()
    np.testing.assert_allclose(l16.cpu().numpy(), lo.numpy(), rtol=3e-2, atol=3e-2)
    cos, rel = [], []
    for n in O.param_shapes(d):
        a, r = e.arena.G[n].double().flatten(), f32.arena.G[n].double().flatten()
        assert bool(torch.isfinite(a).all()), n
        if float(r.norm()) > 1e-7:
            cos.append(float(a @ r / (a.norm() * r.norm() + 1e-30)))
            rel.append(abs(float(a.norm()) / float(r.norm()) - 1.0))
    assert len(cos) >= 300
    assert min(cos) > 0.95 and float(np.mean(cos)) > 0.99, (min(cos), float(np.mean(cos)))
    assert float(np.median(rel)) < 0.02, float(np.median(rel))
    assert abs(float(e.arena.g.double().norm()) / float(f32.arena.g.double().norm()) - 1.0) < 0.02
    # (3) routes
    taken = {k for _, k in routes}
    names = [n for n, _ in routes]
    assert B * T > 6000 and B * T <= 16384
    assert "attn_fwd3_q128" in taken, sorted(taken)
    assert any(k.startswith("g16x<0,0,256,128,") or k.startswith("g16x<0,1,256,128,") for k in taken), sorted(taken)
    assert "linear_tail_fwd" in taken and "kk_linear_tail_fwd" in names
    assert "attn_bwd_pair3" in taken                     # (p = 0 here: the hashing pair launch; the keep-bit one is asserted below)
    assert ("kk_encoder_stack_fwd" in names) == (B <= 8)
    # the memory tail (cross K/V weight gradient + memory gradient + bucket-embedding gradients) ran on the side stream, behind the
    # text encoder's backward (tail_aside mode 3 above 4096 rows): its launches come after the encoder's embedding backward
    assert names.index("kk_bucket_embed_add_bwd_sorted") > names.index("kk_embed_bwd")
    # the training configuration (all dropout on): the attention forward stores its keep bits for ragged sequences too and the
    # backward's pair launch reads them
    e.train_dropout = True
    e.zero_grad()
    routes.clear()
    monkeypatch.setattr(kk, "call", recording_call)
    ld = e.forward_backward(b)["losses"].clone()
    monkeypatch.setattr(kk, "call", real_call)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ld).all()) and bool(torch.isfinite(e.arena.g).all())
    assert "attn_bwd_pair3k" in {k for _, k in routes} and "kk_attn_fwd_kb" in [n for n, _ in routes]
    # a second, replayed pass through the same shape computes the same step (eager -> capture -> replay)
    e.train_dropout = False
    p0 = e.arena.p.clone()
    for _ in range(3):
        e.train_step_graphed(b)
    torch.cuda.synchronize()
    assert e.opt_stats()["skipped"] == 0 and bool(torch.isfinite(e.arena.p).all()) and not torch.equal(p0, e.arena.p)


def test_accumulators_survive_an_aborted_step(eng_mod, golden_dir):
    """the handed-round accumulators (loss sums, per-segment norms) rely on their last reader leaving them zero.  A step
    that dies between writer and cleaner — here: the loss exchange raising after kk_losses_fwd, and a poisoned optimizer accumulator —
    must cost the next step one zero-fill, not silently wrong losses / clip coefficients.  Also: flipping the switch on a live engine."""
    fx, d, batch, P = _load(golden_dir, "tiny_full")
    b = _cuda(batch)
    ref = _engine(eng_mod, d, P, gradient_accumulation_steps=1)
    ref.train_step(b)
    want_l, want_p = ref.losses.clone(), ref.arena.p.clone()
    ref.train_step(b)
    want_l2, want_p2 = ref.losses.clone(), ref.arena.p.clone()

    class Boom:
        capturable = False

        def loss_sync(self, acc, max_dur):
            raise RuntimeError("collective failed")
    e = _engine(eng_mod, d, P, gradient_accumulation_steps=1)
    e.loss_sync = Boom()
    with pytest.raises(RuntimeError, match="collective failed"):
        e.train_step(b)                                   # kk_losses_fwd has added its sums; nobody cleared them
    torch.cuda.synchronize()
    assert not e._acc_clean and float(e.loss_acc.abs().sum()) > 0
    e.loss_sync, e.micro_in_cycle = None, 0
    e.p_sumsq.fill_(123456789)                            # (as if the optimizer had died between kk_adamw_ema and its reader too)
    e.train_step(b)
    torch.testing.assert_close(e.losses, want_l, rtol=1e-6, atol=1e-7)
    # (the embedding-table gradients are fp32 atomic scatter-adds: two runs differ in their last bits, see the bench-shape test)
    torch.testing.assert_close(e.arena.p, want_p, rtol=0, atol=3e-6)
    assert e._acc_clean
    e.self_cleaning_acc = False                           # the A/B switch flipped on a live engine, then back
    assert not e._acc_clean
    e.self_cleaning_acc = True
    e.train_step_graphed(b)
    e2 = _engine(eng_mod, d, P, gradient_accumulation_steps=1)
    e2.train_step(b)
    e2.self_cleaning_acc = False
    e2.train_step(b)
    torch.cuda.synchronize()
    torch.testing.assert_close(e.losses, want_l2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(e2.losses, want_l2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(e2.arena.p, want_p2, rtol=0, atol=6e-6)


def test_gradient_norms_from_the_weight_gradient_epilogues(eng_mod):
    """Round 6 (one GPU): the per-segment norms of the weight matrices come from the grouped weight-gradient launches' tile records and the
    optimizer's norm pass skips those tensors.  (1) Every per-segment sum of squares equals the full pass over the arena to fp64 rounding of
    fp32 partials, and most of the arena is covered; (2) steps with the switch on and off are the same steps; (3) gradient accumulation: the
    records of the boundary micro-batch describe the ACCUMULATED gradient; (4) a caller that hands its own gradients to optimizer_step gets
    the full pass."""
    from kokoro_ruslan_amd import lib as kk
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = O.ModelDims()
    P = O.init_params(d, 0)
    b = _cuda(synthetic_batch(8, 512, 64, seed=5))
    e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
    e.train_dropout = True
    e.zero_grad()
    e.forward_backward(b)
    ready = e._ss_ready
    assert ready is not None and ready[1] > 0
    a = e.arena
    covered_elems = sum(int(np.prod(a.shapes[a.names[s]])) for s in ready[0])
    assert covered_elems > 0.8 * sum(int(np.prod(a.shapes[n])) for n in a.param_names), "the weight matrices are most of the arena"
    full = torch.zeros(a.nseg, dtype=torch.float64, device="cuda")
    kk.call("kk_seg_sumsq", a.g, a.block_seg, a.nblocks, full, a.nseg, torch.empty_like(e.sumsq_ws), None, 0)
    skip = torch.zeros(a.nseg, dtype=torch.int32)
    skip[sorted(ready[0])] = 1
    fused = torch.full((a.nseg,), float("nan"), dtype=torch.float64, device="cuda")
    kk.call("kk_seg_sumsq", a.g, a.block_seg, a.nblocks, fused, a.nseg, e.sumsq_ws, skip.cuda(), ready[1])
    torch.cuda.synchronize()
    rel = ((fused - full).abs() / full.clamp(min=1e-30)).max()
    assert bool(torch.isfinite(fused).all()) and float(rel) < 1e-6, float(rel)
    # (2) + (3): trajectories with and without, G = 1 and G = 2
    for G in (1, 2):
        outs = []
        for on in (True, False):
            e2 = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=G)
            e2.train_dropout = True
            e2.grad_norm_from_wgrads = on
            for it in range(4):
                e2.train_step_graphed(b)
            torch.cuda.synchronize()
            st = e2.opt_stats()
            assert st["skipped"] == 0
            outs.append((e2.arena.p.clone(), st["last_grad_norm"], e2.losses.clone()))
        assert abs(outs[0][1] / outs[1][1] - 1.0) < 1e-5, (G, outs[0][1], outs[1][1])
        assert float((outs[0][0] - outs[1][0]).abs().max()) <= 5e-5 * float(outs[1][0].abs().max())
    # (4) gradients from outside: the full pass, whatever records an earlier micro-batch left
    e.arena.g.mul_(3.0)
    e.optimizer_step(1400)
    torch.cuda.synchronize()
    want = float((e.arena.g.double() ** 2).sum().sqrt())
    assert abs(e.opt_stats()["last_grad_norm"] / want - 1.0) < 0.05      # (pre-clip shrinks some segments; the records would have been 3x off)


@pytest.mark.parametrize("B,T,Pn,layers", [(8, 512, 64, 6), (8, 1024, 128, 4)])
def test_keep_bits_from_their_own_launch_in_the_step(eng_mod, monkeypatch, B, T, Pn, layers):
    """Round 6: with engine.attn_keep_gen the keep bits of the first `layers` decoder layers (what fits beside the persistent encoder: all six at 8 x 512,
    four at 8 x 1024) come from kk_attn_keep_gen and those layers' forwards READ them; the later layers hash and store.  Same bits either way: the losses of
    a dropout-on step are the losses of the step with the generator off, exactly, and the stored arrays are equal byte for byte."""
    from kokoro_ruslan_amd import lib as kk
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = O.ModelDims()
    P = O.init_params(d, 0)
    b = _cuda(synthetic_batch(B, T, Pn, seed=9))
    out = []
    for on in (True, False):
        e = _engine(eng_mod, d, P, math_mode="bf16", gradient_accumulation_steps=1)
        e.train_dropout = True
        e.attn_keep_gen = on
        e.zero_grad()
        routes = []
        real_call = kk.call

        def recording_call(name, *args):
            real_call(name, *args)
            routes.append((name, kk.last_kernel()))
        monkeypatch.setattr(kk, "call", recording_call)
        losses = e.forward_backward(b)["losses"].clone()
        monkeypatch.setattr(kk, "call", real_call)
        torch.cuda.synchronize()
        keeps = {k: v.clone() for k, v in e._ws.items() if k.endswith(".keep") and k.startswith("dec")}
        assert len(keeps) == 2 * d.dec_layers
        out.append((losses, routes, keeps))
    (l_on, r_on, k_on), (l_off, r_off, k_off) = out
    assert torch.equal(l_on, l_off), (l_on, l_off)
    names_on, names_off = [n for n, _ in r_on], [n for n, _ in r_off]
    assert names_on.count("kk_attn_keep_gen") == 1 and "kk_attn_keep_gen" not in names_off
    assert names_on.count("kk_attn_fwd_rb") == 2 * layers, names_on.count("kk_attn_fwd_rb")
    reading = {k for n, k in r_on if n == "kk_attn_fwd_rb"}
    assert reading <= {"attn_fwd3_q64r", "attn_fwd3_q128r"} and reading, reading
    assert "kk_attn_fwd_rb" not in names_off
    nU = (T + 31) // 32
    n = B * d.heads * nU * nU * 128
    assert n == kk.load().kk_attn_keep_bytes(B, d.heads, T, T)
    tril = torch.tril(torch.ones(nU, nU, dtype=torch.bool, device="cuda"))
    for k in k_on:                                   # (units above the diagonal of a causal launch are visited by nobody: compare what is read)
        a, c = k_on[k][:n].view(B * d.heads, nU, nU, 128), k_off[k][:n].view(B * d.heads, nU, nU, 128)
        if k.endswith(".sa.keep"):
            a, c = a[:, tril], c[:, tril]
        assert torch.equal(a, c), k
