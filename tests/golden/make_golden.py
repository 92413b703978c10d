#!/usr/bin/env python3
"""Pin the oracle against the imported reference and emit golden fixtures.

Runs ONLY in the build container (needs /root/reference).  Usage:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It (1) asserts ``oracle/kokoro_oracle.py`` ≡ reference for outputs, 6 losses, all
gradients, the param-group partition, the pre-clip classes, one full optimizer step
(pre-clip → clip → AdamW → EMA → weight-norm) and the LR sequence; (2) writes small
``.npz`` fixtures next to this file.  The fixtures are data only: inputs and the
reference's outputs.
"""
import os
import sys
import types
import json
from unittest import mock

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import numpy as np
import torch

# --- stubs for modules absent from this image -----------------------------
tb = types.ModuleType("torch.utils.tensorboard")


class _SW:  # no-op SummaryWriter
    def __init__(self, *a, **k): pass
    def __getattr__(self, n): return lambda *a, **k: None


tb.SummaryWriter = _SW
sys.modules["torch.utils.tensorboard"] = tb
for name in ("torchaudio", "torchaudio.transforms", "torchaudio.functional"):
    sys.modules[name] = mock.MagicMock()

from kokoro.model.model import KokoroModel                      # noqa: E402
from kokoro.training.losses import calculate_training_losses    # noqa: E402
from kokoro.training.config import TrainingConfig               # noqa: E402
from kokoro.training.trainer import KokoroTrainer               # noqa: E402
from kokoro.training.runtime_policies import RuntimeStepPolicy  # noqa: E402
from kokoro.utils.lengths import vectorized_expand_tokens       # noqa: E402
import logging                                                  # noqa: E402

from oracle import kokoro_oracle as O                           # noqa: E402

torch.manual_seed(0)
logging.disable(logging.CRITICAL)
WN_CEIL = 6.0


def ref_model(d: O.ModelDims) -> KokoroModel:
    m = KokoroModel(d.vocab, d.mel, d.hidden, n_encoder_layers=d.enc_layers, n_heads=d.heads,
                    encoder_ff_dim=d.enc_ff, encoder_dropout=0.0, decoder_dropout=0.0,
                    decoder_input_dropout=0.0, n_decoder_layers=d.dec_layers, decoder_ff_dim=d.dec_ff,
                    max_decoder_seq_len=d.max_len, variance_filter_size=d.var_filter,
                    variance_kernel_size=d.var_kernel, variance_dropout=0.0, n_variance_bins=d.var_bins,
                    pitch_min=0.0, pitch_max=1.0, energy_min=0.0, energy_max=1.0,
                    use_stochastic_depth=False, qk_norm=True, ffn_output_norm=True)
    m.train()
    return m


def make_trainer(model, cfg) -> KokoroTrainer:
    t = KokoroTrainer.__new__(KokoroTrainer)
    t.config = cfg
    t.model = model
    t.device = torch.device("cpu")
    t.device_type = "cpu"
    t.use_mixed_precision = False
    t.scaler = None
    t.mixed_precision_stats = {}
    return t


def ref_losses(model, cfg, out, batch):
    import torch.nn as nn
    return calculate_training_losses(
        device=torch.device("cpu"), config=cfg, model=model,
        criterion_mel=nn.L1Loss(reduction="none"),
        criterion_duration=nn.HuberLoss(reduction="none", delta=1.0),
        criterion_stop_token=nn.BCEWithLogitsLoss(reduction="none",
                                                  pos_weight=torch.tensor([cfg.stop_token_pos_weight])),
        criterion_pitch=nn.HuberLoss(reduction="none", delta=cfg.pitch_huber_delta),
        criterion_energy=nn.HuberLoss(reduction="none", delta=cfg.energy_huber_delta),
        average_by_duration=None, logger=logging.getLogger("x"),
        predicted_mel=out[0], predicted_log_durations=out[1], predicted_stop_logits=out[2],
        mel_specs=batch["mel_specs"], phoneme_durations=batch["phoneme_durations"],
        stop_token_targets=batch["stop_token_targets"], mel_lengths=batch["mel_lengths"],
        phoneme_lengths=batch["phoneme_lengths"], predicted_pitch=out[3], predicted_energy=out[4],
        pitch_targets=batch["pitches"], energy_targets=batch["energies"])


def run_ref(model, cfg, batch):
    out = model(batch["phoneme_indices"], batch["mel_specs"], batch["phoneme_durations"],
                batch["stop_token_targets"], pitch_targets=batch["pitches"],
                energy_targets=batch["energies"], stress_indices=batch["stress_indices"])
    ls = ref_losses(model, cfg, out, batch)
    return out, ls


def check(name, a, b, atol, rtol=0.0):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = float((a - b).abs().max()) if a.numel() else 0.0
    tol = atol + rtol * float(b.abs().max() if b.numel() else 0)
    status = "ok " if err <= tol else "FAIL"
    print(f"  [{status}] {name:48s} max|Δ|={err:.3e} (tol {tol:.1e})")
    assert err <= tol, name


def randomise(model, seed):
    """Push every parameter off its init so gains/biases are not 1/0 (tests scale/shift paths)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.1)


def seeded_params(d: O.ModelDims, seed: int):
    """RNG-free-on-disk weights: both sides regenerate them from the seed (CPU generator)."""
    P = O.init_params(d, seed)
    g = torch.Generator().manual_seed(seed + 1)
    for n, p in P.items():
        if p.dim() == 1:
            p.add_(torch.randn(p.shape, generator=g) * 0.1)
    return P


def section_model(tag: str, d: O.ModelDims, B, T, Pn, seed, ragged, save_step: bool, seeded: bool = True):
    print(f"== {tag}: dims={d} batch=({B},{T},{Pn}) ragged={ragged} seeded={seeded}")
    cfg = TrainingConfig()
    hp = O.StepHyper()
    model = ref_model(d)
    if seeded:
        missing, unexpected = model.load_state_dict(seeded_params(d, seed), strict=False)
        assert not unexpected and all("bins" in m or m.endswith(".pe") for m in missing), (missing, unexpected)
    else:
        randomise(model, seed)
    names = [n for n, _ in model.named_parameters()]
    assert names == list(O.param_shapes(d).keys()), "parameter name order differs"
    for n, p in model.named_parameters():
        assert tuple(p.shape) == O.param_shapes(d)[n], n
    assert list(model.state_dict().keys()) == O.state_dict_order(d), "state_dict order differs"
    Bf = O.make_buffers(d)
    for k, v in Bf.items():
        check(f"buffer {k}", v, model.state_dict()[k], 0.0)

    batch = O.synthetic_batch(B, T, Pn, d, seed=seed, ragged=ragged)
    if ragged:
        # make id 0 ("comma" quirk) appear inside a real sequence too
        batch["phoneme_indices"][0, 2] = 0
    P = {n: p.detach().clone() for n, p in model.named_parameters()}

    out_r, ls_r = run_ref(model, cfg, batch)
    model.zero_grad()
    ls_r[0].backward()
    G_r = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
           for n, p in model.named_parameters()}
    none_grads = [n for n, p in model.named_parameters() if p.grad is None]
    print(f"  reference params with grad None: {none_grads}")

    G_o, ls_o, out_o = O.grads_of(P, Bf, batch, d, hp)
    for k, r in zip(("mel", "log_dur", "stop", "pitch", "energy"), out_r):
        check(f"out.{k}", out_o[k], r.detach(), 2e-5)
    for k, a, b in zip(("total", "mel", "dur", "stop", "pitch", "energy"), ls_o, ls_r):
        check(f"loss.{k}", a, b.detach(), 1e-5)
    worst = 0.0
    for n in names:
        e = float((G_o[n] - G_r[n]).abs().max())
        s = float(G_r[n].abs().max()) + 1e-12
        worst = max(worst, e / max(s, 1e-6))
        assert e <= 2e-6 + 2e-4 * s, (n, e, s)
    print(f"  [ok ] all {len(names)} gradients; worst relative max-err {worst:.2e}")

    fx = {"dims": np.array([d.vocab, d.mel, d.hidden, d.heads, d.enc_layers, d.dec_layers, d.enc_ff,
                            d.dec_ff, d.var_filter, d.var_kernel, d.var_bins, d.max_len])}
    for k, v in batch.items():
        fx[f"batch/{k}"] = v.numpy()
    for k, r in zip(("mel", "log_dur", "stop", "pitch", "energy"), out_r):
        fx[f"out/{k}"] = r.detach().numpy()
    fx["losses"] = np.array([float(x) for x in ls_r], dtype=np.float64)
    if seeded:
        fx["seed"] = np.array(seed)
        fx["grad_norms"] = np.array([float(G_r[n].double().norm()) for n in names])
        fx["grad_sums"] = np.array([float(G_r[n].double().sum()) for n in names])
        for n in names:
            if G_r[n].dim() == 1:
                fx[f"grad/{n}"] = G_r[n].numpy()
    else:
        for n in names:
            fx[f"param/{n}"] = P[n].numpy()
            fx[f"grad/{n}"] = G_r[n].numpy()

    # ---- one full optimizer step through the reference's own step-driver code -------
    if save_step:
        tr = make_trainer(model, cfg)
        tr._setup_optimizer()
        groups = tr.optimizer.param_groups
        id2name = {id(p): n for n, p in model.named_parameters()}
        table = O.group_lr_mult_wd(hp)
        assert len(groups) == 10
        for gi, g in enumerate(groups):
            assert g["group_type"] == O.GROUP_TYPES[gi]
            assert abs(g["lr"] - cfg.learning_rate * table[gi][0]) < 1e-15
            assert abs(g["weight_decay"] - table[gi][1]) < 1e-15
            for p in g["params"]:
                assert O.param_group_of(id2name[id(p)]) == gi, id2name[id(p)]
        print("  [ok ] 10 param groups: membership, lr multipliers, weight decay")
        tr._setup_grad_explosion_tracker()
        # scale grads up so that pre-clip and global clip both actually fire
        with torch.no_grad():
            for p in model.parameters():
                if p.grad is not None:
                    p.grad.mul_(40.0)
        G_big = {n: G_r[n] * 40.0 for n in names}
        clipped = tr._preclip_projection_spikes()
        print(f"  reference pre-clipped {len(clipped)} tensors")
        import copy
        tr.use_ema, tr.ema_update_every, tr.ema_updates = True, 1, 0
        tr.ema_decay = hp.ema_decay
        tr.ema_model = copy.deepcopy(model)
        tr._setup_weight_norm_constraints()
        cfg.dec_ffn_max_weight_norm = WN_CEIL     # low ceiling so projection fires on tiny dims
        ok, post = RuntimeStepPolicy(logging.getLogger("x")).optimizer_step_with_clipping(
            model=model, optimizer=tr.optimizer, use_mixed_precision=False, device_type="cpu",
            scaler=None, mixed_precision_stats={}, clip_norm=cfg.max_grad_norm, step_scheduler=False,
            scheduler_per_batch=True, step_scheduler_fn=lambda: None, update_ema=True,
            update_ema_fn=tr._update_ema)
        assert ok
        tr._apply_weight_norm_constraints()

        hp2 = O.StepHyper(dec_ffn_max_weight_norm=WN_CEIL)
        P2 = {n: P[n].clone() for n in names}
        G2 = {n: G_big[n].clone() for n in names}
        ema = {n: P[n].clone() for n in names}
        ema.update({k: v.clone() for k, v in Bf.items()})
        st = O.OptState()
        info = O.optimizer_step(P2, G2, st, hp2, cfg.learning_rate, cfg.max_grad_norm, ema, Bf)
        n_proj = sum(1 for n in names if O.is_weight_norm_target(n)
                     and float(P2[n].norm()) > WN_CEIL - 1e-3)
        print(f"  oracle: grad_norm={info['grad_norm']:.4f} clip_coef={info['clip_coef']:.5f} "
              f"post_clip(ref)={post:.4f}; weight-norm projected {n_proj} matrices")
        sd, esd = model.state_dict(), tr.ema_model.state_dict()
        for n in names:
            e = float((P2[n] - sd[n]).abs().max())
            assert e <= 1e-7 + 1e-6 * float(sd[n].abs().max()), (n, e)
            e = float((ema[n] - esd[n]).abs().max())
            assert e <= 1e-7 + 1e-6 * float(esd[n].abs().max()), ("ema", n, e)
        for k in Bf:
            check(f"ema buffer {k}", ema[k], esd[k], 1e-6)
        print("  [ok ] params + EMA after one optimizer step (pre-clip, clip, AdamW, EMA, weight-norm)")
        fx["step/grad_scale"] = np.array(40.0)
        fx["step/max_weight_norm"] = np.array(WN_CEIL)
        fx["step/grad_norm"] = np.array(info["grad_norm"])
        fx["step/clip_coef"] = np.array(info["clip_coef"])
        fx["step/param_norms"] = np.array([float(sd[n].double().norm()) for n in names])
        fx["step/param_sums"] = np.array([float(sd[n].double().sum()) for n in names])
        fx["step/delta_norms"] = np.array([float((sd[n].double() - P[n].double()).norm()) for n in names])
        fx["step/ema_delta_norms"] = np.array([float((esd[n].double() - P[n].double()).norm()) for n in names])
        for n in names:
            if sd[n].dim() == 1:
                fx[f"step_param/{n}"] = sd[n].numpy()
                fx[f"step_ema/{n}"] = esd[n].numpy()
        fx["step/preclipped"] = np.array(sorted(clipped.keys()))
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **fx)
    print(f"  wrote {tag}.npz")


def section_inference(d: O.ModelDims, seed: int):
    """forward_inference (autoregressive decode with the KV cache) of the reference on seeded weights: pins
    oracle.generate and is the golden for KokoroEngine.generate.  Only inputs and outputs are stored; both sides
    regenerate the weights from the seed."""
    print(f"== inference: dims={d} seed={seed}")
    model = ref_model(d)
    missing, unexpected = model.load_state_dict(seeded_params(d, seed), strict=False)
    assert not unexpected
    model.eval()
    P = {n: p.detach().clone() for n, p in model.named_parameters()}
    Bf = O.make_buffers(d)
    g = torch.Generator().manual_seed(seed)
    ids1 = torch.randint(1, d.vocab, (1, 7), generator=g)
    ids2 = torch.randint(1, d.vocab, (2, 9), generator=g)
    ids2[1, 6:] = 0                                            # padded second sequence
    st2 = torch.randint(0, 3, (2, 9), generator=g)
    st2[ids2 == 0] = 0
    cases = [("never_stops", ids1, None, dict(max_len=15, stop_threshold=2.0)),
             ("stops_at_min_length", ids1, None, dict(max_len=60, stop_threshold=0.0)),
             ("batch2_padded_stress", ids2, st2, dict(max_len=10, stop_threshold=2.0))]
    save = {"dims": np.array([getattr(d, f) for f in d.__dataclass_fields__]), "seed": np.array(seed)}
    for name, ids, stress, kw in cases:
        with torch.no_grad():
            ref = model.forward_inference(ids, stress_indices=stress, **kw)
        mine, info = O.generate(P, Bf, ids, stress, d, want=True, **kw)
        assert ref.shape == mine.shape, (name, ref.shape, mine.shape)
        check(f"inference {name}: mel {tuple(ref.shape)}", mine, ref, 2e-5)
        save[f"{name}/ids"] = ids.numpy()
        if stress is not None:
            save[f"{name}/stress"] = stress.numpy()
        save[f"{name}/mel"] = ref.numpy()
        save[f"{name}/durations"] = info["durations"].numpy()
        save[f"{name}/kw"] = np.array([kw["max_len"], kw["stop_threshold"]], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "inference_tiny.npz"), **save)


def section_tables():
    print("== tables at default dims")
    d = O.ModelDims()
    model = ref_model(d)
    cfg = TrainingConfig()
    tr = make_trainer(model, cfg)
    tr._setup_optimizer()
    tr._setup_grad_explosion_tracker()
    id2name = {id(p): n for n, p in model.named_parameters()}
    rows = []
    hp = O.StepHyper()
    for p in model.parameters():
        p.grad = torch.ones_like(p) * 1e3          # large → every class clips
    clipped = tr._preclip_projection_spikes()
    for gi, g in enumerate(tr.optimizer.param_groups):
        for p in g["params"]:
            n = id2name[id(p)]
            assert O.param_group_of(n) == gi, n
    for n, p in model.named_parameters():
        mx = O.preclip_max_norm(n, hp)
        assert (n in clipped) == (mx is not None), n
        if mx is not None:
            assert abs(clipped[n][1] - mx) < 1e-12, n
        rows.append({"name": n, "shape": list(p.shape), "group": O.param_group_of(n),
                     "preclip": mx, "weight_norm": O.is_weight_norm_target(n)})
    tr._setup_weight_norm_constraints()
    wn = {id(w) for w in tr._dec_ff_weights + tr._enc_ff_weights}
    assert {id(p) for n, p in model.named_parameters() if O.is_weight_norm_target(n)} == wn
    print(f"  [ok ] {len(rows)} params; {len(clipped)} pre-clip members; {len(wn)} weight-norm targets; "
          f"total {sum(int(np.prod(r['shape'])) for r in rows)} elements")
    with open(os.path.join(HERE, "param_table.json"), "w") as f:
        json.dump({"state_dict_order": list(model.state_dict().keys()), "params": rows}, f, indent=0)

    # LR sequence: drive the reference scheduler exactly as train_epoch does
    for total_steps, warm in ((60, 20), (3000, 1200), (5, 1200)):
        cfg = TrainingConfig(); cfg.warmup_steps = warm
        tr = make_trainer(model, cfg)
        tr._setup_optimizer()

        class _DL:
            def __len__(self): return total_steps
        tr.dataloader = _DL(); cfg.num_epochs = 1; cfg.gradient_accumulation_steps = 1
        tr._setup_scheduler()
        hp = O.StepHyper(warmup_steps=warm)
        sch = O.LRSchedule(hp, total_steps)
        table = O.group_lr_mult_wd(hp)
        seq = []
        for k in range(total_steps + 3):
            lrs = [g["lr"] for g in tr.optimizer.param_groups]
            seq.append(lrs)
            for gi, lr in enumerate(lrs):
                want = sch.base_lr(k) * table[gi][0]
                assert abs(lr - want) <= 1e-12 + 1e-9 * abs(want), (total_steps, k, gi, lr, want)
            tr.optimizer.step()
            tr._step_scheduler_with_warmup()
        print(f"  [ok ] LR sequence total_steps={total_steps} warmup={warm} ({len(seq)} steps × 10 groups)")
        np.save(os.path.join(HERE, f"lr_seq_{total_steps}_{warm}.npy"), np.array(seq))


def section_sampler():
    """Golden batch lists from the reference's DynamicFrameBatchSampler (data/dataset.py:924-1147) on synthetic length
    tables, with the process-global `random` seeded (the reference draws from it): pins kokoro/data/cached.py's
    FrameBudgetBatchSampler, which draws the same stream from random.Random(seed + epoch)."""
    print("== dynamic batch sampler")
    import random as _random
    from kokoro.data.dataset import DynamicFrameBatchSampler

    class FakeDs:
        def __init__(self, lengths):
            self.samples = [{"audio_length": int(x)} for x in lengths]

        def __len__(self):
            return len(self.samples)
    cases = []
    rs = np.random.RandomState(5)
    for name, lengths, kw in [
        ("eleven", [40, 52, 61, 75, 90, 111, 130, 160, 199, 240, 300], dict(max_frames=200, min_batch_size=1, max_batch_size=4)),
        ("two_hundred", rs.randint(60, 1500, size=200).tolist(), dict(max_frames=16384, min_batch_size=4, max_batch_size=32)),
        ("thousand_drop_last", np.clip(rs.lognormal(6.2, 0.5, size=1000), 50, 1800).astype(int).tolist(),
         dict(max_frames=16384, min_batch_size=4, max_batch_size=32, drop_last=True)),
        ("no_shuffle", rs.randint(60, 900, size=150).tolist(), dict(max_frames=8000, min_batch_size=2, max_batch_size=16, shuffle=False)),
        ("ties", [100] * 37 + [200] * 23, dict(max_frames=1000, min_batch_size=1, max_batch_size=8)),
    ]:
        for seed in (0, 7):
            _random.seed(seed)
            ref = DynamicFrameBatchSampler(FakeDs(lengths), **kw)
            cases.append({"name": name, "lengths": lengths, "kwargs": kw, "seed": seed, "batches": [list(map(int, b)) for b in ref.batches]})
            print(f"  {name} seed {seed}: {len(ref.batches)} batches")
    with open(os.path.join(HERE, "sampler.json"), "w") as f:
        json.dump(cases, f)


def section_surface():
    """Dump the reference's Python surface for the drop-in tests: TrainingConfig fields and kokoro-train flags."""
    print("== drop-in surface")
    import argparse
    import dataclasses
    import kokoro.cli.cli as rcli
    fields = [{"name": f.name, "default": repr(f.default)} for f in dataclasses.fields(TrainingConfig) if f.name != "device"]
    cap = {}
    orig = argparse.ArgumentParser.parse_args

    def fake(self, args=None, namespace=None):
        cap["p"] = self
        return orig(self, [])
    argparse.ArgumentParser.parse_args = fake
    try:
        ns = rcli.parse_arguments()
    finally:
        argparse.ArgumentParser.parse_args = orig
    flags = [{"flags": sorted(a.option_strings), "dest": a.dest, "default": repr(a.default), "action": type(a).__name__,
              "type": getattr(a.type, "__name__", None)} for a in cap["p"]._actions if a.dest != "help"]
    cfg = rcli.create_config_from_args(ns)
    mapped = {f.name: repr(getattr(cfg, f.name)) for f in dataclasses.fields(TrainingConfig) if f.name != "device"}
    with open(os.path.join(HERE, "surface.json"), "w") as f:
        json.dump({"config_fields": fields, "cli_flags": flags, "config_from_default_args": mapped}, f, indent=0)
    print(f"  wrote surface.json ({len(fields)} config fields, {len(flags)} flags)")
    # the reference model strictly loads a state dict built from this repo's tables
    d = O.ModelDims()
    m = ref_model(d)
    sd = dict(O.init_params(d, 1))
    sd.update(O.make_buffers(d))
    m.load_state_dict(sd, strict=True)
    print("  [ok ] reference KokoroModel.load_state_dict(strict=True) accepts our 311-entry state dict")


def section_lengths():
    print("== length regulator")
    cases = []
    g = torch.Generator().manual_seed(7)

    def add(tokens, dur, max_len):
        ref = vectorized_expand_tokens(tokens, dur, max_len=max_len)
        got = O.length_regulate(tokens, dur, max_len)
        assert ref.shape == got.shape, (ref.shape, got.shape)
        assert torch.equal(ref, got)
        idx, lens, L = O.length_regulate_index(dur.numpy(), max_len)
        cases.append({"tokens": tokens.numpy(), "dur": dur.numpy(), "max_len": -1 if max_len is None else max_len,
                      "out": ref.numpy(), "idx": idx, "lens": lens})
    # the reference's own known answers (tests/unit/test_utils_lengths.py:9-41)
    add(torch.tensor([[1, 2, 3], [4, 5, 6]]), torch.tensor([[1, 2, 0], [0, 1, 2]]), None)
    add(torch.tensor([[1, 2, 3], [4, 5, 6]]), torch.tensor([[1, 2, 0], [0, 1, 2]]), 2)
    add(torch.tensor([[[1.0], [2.0]], [[3.0], [4.0]]]), torch.tensor([[0, 0], [0, 0]]), None)
    add(torch.tensor([[1, 2]]), torch.tensor([[1, 1]]), 4)
    add(torch.tensor([[1, 2], [3, 4]]), torch.tensor([[1, 2], [0, 1]]), None)
    add(torch.tensor([[[1.0], [2.0]]]), torch.tensor([[0, 0]]), 5)
    for B, Pn, H, mx, ml in ((3, 7, 4, 6, None), (4, 33, 8, 9, 100), (2, 128, 4, 12, 700), (5, 16, 2, 3, 10)):
        tok = torch.randn(B, Pn, H, generator=g)
        dur = torch.randint(0, mx, (B, Pn), generator=g)
        dur[0, Pn // 2:] = 0
        add(tok, dur, ml)
        add(tok, dur.float() + 0.7, ml)              # float durations truncate (lengths.py:31)
        add(tok, dur - 1, ml)                         # negatives clamp to 0
    fx = {}
    for i, c in enumerate(cases):
        for k, v in c.items():
            fx[f"{i}/{k}"] = np.asarray(v)
    fx["n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "length_regulator.npz"), **fx)
    print(f"  [ok ] {len(cases)} cases bit-exact; wrote length_regulator.npz")


def section_loss_known_answers():
    print("== loss known answers")
    # reference's own: tests/unit/test_trainer_loss_stability.py:35-72 (total = ln 2)
    d = O.ModelDims()
    hp = O.StepHyper(duration_loss_weight=0.0, stop_token_loss_weight=1.0, stop_token_pos_weight=1.0,
                     pitch_loss_weight=0.0, energy_loss_weight=0.0)
    batch = {"mel_specs": torch.zeros(1, 2, 80), "phoneme_durations": torch.ones(1, 2, dtype=torch.long),
             "stop_token_targets": torch.zeros(1, 2), "mel_lengths": torch.tensor([1]),
             "phoneme_lengths": torch.tensor([2]), "pitches": torch.zeros(1, 2), "energies": torch.zeros(1, 2)}
    mel_pred = torch.zeros(1, 2, 80); mel_pred[0, 1, 0] = float("nan")
    out = {"mel": mel_pred, "log_dur": torch.log(torch.tensor([[2.0, 2.0]])), "stop": torch.zeros(1, 2),
           "pitch": torch.zeros(1, 2), "energy": torch.zeros(1, 2)}
    ls = O.losses(out, batch, hp)
    assert abs(float(ls[0]) - 0.6931472) < 1e-6, float(ls[0])
    print("  [ok ] total = ln 2 with NaN in a padded frame")


def _reference_block(first: str, last: str) -> str:
    """The statements of KokoroTrainer.train_epoch from the line containing `first` to the line containing `last`
    (inclusive), dedented — the reference keeps the batch-shape heuristics and the explosion bookkeeping inline in its
    epoch loop, so the only way to RUN them is to lift the lines out of the imported source."""
    import inspect
    import textwrap
    lines = inspect.getsource(KokoroTrainer.train_epoch).splitlines()
    a = next(i for i, l in enumerate(lines) if first in l)
    b = next(i for i in range(a, len(lines)) if last in lines[i])
    return textwrap.dedent("\n".join(lines[a:b + 1]))


def section_step_driver():
    """Pins the oracle's step-boundary heuristics against the reference's own statements, executed from its source:
    adaptive loss scale / clip norm from the batch shape (trainer.py:2218-2242) and the gradient-explosion tracker
    (trainer.py:914-925, 1315-1330, 2367-2405) over a scripted sequence of gradient norms."""
    print("== step driver: batch-shape heuristics + explosion tracker")
    cfg = TrainingConfig()
    hp = O.StepHyper()
    model = ref_model(O.ModelDims(vocab=59, mel=20, hidden=64, heads=1, enc_layers=1, dec_layers=1, enc_ff=32, dec_ff=32,
                                  var_filter=32, var_kernel=3, var_bins=16, max_len=100))
    quiet = logging.getLogger("quiet")
    # ---- adaptive loss scale / clip (the block ends with the hard-risk assignments; logging follows it) ----
    src = _reference_block("max_mel_length = 1400", "adaptive_clip_norm = max(0.05")
    code = compile(src, "<trainer.py:2218-2242>", "exec")
    rows = []
    for T in (64, 512, 1399, 1400, 1401, 1500, 1800, 2000, 2800, 5600, 9000):
        for md in (1, 40, 149, 150, 151, 200, 300, 450, 600, 1200):
            tr = make_trainer(model, cfg)
            ns = {"self": tr, "mel_specs": torch.zeros(1, T, 1), "phoneme_durations": torch.tensor([[1, md]]),
                  "logger": quiet, "batch_idx": 0, "getattr": getattr, "max": max, "min": min}
            exec(code, ns)
            scale, clip = O.adaptive_loss_scale_and_clip(T, md, hp.max_grad_norm)
            assert abs(ns["adaptive_loss_scale"] - scale) < 1e-15 and abs(ns["adaptive_clip_norm"] - clip) < 1e-15, (T, md)
            rows.append([T, md, ns["adaptive_loss_scale"], ns["adaptive_clip_norm"]])
    print(f"  [ok ] adaptive loss scale / clip norm: {len(rows)} (mel length, max duration) pairs")
    # ---- explosion tracker over a scripted norm sequence ----
    src = _reference_block("grad_norm_ema_for_threshold =", "self.grad_explosion_ema_steps += 1")
    code = compile(src, "<trainer.py:2367-2405>", "exec")
    rs = np.random.RandomState(3)
    norms = np.abs(rs.lognormal(1.0, 0.6, size=520)) * np.linspace(40.0, 1.0, 520)     # early training: large, decaying
    norms[37] = 9500.0                    # above the warm-up floor (8000 -> 1000 over 400 steps)
    norms[120:123] = [3000.0, 6500.0, 7000.0]       # streak of three while the floor is ~5900: only two are above it
    norms[260] = 2600.0                   # floor 3450 at step 260, EMA-based threshold lower than the floor: quiet
    norms[430] = 1200.0                   # past the floor's warm-up: threshold max(1000, 3*EMA)
    norms[470] = float(norms[440:470].mean() * 3.5) + 1000.0
    norms[500] = 900.0
    tr = make_trainer(model, cfg)
    tr._setup_grad_explosion_tracker()
    tr.optimizer_steps_completed = 0
    tr._has_nonfinite_gradients = lambda: False
    mine = O.ExplosionTracker(hp)
    out = []
    for k, nv in enumerate(norms.tolist()):
        thr_ref = tr._compute_grad_explosion_threshold()
        thr_mine = mine.threshold(k)
        assert abs(thr_ref[0] - thr_mine[0]) <= 1e-9 * max(1.0, abs(thr_ref[0])) and thr_ref[2] == thr_mine[2], (k, thr_ref, thr_mine)
        ns = {"self": tr, "total_grad_norm": nv, "adaptive_clip_norm": cfg.max_grad_norm, "logger": quiet, "batch_idx": k,
              "grad_norms_by_param": [], "has_nonfinite_grads": False}
        exec(code, ns)
        clip, expl = mine.observe(nv, k, hp.max_grad_norm)
        assert clip == ns["adaptive_clip_norm"] and expl == bool(ns["is_exploding"]), (k, nv, clip, ns["adaptive_clip_norm"])
        assert abs(mine.ema - tr.grad_explosion_norm_ema) <= 1e-9 * abs(mine.ema) and mine.streak == tr.grad_explosion_streak
        out.append([nv, thr_ref[0], clip, float(expl), tr.grad_explosion_norm_ema, float(tr.grad_explosion_streak)])
        tr.optimizer_steps_completed += 1          # (every step of this sequence succeeds)
    n_expl = int(sum(r[3] for r in out))
    assert n_expl >= 5, n_expl
    print(f"  [ok ] explosion tracker: {len(out)} steps, {n_expl} explosions, thresholds / clip norms / EMA identical")
    np.savez_compressed(os.path.join(HERE, "step_driver.npz"), adaptive=np.array(rows, dtype=np.float64),
                        explosion=np.array(out, dtype=np.float64))


def section_expanded_length(d: O.ModelDims):
    """Expanded length T' = max_b sum(dur) != mel length T (model.py:607-628): T' > T is served (predictors on T' frames,
    memory and losses on the first T), T' < T fails in the reference's pitch loss (losses.py:126)."""
    print("== expanded length != mel length")
    cfg = TrainingConfig()
    hp = O.StepHyper()
    model = ref_model(d)
    P = seeded_params(d, 31)
    sd = dict(P)
    sd.update(O.make_buffers(d))
    model.load_state_dict(sd, strict=True)
    Bf = O.make_buffers(d)
    save = {"dims": np.array([getattr(d, f) for f in ("vocab", "mel", "hidden", "heads", "enc_layers", "dec_layers", "enc_ff",
                                                       "dec_ff", "var_filter", "var_kernel", "var_bins", "max_len")]),
            "seed": np.array(31)}
    for tag, B, T, Pn, extra in (("longer", 3, 37, 9, (5, 0, 11)), ("longer_chunk", 2, 509, 30, (9, 2))):
        batch = O.synthetic_batch(B, T, Pn, d, seed=40 + B, ragged=True)
        dur = batch["phoneme_durations"].clone()
        for b, e in enumerate(extra):
            dur[b, 1] += e
        batch["phoneme_durations"] = dur
        Tp = int(dur.sum(1).max())
        assert Tp > T
        out, ls = run_ref(model, cfg, batch)
        mine = O.forward(P, Bf, batch, d)
        lm = O.losses(mine, batch, hp)
        for k, i in (("mel", 0), ("log_dur", 1), ("stop", 2), ("pitch", 3), ("energy", 4)):
            assert tuple(out[i].shape) == tuple(mine[k].shape), (k, out[i].shape, mine[k].shape)
            check(f"{tag} (T={T}, T'={Tp}) output {k} {tuple(out[i].shape)}", mine[k], out[i].detach(), 2e-6, 1e-6)
        for a, b in zip(lm, ls):
            check(f"{tag} loss", a, b.detach(), 2e-6, 1e-6)
        G, _, _ = O.grads_of(P, Bf, batch, d, hp)
        model.zero_grad()
        ls[0].backward()
        worst = 0.0
        for n, p in model.named_parameters():
            ref = p.grad if p.grad is not None else torch.zeros_like(p)
            if float(ref.norm()) > 1e-9:
                worst = max(worst, float((G[n] - ref).norm() / ref.norm()))
        assert worst < 1e-4, worst
        print(f"  [ok ] {tag}: all gradients, worst relative error {worst:.2e}")
        for k, v in batch.items():
            save[f"{tag}/batch/{k}"] = v.numpy()
        for k, i in (("mel", 0), ("log_dur", 1), ("stop", 2), ("pitch", 3), ("energy", 4)):
            save[f"{tag}/out/{k}"] = out[i].detach().numpy()
        save[f"{tag}/losses"] = np.array([float(x) for x in ls])
        names = list(O.param_shapes(d))
        save[f"{tag}/grad_norms"] = np.array([float(G[n].norm()) for n in names])
    # shorter: both sides must raise the size-mismatch RuntimeError
    batch = O.synthetic_batch(2, 40, 6, d, seed=3, ragged=True)
    batch["phoneme_durations"][:, 0] = (batch["phoneme_durations"][:, 0] - 3).clamp(min=0)
    for who, fn in (("reference", lambda: run_ref(model, cfg, batch)), ("oracle", lambda: O.losses(O.forward(P, Bf, batch, d), batch, hp))):
        try:
            fn()
        except RuntimeError as e:
            assert "must match the size of tensor" in str(e), str(e)
            print(f"  [ok ] {who} raises for T' < T: {str(e)[:70]}")
        else:
            raise AssertionError(f"{who} accepted T' < T")
    np.savez_compressed(os.path.join(HERE, "expanded_length.npz"), **save)


def section_autocast(d: O.ModelDims, B, T, Pn, seed, ragged, name="autocast_bf16", keep_autocast_mel=True):
    """The reference's OWN bf16 mode (trainer.py:3181-3232: the model forward under torch.autocast(dtype=bfloat16), losses and
    backward outside it) on the `full_dims` batch and weights, next to its fp32 run: how far the reference's mixed precision
    moves losses, outputs and every gradient away from its fp32 numbers.  The MI355X engine's bf16 mode (bf16 MFMA operands and
    operand storage, fp32 accumulate / residual streams / statistics) is held to that yardstick on the GPU
    (tests/test_engine_gpu.py::test_bf16_mode_against_the_references_own_autocast)."""
    print(f"== {name}: dims={d} batch=({B},{T},{Pn})")
    cfg = TrainingConfig()
    model = ref_model(d)
    # The reference enters autocast on CUDA only (trainer.py:935-956); here its model runs under the CPU autocast of the same dtype,
    # the one emulation this container allows.  The bf16 predictions are widened to fp32 before the losses (exact): the CPU
    # HuberLoss backward refuses a bf16 prediction against an fp32 target ("Found dtype Float but expected BFloat16"), where the CUDA
    # kernel promotes to fp32 internally — the same arithmetic.
    missing, unexpected = model.load_state_dict(seeded_params(d, seed), strict=False)
    assert not unexpected
    names = [n for n, _ in model.named_parameters()]
    batch = O.synthetic_batch(B, T, Pn, d, seed=seed, ragged=ragged)
    if ragged:
        batch["phoneme_indices"][0, 2] = 0
    res = {}
    for mode in ("fp32", "autocast"):
        model.zero_grad()
        if mode == "autocast":
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = model(batch["phoneme_indices"], batch["mel_specs"], batch["phoneme_durations"], batch["stop_token_targets"],
                            pitch_targets=batch["pitches"], energy_targets=batch["energies"], stress_indices=batch["stress_indices"])
        else:
            out = model(batch["phoneme_indices"], batch["mel_specs"], batch["phoneme_durations"], batch["stop_token_targets"],
                        pitch_targets=batch["pitches"], energy_targets=batch["energies"], stress_indices=batch["stress_indices"])
        ls = ref_losses(model, cfg, [o.float() for o in out], batch)
        ls[0].backward()
        res[mode] = ([o.detach().float().clone() for o in out], [float(x) for x in ls],
                     {n: (p.grad.detach().float().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()})
        print(f"  {mode:8s} output dtype {out[0].dtype}, losses {[round(float(x), 5) for x in ls]}")
    (o32, l32, g32), (o16, l16, g16) = res["fp32"], res["autocast"]
    cos, ratio = [], []
    for n in names:
        a, b = g16[n].double().flatten(), g32[n].double().flatten()
        nb = float(b.norm())
        cos.append(float(a @ b / (a.norm() * b.norm() + 1e-300)) if nb > 1e-12 else 1.0)
        ratio.append(float(a.norm()) / nb if nb > 1e-12 else 1.0)
    valid = (torch.arange(T)[None, :] < batch["mel_lengths"][:, None])[:, :, None].float()
    mel_l1 = float(((o16[0] - o32[0]).abs() * valid).sum() / (valid.sum() * d.mel))
    print(f"  autocast vs fp32: loss deltas {[round(a - b, 5) for a, b in zip(l16, l32)]}; mel-L1 between the two mel outputs {mel_l1:.4e}; "
          f"gradient cosine min {min(cos):.4f} mean {float(np.mean(cos)):.5f}; norm ratio {min(ratio):.3f}..{max(ratio):.3f}")
    extra = {"mel_autocast": o16[0].numpy()} if keep_autocast_mel else {}      # (the T = 512 fixture keeps the fp32 output only: size)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=np.array(seed), shape=np.array([B, T, Pn]),
                        dims=np.array(list(d.__dict__.values())),
                        losses_fp32=np.array(l32), losses_autocast=np.array(l16), grad_cos=np.array(cos), grad_norm_ratio=np.array(ratio),
                        grad_norms_autocast=np.array([float(g16[n].double().norm()) for n in names]),
                        grad_norms_fp32=np.array([float(g32[n].double().norm()) for n in names]),
                        mel_l1_between=np.array(mel_l1), mel_fp32=o32[0].numpy(), **extra,
                        **{f"batch/{k}": v.numpy() for k, v in batch.items()})
    print(f"  wrote {name}.npz")


if __name__ == "__main__":
    tiny = O.ModelDims(vocab=59, mel=20, hidden=128, heads=2, enc_layers=2, dec_layers=2, enc_ff=96,
                       dec_ff=96, var_filter=32, var_kernel=3, var_bins=16, max_len=700)
    if sys.argv[1:] == ["inference"]:                            # only the decode fixture
        section_inference(tiny, seed=21)
        sys.exit(0)
    if sys.argv[1:] == ["autocast"]:                             # only the reference-autocast fixtures
        section_autocast(O.ModelDims(), B=2, T=96, Pn=12, seed=14, ragged=True)
        section_autocast(O.ModelDims(), B=2, T=512, Pn=64, seed=15, ragged=True, name="autocast_bf16_t512", keep_autocast_mel=False)
        sys.exit(0)
    if sys.argv[1:] == ["step"]:                                 # only the step-driver / expanded-length fixtures
        section_step_driver()
        section_expanded_length(O.ModelDims(vocab=59, mel=80, hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=192,
                                            dec_ff=192, var_filter=64, var_kernel=3, var_bins=256, max_len=700))
        sys.exit(0)
    section_surface()
    section_sampler()
    section_lengths()
    section_loss_known_answers()
    section_model("tiny_full", tiny, B=2, T=40, Pn=6, seed=11, ragged=False, save_step=False)
    section_model("tiny_ragged", tiny, B=3, T=37, Pn=9, seed=12, ragged=True, save_step=True)
    # chunk boundary of the variance predictors (T > 512 ⇒ two GroupNorm chunks), head_dim 64
    mid = O.ModelDims(vocab=59, mel=80, hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=192,
                      dec_ff=192, var_filter=64, var_kernel=3, var_bins=256, max_len=700)
    section_model("mid_chunked", mid, B=2, T=600, Pn=40, seed=13, ragged=True, save_step=False, seeded=True)
    # default dims (49.4 M params), small ragged batch: weights regenerated from the seed on both sides
    section_model("full_dims", O.ModelDims(), B=2, T=96, Pn=12, seed=14, ragged=True, save_step=False, seeded=True)
    section_tables()
    section_step_driver()
    section_expanded_length(mid)
    section_inference(tiny, seed=21)
    section_autocast(O.ModelDims(), B=2, T=96, Pn=12, seed=14, ragged=True)
    # ... and at the bench's frame count (T = 512: the decoder's attention and the 512-frame GroupNorm chunk at full length)
    section_autocast(O.ModelDims(), B=2, T=512, Pn=64, seed=15, ragged=True, name="autocast_bf16_t512", keep_autocast_mel=False)
    print("ALL REFERENCE CHECKS PASSED")
