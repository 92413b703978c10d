"""CPU suite: the drop-in Python surface (TrainingConfig, kokoro-train flags, cached-feature batches, DP sharding)
against tests/golden/surface.json — a dump of the reference's own dataclass and argparse parser."""
import argparse
import dataclasses
import json
import os
import pickle

import pytest
import torch

from kokoro.cli import cli
from kokoro.data import cached
from kokoro.training.config import TrainingConfig


@pytest.fixture(scope="module")
def surface(golden_dir):
    return json.load(open(os.path.join(golden_dir, "surface.json")))


def test_training_config_field_for_field(surface):
    mine = [f for f in dataclasses.fields(TrainingConfig) if f.name != "device"]
    ref = surface["config_fields"]
    assert [f.name for f in mine][:len(ref)] == [r["name"] for r in ref]       # same names, same order
    for f, r in zip(mine, ref):
        assert repr(f.default) == r["default"], f.name
    extra = [f.name for f in mine][len(ref):]
    assert extra == ["mixed_precision_dtype", "dp_world_size", "replica_check_every"]                 # new fields come last, with defaults
    c = TrainingConfig(data_dir="/x/corpus", checkpoint_segments=0)
    assert c.feature_cache_dir == "/x/corpus/.feature_cache" and c.checkpoint_segments == 1
    c2 = pickle.loads(pickle.dumps(c))                                         # checkpoints pickle the config object
    assert type(c2).__module__ == "kokoro.training.config" and c2 == c


def test_cli_flag_for_flag(surface):
    p = cli.build_parser()
    mine = [{"flags": sorted(a.option_strings), "dest": a.dest, "default": repr(a.default), "action": type(a).__name__,
             "type": getattr(a.type, "__name__", None)} for a in p._actions if a.dest != "help"]
    key = lambda d: tuple(d["flags"])
    assert sorted(mine, key=key) == sorted(surface["cli_flags"], key=key)
    cfg = cli.create_config_from_args(cli.parse_arguments([]))
    for k, v in surface["config_from_default_args"].items():
        if k == "use_mixed_precision":          # AMP default follows torch.cuda.is_available() on both sides
            continue
        assert repr(getattr(cfg, k)) == v, k
    cfg = cli.create_config_from_args(cli.parse_arguments(
        ["--no-mfa", "--no-dynamic-batching", "--batch-size", "1", "--epochs", "1", "--no-validation", "-lr", "1e-4"]))
    assert (cfg.use_mfa, cfg.use_dynamic_batching, cfg.batch_size, cfg.num_epochs, cfg.validation_split, cfg.learning_rate) == \
        (False, False, 1, 1, 0.0, 1e-4)
    with pytest.raises(SystemExit):
        cli.parse_arguments(["--enable-amp", "--disable-amp"])


def _fake_cache(tmp_path, n=11):
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = tmp_path / ".feature_cache"
    d.mkdir()
    g = torch.Generator().manual_seed(0)
    for i in range(n):
        T, P = int(torch.randint(20, 90, (1,), generator=g)), int(torch.randint(3, 12, (1,), generator=g))
        b = synthetic_batch(1, T, P, seed=i)
        torch.save({"mel_spec": b["mel_specs"][0].T.contiguous(), "phoneme_indices": b["phoneme_indices"][0],
                    "stress_indices": b["stress_indices"][0], "phoneme_durations": b["phoneme_durations"][0],
                    "stop_token_targets": b["stop_token_targets"][0], "pitch": b["pitches"][0], "energy": b["energies"][0],
                    "text": f"utt {i}", "audio_file": f"utt{i:03d}", "mel_length": T, "phoneme_length": P,
                    "_cache_version": 7}, d / f"utt{i:03d}.pt")
    return str(d)


def test_cached_features_collate_and_samplers(tmp_path):
    cdir = _fake_cache(tmp_path)
    ds = cached.CachedFeatureDataset(cdir)
    assert len(ds) == 11 and [m["audio_length"] for m in ds.samples] == sorted(m["audio_length"] for m in ds.samples)
    tr, va = cached.split_indices(len(ds), 0.1)
    assert len(tr) == 9 and len(va) == 2 and sorted(tr + va) == list(range(11))
    assert cached.split_indices(1, 0.1) == ([], [0])            # BASELINE config 1: the single wav lands in validation
    batch = cached.collate_fn([ds[0], ds[5], ds[10]])
    T, P = int(batch["mel_lengths"].max()), int(batch["phoneme_lengths"].max())
    assert batch["mel_specs"].shape == (3, T, 80) and batch["phoneme_indices"].shape == (3, P)
    for i in range(3):
        t, p = int(batch["mel_lengths"][i]), int(batch["phoneme_lengths"][i])
        assert batch["mel_specs"][i, t:].abs().sum() == 0 and batch["phoneme_durations"][i, p:].sum() == 0
        assert int(batch["phoneme_durations"][i].sum()) == t   # durations sum to the mel length (engine contract)
    s = cached.FrameBudgetBatchSampler(ds, max_frames=200, min_batch_size=1, max_batch_size=4, shuffle=False)
    bl = s.batches()
    assert sorted(i for b in bl for i in b) == list(range(11))
    for b in bl:
        assert len(b) <= 4 and (len(b) == 1 or len(b) * max(ds.samples[i]["audio_length"] for i in b) <= 200)
    parts = [cached.FrameBudgetBatchSampler(ds, 200, 1, 4, True, r, 2, seed=3).batches() for r in range(2)]
    glob = cached.FrameBudgetBatchSampler(ds, 200, 1, 4, True, 0, 1, seed=3).batches()
    assert len(parts[0]) == len(parts[1]) == len(glob) // 2                            # same number of steps on every rank
    assert [b for pair in zip(*parts) for b in pair] == glob[:len(glob) // 2 * 2]      # ranks interleave the global list
    with pytest.raises(FileNotFoundError):
        cached.CachedFeatureDataset(str(tmp_path / "nothing"))


def test_dynamic_batch_sampler_matches_reference(golden_dir):
    """FrameBudgetBatchSampler == the reference's DynamicFrameBatchSampler (golden batch lists dumped from the reference
    with the global RNG seeded): quantile buckets, greedy packing, min-size/drop_last, heavy-batch spreading, ties."""
    import json
    import types
    from kokoro.data import cached
    cases = json.load(open(os.path.join(golden_dir, "sampler.json")))
    assert len(cases) >= 10
    for c in cases:
        ds = types.SimpleNamespace(samples=[{"audio_length": x} for x in c["lengths"]])
        kw = dict(c["kwargs"])
        s = cached.FrameBudgetBatchSampler(ds, kw.pop("max_frames"), kw.pop("min_batch_size"), kw.pop("max_batch_size"),
                                           kw.pop("shuffle", True), 0, 1, seed=c["seed"], drop_last=kw.pop("drop_last", False))
        assert not kw
        assert s.global_batches() == c["batches"], c["name"]
        if c["kwargs"].get("shuffle", True):
            s.epoch = 1
            assert s.global_batches() != c["batches"], "a new epoch must reshuffle"


def test_trainer_helpers():
    from kokoro.training.trainer import cap_batch, effective_accumulation_divisor, recommended_ema_decay
    # reference tests/unit/test_trainer_accumulation_divisor.py:4-35
    assert effective_accumulation_divisor(4, 0, 0, 10) == 4
    assert effective_accumulation_divisor(4, 0, 8, 10) == 2
    assert effective_accumulation_divisor(4, 1, 9, 10) == 2
    assert effective_accumulation_divisor(1, 0, 3, 10) == 1
    assert abs(recommended_ema_decay(677, 1.0) - 0.998976) < 1e-5 and recommended_ema_decay(0, 1.0) == 0.9999
    b = {"mel_specs": torch.zeros(2, 30, 4), "stop_token_targets": torch.zeros(2, 30), "pitches": torch.zeros(2, 30),
         "energies": torch.zeros(2, 30), "mel_lengths": torch.tensor([30, 12]), "phoneme_indices": torch.zeros(2, 5, dtype=torch.long),
         "phoneme_durations": torch.zeros(2, 5, dtype=torch.long), "stress_indices": torch.zeros(2, 5, dtype=torch.long),
         "phoneme_lengths": torch.tensor([5, 3])}
    c = cap_batch(b, max_mel=20, max_ph=4)
    assert c["mel_specs"].shape == (2, 20, 4) and c["mel_lengths"].tolist() == [20, 12] and c["phoneme_indices"].shape == (2, 4)


def test_clip_durations_keeps_the_prefix():
    import torch
    from kokoro.data.cached import clip_durations
    from oracle import kokoro_oracle as O
    dur = torch.tensor([3, 0, 5, 2, 7, 1])
    for T in (0, 1, 3, 4, 8, 10, 17, 18, 40):
        c = clip_durations(dur, T)
        assert int(c.sum()) == min(int(dur.sum()), T) and bool((c >= 0).all()) and bool((c <= dur).all())
        idx_full, _, _ = O.length_regulate_index(dur[None].numpy(), None)
        idx_clip, lens, _ = O.length_regulate_index(c[None].numpy(), None)
        n = int(c.sum())
        assert (idx_clip[0, :n] == idx_full[0, :n]).all()               # the expansion is a prefix of the original one


def test_clip_durations_differs_from_the_reference_rule_on_purpose():
    """The reader's own clip (a cache longer than the run's max_seq_length) keeps sum(dur) == T; the reference's precompute-time
    reconciliation (dataset.py:769-776) keeps every duration >= 1 and may leave sum(dur) > T.  Both pinned, so nobody takes one
    for the other."""
    import torch
    from kokoro.data.cached import clip_durations, reference_reconcile
    dur = torch.tensor([3, 0, 5, 2, 7, 1])                            # sums to 18
    assert reference_reconcile(dur, 18).tolist() == [3, 1, 5, 2, 7, 1]       # clamp(min=1) alone: T' = 19 > T
    assert reference_reconcile(dur, 12).tolist() == [3, 1, 5, 2, 7, 1]       # last = max(1, 1 - 6) = 1: T' = 19 > T = 12
    assert reference_reconcile(dur, 25).tolist() == [3, 1, 5, 2, 7, 8]       # last += 7
    assert clip_durations(dur, 12).tolist() == [3, 0, 5, 2, 2, 0] and int(clip_durations(dur, 12).sum()) == 12


def test_scan_cache_discards_a_stale_index(tmp_path):
    """An index written for another FEATURE_CACHE_VERSION (or by the round-2 reader: second-resolution mtimes, no per-file
    version) is not trusted: every file is re-read, and a file of the wrong version is refused."""
    import json
    import pytest
    import torch
    from kokoro.data import cached
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = tmp_path / ".feature_cache"
    d.mkdir()
    b = synthetic_batch(1, 40, 5, seed=0)
    item = {"mel_spec": b["mel_specs"][0].T.contiguous(), "phoneme_indices": b["phoneme_indices"][0], "stress_indices": b["stress_indices"][0],
            "phoneme_durations": b["phoneme_durations"][0], "stop_token_targets": b["stop_token_targets"][0], "pitch": b["pitches"][0],
            "energy": b["energies"][0], "text": "x", "audio_file": "u0", "mel_length": 40, "phoneme_length": 5, "_cache_version": 6}
    torch.save(item, d / "u0.pt")
    st = (d / "u0.pt").stat()
    # a round-2 style index vouching for the old file with a wrong length
    (d / ".kk_index.json").write_text(json.dumps({"version": 7, "entries": [
        {"name": "u0.pt", "size": st.st_size, "mtime": int(st.st_mtime), "mel_length": 999, "phoneme_length": 5}]}))
    with pytest.raises(RuntimeError, match="feature cache version 6"):
        cached.scan_cache(str(d))
    item["_cache_version"] = 7
    torch.save(item, d / "u0.pt")
    assert cached.scan_cache(str(d))[0]["audio_length"] == 40
    idx = json.loads((d / ".kk_index.json").read_text())
    assert idx["index_format"] == 2 and idx["entries"][0]["cache_version"] == 7 and "mtime_ns" in idx["entries"][0]
    idx["version"] = 6                                                 # an index of another schema version
    idx["entries"][0]["mel_length"] = 123
    (d / ".kk_index.json").write_text(json.dumps(idx))
    assert cached.scan_cache(str(d))[0]["audio_length"] == 40


def test_cached_dataset_scans_once_and_clips_durations(tmp_path):
    import torch
    from kokoro.data import cached
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = tmp_path / ".feature_cache"
    d.mkdir()
    for i, (T, P) in enumerate(((50, 6), (90, 9), (70, 5))):
        b = synthetic_batch(1, T, P, seed=i)
        torch.save({"mel_spec": b["mel_specs"][0].T.contiguous(), "phoneme_indices": b["phoneme_indices"][0],
                    "stress_indices": b["stress_indices"][0], "phoneme_durations": b["phoneme_durations"][0],
                    "stop_token_targets": b["stop_token_targets"][0], "pitch": b["pitches"][0], "energy": b["energies"][0],
                    "text": "x", "audio_file": f"u{i}", "mel_length": T, "phoneme_length": P, "_cache_version": 7}, d / f"u{i}.pt")
    metas = cached.scan_cache(str(d))
    assert [m["audio_length"] for m in metas] == [50, 70, 90] and (d / ".kk_index.json").exists()
    loads = []
    orig = torch.load
    torch.load = lambda *a, **k: (loads.append(1), orig(*a, **k))[1]
    try:
        assert [m["audio_length"] for m in cached.scan_cache(str(d))] == [50, 70, 90]
    finally:
        torch.load = orig
    assert not loads, "the second scan must come from the index file"
    ds = cached.CachedFeatureDataset(str(d), None, max_seq_length=60, memory_cache=False, metas=metas)
    assert [m["audio_length"] for m in ds.samples] == [50, 60, 60]
    for i in range(3):
        it = ds[i]
        T = int(it["mel_length"])
        assert T <= 60 and it["mel_spec"].shape[1] == T and int(it["phoneme_durations"].sum()) == T
    batch = cached.collate_fn([ds[i] for i in range(3)])
    assert int(batch["phoneme_durations"].sum(1).max()) == batch["mel_specs"].shape[1]


def test_collate_into_equals_capped_collate_fn(tmp_path):
    """The loader thread's numpy fill of a flat staging buffer == cap_batch(collate_fn(...)) of the reference contract."""
    import numpy as np
    import torch
    from kokoro.data import cached
    from kokoro.training.trainer import cap_batch
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    items = []
    for i, (T, P) in enumerate(((50, 6), (90, 9), (70, 5), (33, 12))):
        b = synthetic_batch(1, T, P, seed=i)
        items.append({"mel_spec": b["mel_specs"][0].T.contiguous(), "phoneme_indices": b["phoneme_indices"][0],
                      "stress_indices": b["stress_indices"][0], "phoneme_durations": b["phoneme_durations"][0],
                      "stop_token_targets": b["stop_token_targets"][0], "pitch": b["pitches"][0], "energy": b["energies"][0],
                      "mel_length": T, "phoneme_length": P})
    for max_mel, max_ph in ((2000, 2000), (60, 8)):
        ref = cap_batch(cached.collate_fn(items), max_mel, max_ph)
        B, T, P, M, mel_len, ph_len, plan, total = cached.batch_layout(items, max_mel, max_ph)
        buf = np.full(total, 0xAB, dtype=np.uint8)                       # dirty memory: padding must be written
        out = {k: buf[off:off + n].view(dt).reshape(shape) for k, off, n, shape, dt in plan}
        cached.collate_into(items, out, mel_len, ph_len)
        assert set(out) == set(ref)
        for k in ref:
            assert tuple(out[k].shape) == tuple(ref[k].shape), k
            assert np.array_equal(out[k], ref[k].numpy()), k
        assert all(off % 256 == 0 for _, off, _, _, _ in plan)
    again = cached.collate_fn(items)                                     # items stay valid for the tensor-level collate
    assert torch.equal(again["mel_specs"], cached.collate_fn(items)["mel_specs"])


def test_checkpoint_assembles_from_plain_cpu_state():
    """The checkpoint dictionary is built from CPU state dicts alone (no engine, no GPU) — the function save_checkpoint
    uses and tests/golden/checkpoint_roundtrip.py feeds to the reference's load_checkpoint / KokoroTTS._load_model."""
    import torch
    from kokoro.training import checkpoint as ckpt
    from kokoro.training.config import TrainingConfig
    from kokoro_ruslan_amd import spec
    dims = spec.ModelDims(hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=96, dec_ff=96, var_filter=32, var_bins=16, mel=20, max_len=300)
    hp = spec.StepHyper()
    P = spec.init_params(dims, 1)
    sd = dict(P)
    sd.update(spec.make_buffers(dims))
    sd = {n: sd[n] for n in spec.state_dict_order(dims)}
    osd = ckpt.adamw_state_dict(list(P), lambda n: torch.zeros_like(P[n]), lambda n: torch.ones_like(P[n]), 5, hp, 1e-5)
    c = ckpt.assemble_checkpoint(model_sd=sd, ema_sd=sd, optimizer_sd=osd, hp=hp, dims=dims, config=TrainingConfig(), total_steps=100,
                                 epoch=0, loss=1.0, steps_done=5)
    ref_keys = {"epoch", "global_step", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "current_optimizer_step",
                "optimizer_steps_completed", "loss", "train_loss", "val_loss", "val_mel_loss", "val_stop_loss", "val_dur_loss",
                "best_val_loss", "best_val_epoch", "config", "model_metadata", "scheduler_config", "ema_model_state_dict", "ema_updates"}
    assert ref_keys <= set(c)
    assert len(osd["param_groups"]) == 10 and sum(len(g["params"]) for g in osd["param_groups"]) == len(P) == len(osd["state"])
    assert [g["group_type"] for g in osd["param_groups"]] == list(spec.GROUP_TYPES)
    assert c["scheduler_state_dict"]["_last_lr"] == [g["lr"] for g in osd["param_groups"]]
    assert c["model_metadata"]["architecture"]["hidden_dim"] == 128


def test_legacy_schedule_is_torch_cosine_warm_restarts_per_epoch():
    """use_onecycle_lr = False (reference trainer.py:789-799, 2885-2887): CosineAnnealingWarmRestarts(T_0, T_mult, eta_min) over the
    param groups' own initial lr (learning_rate x group multiplier), stepped once per epoch, no warm-up.  The oracle's and the host
    side's closed form against torch's scheduler itself, across two restarts — and the checkpoint's scheduler / group state loads into
    a fresh torch scheduler that continues with the same values."""
    import sys
    from kokoro.training import checkpoint as ckpt
    from kokoro_ruslan_amd import spec
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import kokoro_oracle as O
    hp = spec.StepHyper(use_onecycle_lr=False, lr_T_0=3, lr_T_mult=2, lr_eta_min=1e-6, learning_rate=5e-5)
    ohp = O.StepHyper(use_onecycle_lr=False, lr_T_0=3, lr_T_mult=2, lr_eta_min=1e-6, learning_rate=5e-5)
    mults = [m for m, _ in spec.group_lr_mult_wd(hp)]
    make = lambda: torch.optim.AdamW([{"params": [torch.nn.Parameter(torch.zeros(1))], "lr": hp.learning_rate * m} for m in mults])
    opt = make()
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=3, T_mult=2, eta_min=1e-6)
    for epoch in range(25):                                  # restarts after 3 and 9 epochs, the third period runs past the end
        f = spec.cosine_restart_factor(epoch, 3, 2)
        assert f == O.cosine_restart_factor(epoch, 3, 2)
        for g, m in zip(opt.param_groups, mults):
            want = g["lr"]
            assert abs(O.legacy_group_lr(ohp, m, epoch) - want) <= 1e-12 * want + 1e-18, (epoch, m)
            assert abs(hp.lr_eta_min + (hp.learning_rate * m - hp.lr_eta_min) * f - want) <= 1e-12 * want + 1e-18
        if epoch == 7:                                       # the checkpoint written at the end of epoch 7 (after the scheduler step)
            dims = spec.ModelDims(hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=96, dec_ff=96, var_filter=32, var_bins=16, mel=20, max_len=300)
            P = spec.init_params(dims, 1)
            osd = ckpt.adamw_state_dict(list(P), lambda n: torch.zeros_like(P[n]), lambda n: torch.ones_like(P[n]), 5, hp, 1e-5)
            c = ckpt.assemble_checkpoint(model_sd={}, ema_sd=None, optimizer_sd=osd, hp=hp, dims=dims, config=TrainingConfig(),
                                         total_steps=100, epoch=7, loss=1.0, steps_done=5)
            saved = c
        opt.step()
        sch.step()
    # resume from the epoch-7 checkpoint with torch's own classes, like the reference's load_checkpoint does
    opt2 = make()
    sch2 = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt2, T_0=3, T_mult=2, eta_min=1e-6)
    for g, s in zip(opt2.param_groups, saved["optimizer_state_dict"]["param_groups"]):
        g["lr"], g["initial_lr"] = s["lr"], s["initial_lr"]
    sch2.load_state_dict(saved["scheduler_state_dict"])
    for epoch in range(8, 12):
        for g, m in zip(opt2.param_groups, mults):
            assert abs(g["lr"] - O.legacy_group_lr(ohp, m, epoch)) <= 1e-12 * g["lr"] + 1e-18, (epoch, m)
        opt2.step()
        sch2.step()
    assert saved["scheduler_config"]["onecycle_steps"] is None
