"""GPU suite: `kokoro-train` end to end on a synthetic feature cache — epochs, validation on the EMA weights,
checkpoints in the reference layout, strict resume."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

REF_CKPT_KEYS = {"epoch", "global_step", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict",
                 "current_optimizer_step", "optimizer_steps_completed", "loss", "train_loss", "val_loss", "val_mel_loss",
                 "val_stop_loss", "val_dur_loss", "best_val_loss", "best_val_epoch", "config", "model_metadata",
                 "scheduler_config", "ema_model_state_dict", "ema_updates"}


def _fake_cache(root, n=12):
    from kokoro_ruslan_amd.synthetic import synthetic_batch
    d = root / ".feature_cache"
    d.mkdir(parents=True)
    g = torch.Generator().manual_seed(0)
    for i in range(n):
        T, P = int(torch.randint(40, 120, (1,), generator=g)), int(torch.randint(4, 14, (1,), generator=g))
        b = synthetic_batch(1, T, P, seed=i)
        torch.save({"mel_spec": b["mel_specs"][0].T.contiguous(), "phoneme_indices": b["phoneme_indices"][0],
                    "stress_indices": b["stress_indices"][0], "phoneme_durations": b["phoneme_durations"][0],
                    "stop_token_targets": b["stop_token_targets"][0], "pitch": b["pitches"][0], "energy": b["energies"][0],
                    "text": f"utt {i}", "audio_file": f"utt{i:03d}", "mel_length": T, "phoneme_length": P,
                    "_cache_version": 7}, d / f"utt{i:03d}.pt")


def test_kokoro_train_cli_end_to_end(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from kokoro.cli.training import main
    from kokoro.training import checkpoint as ckpt
    from kokoro_ruslan_amd import spec
    corpus, out = tmp_path / "corpus", tmp_path / "model"
    _fake_cache(corpus)
    args = ["--corpus", str(corpus), "--output", str(out), "--no-mfa", "--no-dynamic-batching", "--batch-size", "4",
            "--epochs", "2", "--save-every", "1", "--val-split", "0.25"]
    assert main(args) == 0
    files = sorted(os.listdir(out))
    assert "checkpoint_epoch_1.pth" in files and "checkpoint_epoch_2.pth" in files and "kokoro_russian_final.pth" in files
    c = torch.load(out / "checkpoint_epoch_2.pth", map_location="cpu", weights_only=False)
    assert REF_CKPT_KEYS <= set(c.keys())
    assert list(c["model_state_dict"].keys()) == spec.state_dict_order(spec.ModelDims())
    assert type(c["config"]).__module__ == "kokoro.training.config"
    arch = c["model_metadata"]["architecture"]
    assert c["model_metadata"]["schema_version"] == 2 and arch["hidden_dim"] == 512 and arch["vocab_size"] == 59
    osd = c["optimizer_state_dict"]
    assert [g["group_type"] for g in osd["param_groups"]] == list(spec.GROUP_TYPES)
    assert sum(len(g["params"]) for g in osd["param_groups"]) == 308 and len(osd["state"]) == 308
    steps = c["optimizer_steps_completed"]
    assert steps == 4                                        # 9 train utts / batch 4 = 3 batches, G=2 -> 2 steps/epoch
    assert c["val_loss"] is not None and c["val_loss"] == c["val_loss"]          # finite number
    fin = torch.load(out / "kokoro_russian_final.pth", map_location="cpu", weights_only=False)
    assert set(fin.keys()) == {"model_state_dict", "config", "model_metadata"}
    # EMA differs from the live weights after training and both are finite
    w, e = c["model_state_dict"]["decoder.layers.0.ff.linear1.weight"], c["ema_model_state_dict"]["decoder.layers.0.ff.linear1.weight"]
    assert torch.isfinite(w).all() and torch.isfinite(e).all() and not torch.equal(w, e)
    # strict resume: a third epoch continues from the saved optimizer-step counter
    assert main(args[:-6] + ["--epochs", "3", "--save-every", "1", "--val-split", "0.25", "--resume", "auto"]) == 0
    c3 = torch.load(out / "checkpoint_epoch_3.pth", map_location="cpu", weights_only=False)
    assert c3["optimizer_steps_completed"] == steps + 2 and c3["epoch"] == 2
    # architecture mismatch is refused like the reference's strict loader
    from kokoro_ruslan_amd.engine import KokoroEngine
    small = KokoroEngine(spec.ModelDims(hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=96, dec_ff=96, var_filter=32,
                                        var_bins=16, mel=80, max_len=4000), spec.StepHyper())
    with pytest.raises(RuntimeError, match="architecture mismatch"):
        ckpt.load_checkpoint(small, str(out / "checkpoint_epoch_3.pth"))


def test_validation_metrics_match_per_sample_loops(tmp_path):
    """validate_epoch's masked, on-device spectral convergence / F0-RMSE == the reference's per-sample Python loops
    (trainer.py:1866-1910) on the same predictions."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import math
    from kokoro.cli.cli import create_config_from_args, parse_arguments
    from kokoro.data.cached import collate_fn
    from kokoro.training.trainer import KokoroTrainer, cap_batch
    corpus = tmp_path / "corpus"
    _fake_cache(corpus, n=14)
    import sys
    argv, sys.argv = sys.argv, ["kokoro-train", "--corpus", str(corpus), "--output", str(tmp_path / "m"), "--no-mfa",
                                "--no-dynamic-batching", "--batch-size", "3", "--epochs", "1", "--val-split", "0.4"]
    try:
        cfg = create_config_from_args(parse_arguments())
    finally:
        sys.argv = argv
    tr = KokoroTrainer(cfg)
    val = tr.validate_epoch()
    e = tr.engine
    sc_sum = sc_n = f0_sum = f0_n = 0.0
    with e.fp32_math():
        for idxs in tr._val_batches():
            batch = cap_batch(tr._to_device(collate_fn([tr.val_dataset[j] for j in idxs])))
            out = e.forward_backward(batch, backward=False)
            bs = bn = fs = fn = 0.0
            for b in range(batch["mel_specs"].size(0)):
                L = int(batch["mel_lengths"][b])
                ref, pred = batch["mel_specs"][b, :L], out["mel"][b, :L]
                den = float(torch.norm(ref))
                if den > 0:
                    bs += float(torch.norm(ref - pred)) / den
                    bn += 1
                fs += math.sqrt(float(torch.mean((batch["pitches"][b, :L] - out["pitch"][b, :L]) ** 2)))
                fn += 1
            sc_sum, sc_n, f0_sum, f0_n = sc_sum + bs / bn, sc_n + 1, f0_sum + fs / fn, f0_n + 1
    assert abs(val["spectral_convergence"] - sc_sum / sc_n) < 1e-5 * max(1.0, sc_sum / sc_n)
    assert abs(val["f0_rmse"] - f0_sum / f0_n) < 1e-5
    assert len(tr._val_batches()) >= 2 and sorted(i for b in tr._val_batches() for i in b) == list(range(len(tr.val_dataset)))
