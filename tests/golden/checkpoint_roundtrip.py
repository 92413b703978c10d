#!/usr/bin/env python3
"""Checkpoint hand-off proof: a checkpoint written by THIS repo's writer is read by the reference's
own loaders.  Runs ONLY in the build container (needs /root/reference); no GPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/checkpoint_roundtrip.py

Two processes, because both trees ship a package called `kokoro`:
  write  (this repo on sys.path)   kokoro.training.checkpoint.assemble_checkpoint — the function save_checkpoint uses —
         from plain CPU state: seeded weights, a different EMA replica, random Adam moments, 37 completed steps;
         writes checkpoint_epoch_4.pth, kokoro_russian_final.pth and the expected tensors' checksums.
  load   (/root/reference/src on sys.path)
         1. training/checkpoint_manager.py:load_checkpoint (:287-544) into the reference's KokoroModel + the AdamW that
            KokoroTrainer._setup_optimizer builds (10 groups) + its OneCycleLR: strict state-dict load, architecture
            metadata validation, optimizer moments land on the right parameters, start epoch / loss come back;
         2. inference/inference.py:KokoroTTS._load_model (:109-200): 'auto' prefers the EMA replica, 'model' the live
            weights; kokoro_russian_final.pth (no EMA inside) loads as the live weights.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"


def digest(t) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def write(out: str) -> None:
    sys.path.insert(0, ROOT)
    import torch
    from kokoro.training import checkpoint as ckpt
    from kokoro.training.config import TrainingConfig
    from kokoro_ruslan_amd import spec
    dims, hp, cfg = spec.ModelDims(), spec.StepHyper(), TrainingConfig()
    names = list(spec.param_shapes(dims))
    P = spec.init_params(dims, 3)
    g = torch.Generator().manual_seed(9)
    model_sd, ema_sd = {}, {}
    buffers = spec.make_buffers(dims)
    for n in spec.state_dict_order(dims):
        t = P[n] if n in P else buffers[n]
        model_sd[n] = t.clone()
        ema_sd[n] = t.clone() + (0.01 * torch.randn(t.shape, generator=g) if n in P else 0)
    m = {n: 0.01 * torch.randn(P[n].shape, generator=g) for n in names}
    v = {n: 1e-4 * torch.rand(P[n].shape, generator=g) for n in names}
    steps = 37
    osd = ckpt.adamw_state_dict(names, lambda n: m[n], lambda n: v[n], steps, hp, last_base_lr=1.6e-6)
    c = ckpt.assemble_checkpoint(model_sd=model_sd, ema_sd=ema_sd, optimizer_sd=osd, hp=hp, dims=dims, config=cfg,
                                 total_steps=3000, epoch=3, loss=1.2345, steps_done=steps, val={"total": 1.5, "mel": 1.1, "stop": 0.2, "dur": 0.3},
                                 best_val_loss=1.5, best_val_epoch=3)
    os.makedirs(out, exist_ok=True)
    torch.save(c, os.path.join(out, "checkpoint_epoch_4.pth"))
    groups = [[] for _ in range(10)]
    for n in names:
        groups[spec.param_group_of(n)].append(n)
    order = [n for grp in groups for n in grp]
    probe = ["text_embedding.weight", "decoder.layers.3.cross_attn.w_k.weight", "stop_token_predictor.bias",
             "duration_adaptor.variance_adaptor.pitch_embedding.weight", "transformer_encoder_layers.5.ff.linear2.weight"]
    json.dump({"model": {n: digest(model_sd[n]) for n in model_sd}, "ema": {n: digest(ema_sd[n]) for n in ema_sd},
               "exp_avg": {n: digest(m[n]) for n in probe}, "exp_avg_sq": {n: digest(v[n]) for n in probe},
               "group_lr": [g_["lr"] for g_ in osd["param_groups"]], "steps": steps, "order": order},
              open(os.path.join(out, "expected.json"), "w"))
    print(f"[write] checkpoint_epoch_4.pth: {len(model_sd)} tensors, {len(osd['state'])} optimizer entries, keys {sorted(c)[:6]}...")


def write_final(out: str) -> None:
    sys.path.insert(0, ROOT)
    import torch
    from kokoro.training import checkpoint as ckpt
    from kokoro.training.config import TrainingConfig
    from kokoro_ruslan_amd import spec
    c = torch.load(os.path.join(out, "checkpoint_epoch_4.pth"), map_location="cpu", weights_only=False)
    torch.save({"model_state_dict": c["model_state_dict"], "config": TrainingConfig(),
                "model_metadata": ckpt.build_model_metadata(TrainingConfig(), spec.ModelDims())}, os.path.join(out, "kokoro_russian_final.pth"))
    print("[write] kokoro_russian_final.pth")


def load(out: str, phase: str) -> None:
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import types
    from unittest import mock
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SW", (), {"__init__": lambda s, *a, **k: None, "__getattr__": lambda s, n: (lambda *a, **k: None)})
    sys.modules["torch.utils.tensorboard"] = tb
    for name in ("torchaudio", "torchaudio.transforms", "torchaudio.functional"):
        sys.modules[name] = mock.MagicMock()
    import logging
    import torch
    logging.disable(logging.CRITICAL)
    import kokoro
    assert kokoro.__file__.startswith(REF), kokoro.__file__
    from kokoro.data.russian_phoneme_processor import RussianPhonemeProcessor
    from kokoro.training import checkpoint_manager as cm
    from kokoro.training.config import TrainingConfig
    from kokoro.training.trainer import KokoroTrainer
    from kokoro.model.model import KokoroModel
    exp = json.load(open(os.path.join(out, "expected.json")))
    proc = RussianPhonemeProcessor()
    assert len(proc.phoneme_to_id) == 59
    if phase == "resume":
        cm.save_phoneme_processor(proc, out)
        cfg = TrainingConfig()
        model = KokoroModel(len(proc.phoneme_to_id), cfg.n_mels, cfg.hidden_dim, n_encoder_layers=cfg.n_encoder_layers, n_heads=cfg.n_heads,
                            encoder_ff_dim=cfg.encoder_ff_dim, encoder_dropout=cfg.encoder_dropout, n_decoder_layers=cfg.n_decoder_layers,
                            decoder_ff_dim=cfg.decoder_ff_dim, max_decoder_seq_len=cfg.max_decoder_seq_len,
                            variance_filter_size=cfg.variance_filter_size, variance_kernel_size=cfg.variance_kernel_size,
                            n_variance_bins=cfg.n_variance_bins, qk_norm=True, ffn_output_norm=True)
        tr = KokoroTrainer.__new__(KokoroTrainer)
        tr.config, tr.model, tr.device, tr.device_type = cfg, model, torch.device("cpu"), "cpu"
        tr.use_mixed_precision, tr.scaler, tr.mixed_precision_stats = False, None, {}
        tr._setup_optimizer()
        tr.dataloader = type("DL", (), {"__len__": lambda s: 1500})()
        cfg.num_epochs, cfg.gradient_accumulation_steps = 2, 1
        tr._setup_scheduler()
        start_epoch, best_loss, p2 = cm.load_checkpoint(os.path.join(out, "checkpoint_epoch_4.pth"), model, tr.optimizer, tr.scheduler, out)
        assert start_epoch == 4 and abs(best_loss - 1.2345) < 1e-12 and isinstance(p2, RussianPhonemeProcessor)
        sd = model.state_dict()
        assert list(sd.keys()) == list(exp["model"].keys())
        for n, t in sd.items():
            assert digest(t) == exp["model"][n], f"model tensor {n} differs after the reference's strict load"
        id2name = {id(p): n for n, p in model.named_parameters()}
        flat = [id2name[id(p)] for g_ in tr.optimizer.param_groups for p in g_["params"]]
        assert flat == exp["order"], "optimizer parameter numbering differs from the reference's groups"
        assert [g_["group_type"] for g_ in tr.optimizer.param_groups] == ["encoder", "encoder", "decoder_other", "decoder_other",
                                                                          "decoder_attn", "decoder_attn", "decoder_ffn", "decoder_ffn",
                                                                          "variance_embed", "stop_head"]
        name2p = dict(model.named_parameters())
        for n, d in exp["exp_avg"].items():
            st = tr.optimizer.state[name2p[n]]
            assert digest(st["exp_avg"]) == d and digest(st["exp_avg_sq"]) == exp["exp_avg_sq"][n] and float(st["step"]) == exp["steps"], n
        assert [g_["lr"] for g_ in tr.optimizer.param_groups] == exp["group_lr"]
        assert tr.scheduler.last_epoch == 0 and tr.scheduler.total_steps == 3000 - 1200
        tr.optimizer.step()                                           # the restored state is usable by torch's fused-free AdamW
        print(f"[load ] reference load_checkpoint: strict load of {len(sd)} tensors, 308 optimizer states on the right parameters, "
              f"start epoch {start_epoch}")
    else:
        from kokoro.inference.inference import KokoroTTS
        for pref, which in (("auto", "ema"), ("ema", "ema"), ("model", "model")):
            tts = KokoroTTS.__new__(KokoroTTS)
            tts.model_dir, tts.device, tts.weights_preference, tts.phoneme_processor = __import__("pathlib").Path(out), torch.device("cpu"), pref, proc
            tts.inference_max_len = tts.inference_stop_threshold = tts.inference_min_len_ratio = tts.inference_min_len_floor = None
            tts._explicit_inference_max_len = tts._explicit_inference_stop_threshold = False
            tts._explicit_inference_min_len_ratio = tts._explicit_inference_min_len_floor = False
            m = tts._load_model()
            for n, t in m.state_dict().items():
                assert digest(t) == exp[which][n], (pref, n)
            assert not m.training
            label = "live weights (the final file carries no EMA replica)" if phase == "final" else f"{which} weights"
            print(f"[load ] reference KokoroTTS._load_model(weights='{pref}') ({phase}): {label}, inference controls "
                  f"max_len={tts.inference_max_len} stop={tts.inference_stop_threshold}")
            if phase == "final":
                break


def main() -> None:
    if len(sys.argv) >= 3:
        {"write": write, "write_final": write_final}.get(sys.argv[1], lambda o: load(o, sys.argv[1]))(sys.argv[2])
        return
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    with tempfile.TemporaryDirectory() as out:
        for phase in ("write", "resume", "latest"):
            subprocess.run([sys.executable, os.path.abspath(__file__), phase, out], check=True, env=env)
        # with kokoro_russian_final.pth present the inference loader picks it (no EMA inside: live weights)
        subprocess.run([sys.executable, os.path.abspath(__file__), "write_final", out], check=True, env=env)
        exp = json.load(open(os.path.join(out, "expected.json")))
        exp["ema"] = exp["model"]
        json.dump(exp, open(os.path.join(out, "expected.json"), "w"))
        subprocess.run([sys.executable, os.path.abspath(__file__), "final", out], check=True, env=env)
    print("CHECKPOINT ROUND TRIP THROUGH THE REFERENCE LOADERS: OK")


if __name__ == "__main__":
    main()
